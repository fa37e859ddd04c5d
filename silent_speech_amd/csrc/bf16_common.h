// bf16-MFMA helpers of the BASELINE config-5 path (96x96 ROI, CNN 16/32/64/96, BiGRU H = 512): gfx950 only.
//
// v_mfma_f32_16x16x32_bf16 operand maps (cdna_hip_programming.md section 3):
//   A: lane l holds A[row = l & 15][k = 8 (l >> 4) + j], j = 0..7   (8 bf16 = 16 bytes)
//   B: lane l holds B[k = 8 (l >> 4) + j][col = l & 15]
//   D: lane l holds D[row = 4 (l >> 4) + r][col = l & 15], r = 0..3 (f32)
// f32 accumulation everywhere; bf16 only as the multiplicands.
#pragma once
#include "ss_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short bf16_t;  // storage type of a bf16 value in memory

__device__ __forceinline__ f32x4 mfma_bf16(s16x8 a, s16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// round-to-nearest-even f32 -> bf16 (hipcc emits v_cvt_pk_bf16_f32; a NaN stays a NaN)
__device__ __forceinline__ bf16_t to_bf16(float x) { return __builtin_bit_cast(bf16_t, (__bf16)x); }
__device__ __forceinline__ float from_bf16(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) { return (unsigned)to_bf16(lo) | ((unsigned)to_bf16(hi) << 16); }
__device__ __forceinline__ uint2 pack_bf16x4(float a, float b, float c, float d) { return uint2{pack_bf16(a, b), pack_bf16(c, d)}; }

// 16-byte LDS read of an MFMA operand whose 8 k values are contiguous
__device__ __forceinline__ s16x8 lds_frag(const bf16_t* p) { return *reinterpret_cast<const s16x8*>(p); }

// MFMA operand out of a k-major LDS image (element (k, x) at base[k * ld + x], x = row of A / column of B):
// ds_read_b64_tr_b16 gathers a 4 (k) x 16 (x) block per 16-lane group and hands lane i column i.  Lane 4q + p of a group
// supplies the address of block row q, columns 4p..4p+3 (8-byte aligned).  Two reads give the lane its 8 k values
// k = 8 (l >> 4) + 0..7 for x = x0 + (l & 15).  ``base`` points at element (k0, x0) of the tile; EXEC must be all ones.
__device__ __forceinline__ s16x8 lds_frag_tr(const bf16_t* base, int ld, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const bf16_t* a0 = base + (8 * g + q) * ld + 4 * p;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * ld));
  return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
