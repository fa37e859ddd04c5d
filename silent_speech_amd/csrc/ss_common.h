// Shared device helpers for the Silent-Speech hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ss_hotpath.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SS_WAVE 64

// launch-site error handling: every entry point returns 0 or a negative ss_status
#define SS_REQUIRE(cond, code) \
  do {                         \
    if (!(cond)) return (code); \
  } while (0)

static inline int ss_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SS_OK : SS_ERR_LAUNCH;
}

// v_mfma_f32_16x16x4_f32: A[i=lane&15][k=lane>>4], B[k=lane>>4][j=lane&15],
// D[row=(lane>>4)*4+reg][col=lane&15].  Exact f32 fma chain.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Fence for the instruction scheduler: keeps a batch of ds_reads issued ahead of the MFMAs that consume them
// (left alone, hipcc sinks every read next to its MFMA and waits lgkmcnt(0) in between: one LDS latency per MFMA).
#define SS_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// LDS-DMA: 64 lanes x 16 B from per-lane global addresses to lds_byte_addr + lane*16 (wave-uniform base), no VGPRs.
// Inline asm on purpose: hipcc does not track it, so the workgroup barriers between issue and use do not drain it;
// the consumer waits with ss_dma_wait() before its barrier (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void ss_dma16(const void* gsrc_lane, unsigned lds_byte_addr) {
  lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);  // wave-uniform by contract: make it an SGPR
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc_lane), "s"(lds_byte_addr)
               : "memory");
}
// same with 4 bytes per lane (lane l -> lds_byte_addr + 4 l)
__device__ __forceinline__ void ss_dma4(const void* gsrc_lane, unsigned lds_byte_addr) {
  lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc_lane), "s"(lds_byte_addr)
               : "memory");
}
__device__ __forceinline__ void ss_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Wave-wide reductions on the DPP crossbar instead of __shfl_xor (which lowers to ds_bpermute_b32: one LDS round
// trip, ~100 cycles, per butterfly level).  Four DPP levels leave every lane of a 16-lane row with its row's result,
// four v_readlane + three VALU ops combine the rows: ~40 cycles for a 64-lane reduction instead of ~600.
#define SS_DPP_F(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, false))
#define SS_DPP_I(v, ctrl) __builtin_amdgcn_update_dpp(0, (int)(v), ctrl, 0xf, 0xf, false)
constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7 - i inside each half row
constexpr int DPP_MIRROR = 0x140;      // lane i <-> 15 - i inside each row

__device__ __forceinline__ float row_sum(float v) {  // sum over each 16-lane row, result in every lane of the row
  v += SS_DPP_F(v, DPP_XOR1);
  v += SS_DPP_F(v, DPP_XOR2);
  v += SS_DPP_F(v, DPP_HALF_MIRROR);
  v += SS_DPP_F(v, DPP_MIRROR);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
  v = row_sum(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
  v += (unsigned)SS_DPP_I(v, DPP_XOR1);
  v += (unsigned)SS_DPP_I(v, DPP_XOR2);
  v += (unsigned)SS_DPP_I(v, DPP_HALF_MIRROR);
  v += (unsigned)SS_DPP_I(v, DPP_MIRROR);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 0) + (unsigned)__builtin_amdgcn_readlane((int)v, 16) +
         (unsigned)__builtin_amdgcn_readlane((int)v, 32) + (unsigned)__builtin_amdgcn_readlane((int)v, 48);
}

__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, SS_DPP_F(v, DPP_XOR1));
  v = fmaxf(v, SS_DPP_F(v, DPP_XOR2));
  v = fmaxf(v, SS_DPP_F(v, DPP_HALF_MIRROR));
  v = fmaxf(v, SS_DPP_F(v, DPP_MIRROR));
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// v_exp_f32 + v_rcp_f32 (1 ulp each): the gates sit on the recurrence's critical path between two workgroup barriers,
// an IEEE divide there costs ~10 VALU instructions per element; the logit tolerance (1e-3) leaves 4 orders of margin
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// tanh via one exp; saturates cleanly (exp->inf gives 1, exp->0 gives -1)
__device__ __forceinline__ float tanh_f(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// Philox4x32-10, one counter per 4 consecutive elements: the dropout stream of ss_dropout (pool_head.hip), also drawn
// inside the fused kernels so a mask never has to be materialised.
__device__ __forceinline__ void philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                        uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// keep-scales of the 4 elements [4q, 4q+4) of a dropout stream (p > 0)
__device__ __forceinline__ f32x4 drop_scale4(long q, float p, uint64_t seed, uint64_t offset) {
  uint32_t rnd[4];
  const uint64_t ctr = offset + (uint64_t)q;
  philox4((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), rnd);
  const uint32_t thr = (uint32_t)((double)p * 4294967296.0);
  const float keep = 1.0f / (1.0f - p);
  return f32x4{rnd[0] >= thr ? keep : 0.f, rnd[1] >= thr ? keep : 0.f, rnd[2] >= thr ? keep : 0.f, rnd[3] >= thr ? keep : 0.f};
}

// ReLU as torch.relu computes it: a NaN stays a NaN (fmaxf(NaN, 0) = 0 would swallow the poison a failed kernel leaves)
__device__ __forceinline__ float relu_f(float x) { return x < 0.f ? 0.f : x; }

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Correctly rounded float32 square root (what np.sqrt / np.linalg.norm give on float32): the double root, rounded once more to
// float32 -- exact, since 53 >= 2 * 24 + 2 bits.  HIP's __fsqrt_rn is the hardware's v_sqrt_f32, one ulp off on ~10 % of
// arguments (found against the reference's own extract_83_and_openness outputs, tests/golden/serving.npz).
__device__ __forceinline__ float ss_sqrt_rn_f32(float x) { return (float)__dsqrt_rn((double)x); }

// Compute units of the current device (MI355X: 256).  Every persistent grid, co-residency bound and workspace layout that depends
// on the chip's size asks here -- no launcher hard-codes 256.  One query per translation unit, then cached.
static inline int ss_device_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    int n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    cus = n > 0 ? n : 256;  // (no device: the size queries a CPU test makes answer for the target chip)
  }
  return cus;
}

// ---- diagnostic build only (-DSS_STAMP): per-stage cycle shares of the persistent kernels.
// Thread 0 of every workgroup accumulates s_memtime deltas between barriers; the real build has no stamps.
#ifdef SS_STAMP
#define SS_STAMP_SLOTS 24
#define SS_STAMP_WGS 512
// one table per translation unit (no relocatable device code): STAMP_TABLE(fn) defines it and its host reader
#define STAMP_TABLE(reader)                                                                           \
  __device__ unsigned long long ss_stamp_buf[SS_STAMP_WGS * SS_STAMP_SLOTS];                          \
  extern "C" int reader(unsigned long long* host_out) {                                               \
    if (hipDeviceSynchronize() != hipSuccess) return SS_ERR_LAUNCH;                                   \
    const size_t bytes_ = sizeof(unsigned long long) * SS_STAMP_WGS * SS_STAMP_SLOTS;                 \
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ss_stamp_buf), bytes_) != hipSuccess) return SS_ERR_LAUNCH; \
    void* dev_ = nullptr;  /* the table accumulates over launches: reading it clears it */            \
    if (hipGetSymbolAddress(&dev_, HIP_SYMBOL(ss_stamp_buf)) != hipSuccess) return SS_ERR_LAUNCH;     \
    return hipMemset(dev_, 0, bytes_) == hipSuccess ? SS_OK : SS_ERR_LAUNCH;                          \
  }
// Thread 0 adds every delta straight into the table with a no-return atomic: the timers cost three registers, not a
// 24-entry array (which pushed the 250-register kernels into scratch and distorted what it measured).
// STAMP_ENTRY at the top of the kernel: slot 14 = cycles before the frame loop, slots 12 / 13 = wall clock (100 MHz, the same
// counter on every CU) at entry / exit -- dispatch stagger and tail imbalance across workgroups
// (row of the table: SS_STAMP_ROW, the workgroup's x index unless the file defines its own before its first stamp)
#ifndef SS_STAMP_ROW
#define SS_STAMP_ROW ((int)blockIdx.x)
#endif
#define STAMP_ADD_(k, v)                                                                              \
  do {                                                                                                \
    if (SS_STAMP_ROW < SS_STAMP_WGS) atomicAdd(&ss_stamp_buf[SS_STAMP_ROW * SS_STAMP_SLOTS + (k)], (unsigned long long)(v)); \
  } while (0)
#define STAMP_ENTRY unsigned long long st_entry = clock64(), st_wall0 = wall_clock64()
#define STAMP_DECL                                   \
  unsigned long long st_last = clock64();            \
  if (threadIdx.x == 0) {                            \
    STAMP_ADD_(14, st_last - st_entry);              \
    STAMP_ADD_(12, st_wall0);                        \
  }
#define STAMP(k)                               \
  do {                                         \
    if (threadIdx.x == 0) {                    \
      unsigned long long t_ = clock64();       \
      STAMP_ADD_(k, t_ - st_last);             \
      st_last = t_;                            \
    }                                          \
  } while (0)
// (A stamp never contains a barrier or a wait of its own: a diagnostic macro that synchronises only under -DSS_STAMP
// hid a missing production barrier for a whole round.  Where a stamp wants a barrier in front, the kernel has it.)
#define STAMP_FLUSH()                                        \
  do {                                                       \
    if (threadIdx.x == 0) STAMP_ADD_(13, wall_clock64());    \
  } while (0)
#else
#define STAMP_ENTRY
#define STAMP_DECL
#define STAMP(k)
#define STAMP_FLUSH()
#define STAMP_TABLE(reader)
#endif
