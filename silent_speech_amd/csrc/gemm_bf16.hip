// bf16-MFMA GEMM of the BASELINE config-5 GRU layers (H = 512): C[M,N] (+)= opA[M,K] * opB[K,N] (+ bias[N]).
//
// Same operand conventions as ss_gemm_f32_batched (include/ss_hotpath.h): f32 operands in HBM, either k-contiguous
// ([row][k]) or k-major ([k][row]) with the storage-row remap that pairs dG[b][t] with h[b][t -+ 1].  The operands are
// rounded to bf16 (nearest even) while they are staged into LDS and multiplied on v_mfma_f32_16x16x32_bf16 with f32
// accumulation; C, the bias and every accumulation (split-K float atomics) stay f32.
//
// Operands may also be bf16 already (flags bit 3: the recurrence kernels and ss_cvt_bf16_rows leave bf16 copies of the layer
// inputs, the gate gradients and the weights): 16-byte loads straight into LDS, 64-deep k tiles.
// 128 x 128 x 32 tiles, 4 waves (2 x 2, 64 x 64 each = 16 accumulators), double-buffered LDS, one barrier per k tile,
// global loads of tile t+1 issued before the MFMAs of tile t.  k-contiguous operands sit in LDS as [row][32 + 8] and
// are read with ds_read_b128 (row stride 80 B = 20 banks: the 16 rows of a fragment cover all 64 banks); k-major
// operands sit as [k][128 + 8] (written with 8-byte stores as they arrive) and are read with the transposing
// ds_read_b64_tr_b16 -- no transpose pass, no strided global loads.
#include "bf16_common.h"

namespace {

STAMP_TABLE(ss_debug_stamps_gemm_bf16)

constexpr int BM = 128, BN = 128;

struct GemmBfParams {
  int M, N, K;
  const void* A; int lda, a_group, a_gstride, a_off;
  const void* B; int ldb, b_group, b_gstride, b_off;
  float* C; int ldc;
  const float* bias;
  int flags, splits;
  long sa, sb, sc, sbias;
};

__device__ __forceinline__ long remap_row(int r, int group, int gstride, int off) {
  return (long)(r / group) * gstride + (r % group) + off;
}

// LDS images of one operand tile (128 rows x BK k): k-contiguous operands as [row][BK + 16] (conflict-free under the lane groups of
// ds_read_b128: rows 0-3, 12-15 at chunk g with rows 4-11 at chunk g + 1), k-major operands as [k][128 + 24]: a transposing read takes 8-byte pieces of 4 k lines
// per 16-lane group and two groups (8 k lines apart) per pass -- with a line stride of 76 dwords (== 12 mod 64) the four lines
// of a group sit 12 banks apart and the second group 32 banks further: no conflicts (128 + 8 gave 2-way ones, a third of the
// kernel's active LDS cycles).
template <int BK> struct Tile {
  static constexpr int LDK = BK + 16, LDR = 128 + 24;  // LDK == 16 (mod 32): see Wmat in cnn_bf16.h
  static constexpr int ELEMS = (BM * LDK > BK * LDR) ? BM * LDK : BK * LDR;
};

// ---- f32 operands in HBM (rounded to bf16 on their way into LDS), BK = 32: 4 float4 per thread and tile
// KC = 1: storage [row][k]: thread -> (row = idx / 8, k = 4 (idx % 8)), idx = tid + 256 i.
// KC = 0: storage [k][row]: thread -> (k = idx / 32, row = 4 (idx % 32)).
template <int KC>
__device__ __forceinline__ void load_tile_f32(const float* __restrict__ src, int ld, int group, int gstride, int off, int row0, int rows,
                                              int k0, int k_end, int tid, f32x4 v[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i;
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    if (KC) {
      const int r = row0 + (idx >> 3), k = k0 + 4 * (idx & 7);
      if (r < rows && k < k_end) x = *reinterpret_cast<const f32x4*>(src + remap_row(r, group, gstride, off) * ld + k);
    } else {
      const int k = k0 + (idx >> 5), r = row0 + 4 * (idx & 31);
      if (k < k_end && r < rows) x = *reinterpret_cast<const f32x4*>(src + remap_row(k, group, gstride, off) * ld + r);
    }
    v[i] = x;
  }
}

template <int KC>
__device__ __forceinline__ void store_tile_f32(bf16_t* tile, int tid, const f32x4 v[4]) {
  using T = Tile<32>;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i;
    bf16_t* dst = KC ? tile + (idx >> 3) * T::LDK + 4 * (idx & 7) : tile + (idx >> 5) * T::LDR + 4 * (idx & 31);
    *reinterpret_cast<uint2*>(dst) = pack_bf16x4(v[i][0], v[i][1], v[i][2], v[i][3]);
  }
}

// ---- bf16 operands in HBM, BK = 64: 128 x 64 x 2 B = 16 KB per tile = 4 16-byte chunks per thread, no conversion.
// KC = 1: thread -> (row = idx / 8, k = 8 (idx % 8));  KC = 0: thread -> (k = idx / 16, row = 8 (idx % 16)).
// A chunk that starts inside the operand may run past its last row / k into the padding of the leading dimension (the caller
// pads ld to a multiple of 8 with zeros: ss_cvt_bf16_rows), so only the chunk's first element is tested.
// The storage-row remap (r / group) * gstride + r % group + off is an integer division per chunk: done ONCE, in front of the
// k loop; a k-major operand then walks its rows tile by tile with an add and one conditional carry (the first version divided
// per chunk and tile: 10 vector instructions per MFMA, the kernel was VALU-bound at 9 % MFMA occupancy).
// Loads are BUFFER loads with the range check doing the predication: a chunk outside the operand gets an offset beyond
// num_records and comes back as zeros.  (Loads inside `if (in range)` branches made hipcc wait vmcnt(0) before the LDS stores --
// it cannot count loads through divergent branches -- which also waited for the tiles just requested: no prefetch at all.)
typedef unsigned int gu32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;  // beyond num_records with or without the (small) scalar offset added
// The kernel is bound by vector-instruction ISSUE, not by the matrix pipes: an MFMA 16x16x32 holds the issue port for 8 of its 16
// cycles and every other vector instruction adds its full cost (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost') -- at 3
// address / predicate instructions per MFMA a k tile took 2 000 cycles of a SIMD's two waves for 1 024 cycles of MFMA.  So an
// operand without a row remap (every operand but the two of the d W_hh GEMMs) advances through k with the buffer instruction's
// SCALAR offset: its per-lane offsets are fixed for the whole K loop and the k range is only tested on a ragged last tile.
template <int KC>
struct TileLoader16 {
  __amdgpu_buffer_rsrc_t rs;
  unsigned boff[4];       // identity map: byte offset of the chunk at k = 0 (OOB: outside the operand)
                          // remapped:     KC = 1 the chunk's row at k = 0; KC = 0 its column in row 0
  int quo[4], rem[4];     // remapped, KC = 0: storage row of the chunk's current k = quo * gstride + rem + off
  int ld, group, gstride, off, dq, dm, k_end;
  bool ident;
  __device__ __forceinline__ void init(const bf16_t* __restrict__ src, int ld_, int group_, int gstride_, int off_, int row0, int rows,
                                       int k0, int k_end_, int tid) {
    ld = ld_; group = group_; gstride = gstride_; off = off_; k_end = k_end_;
    dq = 64 / group; dm = 64 % group;
    ident = group == 0x7FFFFFFF && off == 0;
    rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(src), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      if (KC) {
        const int r = row0 + (idx >> 3);
        boff[i] = r < rows ? (unsigned)((remap_row(r, group, gstride, off) * ld + 8 * (idx & 7)) * 2) : OOB;
        quo[i] = rem[i] = 0;
      } else {
        const int c = row0 + 8 * (idx & 15), k = k0 + (idx >> 4);
        boff[i] = c < rows ? (unsigned)(2 * c) + (ident ? (unsigned)(2 * (idx >> 4) * ld) : 0u) : OOB;
        quo[i] = k / group; rem[i] = k % group;
      }
    }
  }
  // the tile at k0 (tiles are requested in ascending order, one BK apart)
  __device__ __forceinline__ void load(int k0, int tid, s16x8 v[4]) {
    if (ident) {  // wave-uniform
      const int soff = KC ? 2 * k0 : 2 * k0 * ld;  // scalar: the whole tile's step through k
      if (k0 + 64 <= k_end) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)boff[i], soff, 0));
      } else {  // ragged last tile
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int idx = tid + 256 * i;
          const int k = KC ? k0 + 8 * (idx & 7) : k0 + (idx >> 4);
          v[i] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(k < k_end ? boff[i] : OOB), soff, 0));
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      unsigned o;
      if (KC) {
        const int k = k0 + 8 * (idx & 7);
        o = (boff[i] != OOB && k < k_end) ? boff[i] + 2u * (unsigned)k0 : OOB;
      } else {
        const int k = k0 + (idx >> 4);
        o = (boff[i] != OOB && k < k_end) ? boff[i] + 2u * (unsigned)(((long)quo[i] * gstride + rem[i] + off) * ld) : OOB;
        quo[i] += dq; rem[i] += dm;
        if (rem[i] >= group) { rem[i] -= group; ++quo[i]; }
      }
      v[i] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o, 0, 0));
    }
  }
};

template <int KC>
__device__ __forceinline__ void store_tile_b16(bf16_t* tile, int tid, const s16x8 v[4]) {
  using T = Tile<64>;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i;
    bf16_t* dst = KC ? tile + (idx >> 3) * T::LDK + 8 * (idx & 7) : tile + (idx >> 4) * T::LDR + 8 * (idx & 15);
    *reinterpret_cast<s16x8*>(dst) = v[i];
  }
}

// SRC16 = 0: f32 operands (BK = 32); 1: bf16 operands (BK = 64).  Two workgroups per CU (the LDS images take 41 / 74 KB).
template <int AKC, int BKC, int SRC16>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmBfParams p) {
  constexpr int BK = SRC16 ? 64 : 32;
  using T = Tile<BK>;
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];  // A0 B0 A1 B1
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int batch = blockIdx.z / p.splits, split = blockIdx.z % p.splits;
  float* C = p.C + batch * p.sc;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  // K range of this split, in whole k tiles
  const int nkt = (p.K + BK - 1) / BK;
  const int per = (nkt + p.splits - 1) / p.splits;
  const int kt0 = split * per, kt1 = min(nkt, kt0 + per);
  const int wm = (wv >> 1) * 64, wn = (wv & 1) * 64;

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // k tiles kt0 .. kt1-1.  Two register stages: the loads of tile t + 2 are issued before the MFMAs of tile t and written to LDS
  // at the end of tile t + 1, so an operand has two tile times (and the other workgroup of the CU) to arrive -- with one stage a
  // wave sat through a memory latency per tile (MFMA pipes 9 % busy).
  STAMP_ENTRY;
  STAMP_DECL;
  auto mainloop = [&](auto load_a, auto load_b, auto store_a, auto store_b, auto& va, auto& vb) {
    load_a(kt0 * BK, va[0]);
    load_b(kt0 * BK, vb[0]);
    if (kt0 + 1 < kt1) {
      load_a((kt0 + 1) * BK, va[1]);
      load_b((kt0 + 1) * BK, vb[1]);
    }
    store_a(lds, va[0]);
    store_b(lds + T::ELEMS, vb[0]);
    __syncthreads();
    STAMP(15);
    auto tile = [&](int kt, int cur) {  // cur = (kt - kt0) & 1, a compile-time constant at both call sites
      const bf16_t* As = lds + (2 * cur) * T::ELEMS;
      const bf16_t* Bs = lds + (2 * cur + 1) * T::ELEMS;
      if (kt + 2 < kt1) {  // stage `cur` was written to LDS at the end of the previous tile
        load_a((kt + 2) * BK, va[cur]);
        load_b((kt + 2) * BK, vb[cur]);
      }
      // every fragment of the tile is requested before its first MFMA, behind a scheduling fence.  Stage timers of one workgroup
      // alone on its CU (a d W_hh slice, 116 k tiles; finer stamps than the ones kept here): issuing the tile's 8 buffer loads and
      // 32 transposing reads 980 cycles, the reads landing 260, the 32 MFMAs 760 (512 without the timers), the wait for the
      // operands of the next tile + their LDS stores 970, the barrier 290 -- every phase waits for the one before it.  What the
      // kernel needs next is the f32 GEMM's recipe (gemm.hip): operands by LDS-DMA into a ring three tiles deep, no staging
      // registers, no LDS store instructions; the changes of round 3 (counted waits, scalar k advance, conflict-free strides)
      // moved it from 250 to 400 - 500 TFLOP/s.
      s16x8 fa[BK / 32][4], fb[BK / 32][4];
#pragma unroll
      for (int k2 = 0; k2 < BK / 32; ++k2)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int kk = 32 * k2;
          fa[k2][t] = AKC ? lds_frag(As + (wm + 16 * t + li) * T::LDK + kk + 8 * g) : lds_frag_tr(As + kk * T::LDR + wm + 16 * t, T::LDR, lane);
          fb[k2][t] = BKC ? lds_frag(Bs + (wn + 16 * t + li) * T::LDK + kk + 8 * g) : lds_frag_tr(Bs + kk * T::LDR + wn + 16 * t, T::LDR, lane);
        }
      SS_SCHED_FENCE();
#pragma unroll
      for (int k2 = 0; k2 < BK / 32; ++k2)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = mfma_bf16(fa[k2][a], fb[k2][b], acc[a][b]);
      SS_SCHED_FENCE();
      STAMP(1);
      if (kt + 1 < kt1) {
        store_a(lds + (2 * (cur ^ 1)) * T::ELEMS, va[cur ^ 1]);
        store_b(lds + (2 * (cur ^ 1) + 1) * T::ELEMS, vb[cur ^ 1]);
      }
      STAMP(2);
      __syncthreads();
      STAMP(3);
    };
    int kt = kt0;
    for (; kt + 1 < kt1; kt += 2) {
      tile(kt, 0);
      tile(kt + 1, 1);
    }
    if (kt < kt1) tile(kt, 0);
  };

  if (kt0 < kt1) {
    if constexpr (SRC16) {
      const bf16_t* A = static_cast<const bf16_t*>(p.A) + batch * p.sa;
      const bf16_t* B = static_cast<const bf16_t*>(p.B) + batch * p.sb;
      s16x8 va[2][4], vb[2][4];
      TileLoader16<AKC> la;
      TileLoader16<BKC> lb;
      la.init(A, p.lda, p.a_group, p.a_gstride, p.a_off, m0, p.M, kt0 * BK, p.K, tid);
      lb.init(B, p.ldb, p.b_group, p.b_gstride, p.b_off, n0, p.N, kt0 * BK, p.K, tid);
      mainloop([&](int k0, s16x8* v) { la.load(k0, tid, v); }, [&](int k0, s16x8* v) { lb.load(k0, tid, v); },
               [&](bf16_t* t, const s16x8* v) { store_tile_b16<AKC>(t, tid, v); },
               [&](bf16_t* t, const s16x8* v) { store_tile_b16<BKC>(t, tid, v); }, va, vb);
    } else {
      const float* A = static_cast<const float*>(p.A) + batch * p.sa;
      const float* B = static_cast<const float*>(p.B) + batch * p.sb;
      f32x4 va[2][4], vb[2][4];
      mainloop([&](int k0, f32x4* v) { load_tile_f32<AKC>(A, p.lda, p.a_group, p.a_gstride, p.a_off, m0, p.M, k0, p.K, tid, v); },
               [&](int k0, f32x4* v) { load_tile_f32<BKC>(B, p.ldb, p.b_group, p.b_gstride, p.b_off, n0, p.N, k0, p.K, tid, v); },
               [&](bf16_t* t, const f32x4* v) { store_tile_f32<AKC>(t, tid, v); },
               [&](bf16_t* t, const f32x4* v) { store_tile_f32<BKC>(t, tid, v); }, va, vb);
    }
  }

  // epilogue: D row = 4 g + r, column = li
  const bool accumulate = p.flags & 1, atomic = (p.flags & 4) || p.splits > 1;
  const float* bias = p.bias ? p.bias + batch * p.sbias : nullptr;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int n = n0 + wn + 16 * b + li;
    if (n >= p.N) continue;
    const float bv = (bias && split == 0) ? bias[n] : 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + 16 * a + 4 * g + r;
        if (m >= p.M) continue;
        float* dst = C + (long)m * p.ldc + n;
        const float v = acc[a][b][r] + bv;
        if (atomic) atomicAdd(dst, v);
        else if (accumulate) *dst += v;
        else *dst = v;
      }
  }
  STAMP(4);
  STAMP_FLUSH();
}

// (A 256 x 256 x 64 tile with 8 waves of 128 x 64 -- half the operand bytes per FLOP -- was built and measured in round 3: the same
// 400 - 470 TFLOP/s on the large shapes, half the rate on the small weight-gradient outputs.  The stage timers
// (tools/gemm_bf16_stamp.py) put a k tile of this kernel at 3.7x its MFMA time inside the multiply phase itself: fragment reads and
// MFMAs of a wave do not overlap as hipcc schedules them, and the padded [row][k] image is 2-way conflicted under the real lane
// groups of ds_read_b128 (MI355X_MICROARCH.md, LDS table).  Next step: explicit read-ahead of the fragments and an XOR-swizzled
// unpadded image; not a bigger tile.)

template <int AKC, int BKC, int SRC16>
int launch_one(const GemmBfParams& p, dim3 grid, hipStream_t s) {
  constexpr size_t lds = 4 * Tile<SRC16 ? 64 : 32>::ELEMS * sizeof(bf16_t);
  static bool attr_set = false;  // the bf16-operand tiles take 72 KB of dynamic LDS: above the 64 KB a launch gets by default
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<AKC, BKC, SRC16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return SS_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_bf16_kernel<AKC, BKC, SRC16>), grid, dim3(256), lds, s, p);
  return ss_launch_status();
}

template <int SRC16>
int launch_gemm_bf16(int a_kcontig, int b_kcontig, const GemmBfParams& p, dim3 grid, hipStream_t s) {
  if (a_kcontig && b_kcontig) return launch_one<1, 1, SRC16>(p, grid, s);
  if (a_kcontig) return launch_one<1, 0, SRC16>(p, grid, s);
  if (b_kcontig) return launch_one<0, 1, SRC16>(p, grid, s);
  return launch_one<0, 0, SRC16>(p, grid, s);
}

}  // namespace

extern "C" int ss_gemm_bf16_batched(int a_kcontig, int b_kcontig, int M, int N, int K, const void* A, int lda, int a_group,
                                    int a_gstride, int a_off, const void* B, int ldb, int b_group, int b_gstride, int b_off,
                                    float* C, int ldc, const float* bias, int flags, int splits, int batch, long stride_a,
                                    long stride_b, long stride_c, long stride_bias, ss_stream_t stream) {
  SS_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0 && splits > 0, SS_ERR_ARG);
  SS_REQUIRE(a_group > 0 && b_group > 0, SS_ERR_ARG);
  SS_REQUIRE(!(flags & ~13), SS_ERR_UNSUPPORTED);                // bit0 accumulate, bit2 atomics, bit3 operands are bf16 in HBM
  SS_REQUIRE(splits == 1 || (flags & 1), SS_ERR_ARG);            // K slices add into a C the caller initialised
  SS_REQUIRE(!(flags & 4) || (flags & 1), SS_ERR_ARG);
  const bool src16 = flags & 8;
  // 16-byte loads along the contiguous dimension of either layout: 4 floats / 8 bf16 (bf16 operands: the leading dimension is
  // padded to it, the extent itself need not be)
  const int q = src16 ? 8 : 4;
  SS_REQUIRE(lda % q == 0 && ldb % q == 0 && stride_a % q == 0 && stride_b % q == 0, SS_ERR_UNSUPPORTED);
  SS_REQUIRE(src16 || ((a_kcontig ? K : M) % 4 == 0 && (b_kcontig ? K : N) % 4 == 0), SS_ERR_UNSUPPORTED);
  SS_REQUIRE(!src16 || ((a_kcontig ? K : M) <= lda && (b_kcontig ? K : N) <= ldb), SS_ERR_ARG);
  // (bf16 operands are read through buffer descriptors: an operand of one batch entry must span less than 4 GB)
  SS_REQUIRE((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0, SS_ERR_UNSUPPORTED);
  GemmBfParams p;
  p.M = M; p.N = N; p.K = K;
  p.A = A; p.lda = lda; p.a_group = a_group; p.a_gstride = a_gstride; p.a_off = a_off;
  p.B = B; p.ldb = ldb; p.b_group = b_group; p.b_gstride = b_gstride; p.b_off = b_off;
  p.C = C; p.ldc = ldc; p.bias = bias; p.flags = flags;
  const int bk = src16 ? 64 : 32;
  const int nkt = (K + bk - 1) / bk;
  p.splits = splits < nkt ? splits : nkt;
  p.sa = stride_a; p.sb = stride_b; p.sc = stride_c; p.sbias = stride_bias;
  dim3 grid(ceil_div(N, BN), ceil_div(M, BM), batch * p.splits);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return src16 ? launch_gemm_bf16<1>(a_kcontig, b_kcontig, p, grid, s) : launch_gemm_bf16<0>(a_kcontig, b_kcontig, p, grid, s);
}
