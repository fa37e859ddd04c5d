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
#include <stdlib.h>

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


// =====================================================================================================================
// Ring kernel (round 3, second half): 256 x 128 x 64 tiles, 8 waves of 64 x 64, ONE workgroup per CU, bf16 operands in HBM.
//
// The kernel above is a register-staged pipeline: its stage timers put a k tile at ~3 000 cycles for 512 cycles of MFMA (issue
// of the loads and reads, the reads landing, the MFMAs, the wait for the next operands + their LDS stores, the barrier: every
// phase waits for the one before it).  This one is the f32 GEMM's recipe (gemm.hip::gemm_dma_body) on bf16 tiles:
//   * operands go HBM/L2 -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`: no staging registers, no ds_write, the k advance in
//     the instruction's SCALAR offset, rows / columns outside the operand get an offset beyond num_records and arrive as zeros);
//   * a ring of three stages (3 x 48 KB): tile t + 2 is requested while tile t is multiplied, `s_waitcnt vmcnt(6)` (the six
//     DMA instructions of the younger tile stay in flight), one raw `s_barrier` per k tile;
//   * the DMA fixes where a lane's 16 bytes land (base + 16 lane), so the LDS images are unpadded and the SOURCE addresses carry
//     the swizzle that keeps the operand reads conflict-free under the hardware's lane groups (MI355X_MICROARCH.md, LDS table):
//       [row][64 k]  (k-contiguous, 128-byte rows): 16-byte chunk c of row r sits at position c ^ ((r >> 1) & 7) -- the 16 lanes
//                    of a ds_read_b128 group (rows 0-3, 12-15 at chunk g, rows 4-11 at chunk g + 1) cover 16 different 16-byte
//                    slots of the 256-byte bank row;
//       [64 k][rows] (k-major, 512- or 256-byte lines): 32-byte segment s of line k sits at s ^ h(k), h = (k & 3) | ((k >> 3) & 1) << 2
//                    -- the 32 lanes of a ds_read_b64_tr_b16 pass read 32-byte pieces of 8 lines (k = 8 g + q, g in {0,1} or {2,3}):
//                    8 different h.
// K must be a multiple of 64 (the caller falls back to the kernel above otherwise); M and N are free (zeros / guarded stores).
namespace ring {

constexpr int RBM = 256, RBN = 128, RBK = 64, RST = 3;
constexpr int A_BYTES = RBM * RBK * 2, B_BYTES = RBN * RBK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
constexpr int LDS_BYTES = RST * STAGE_BYTES;  // 144 KB
constexpr int LDC = RBN + 4;                  // f32 staging image of the output tile (plain stores): 256 x 132 x 4 = 132 KB
static_assert(RBM * LDC * 4 <= LDS_BYTES, "the output tile is staged in the ring");

typedef int i32x4 __attribute__((ext_vector_type(4)));

// 64 lanes x 16 bytes from buffer offsets voff (per lane) + soff (scalar) to LDS bytes [lds_addr, lds_addr + 1024), lane l at
// lds_addr + 16 l.  Inline asm on purpose (ss_common.h::ss_dma16): hipcc does not count it, so no barrier drains it; the
// consumer waits with a counted s_waitcnt vmcnt in front of its barrier.
__device__ __forceinline__ void buf_dma16(i32x4 rs, unsigned voff, int soff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(rs), "s"(soff), "s"(lds_addr)
               : "memory");
}
template <int N>
__device__ __forceinline__ void vmcnt_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void raw_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// One operand's share of a k tile for one wave: NL pieces of 1 KB (piece j = wave + 8 n of the tile's ROWS / 8).
//   KC = 1: piece j = rows 8 j .. 8 j + 7 (128 bytes each); lane l -> row 8 j + (l >> 3), position l & 7.
//   KC = 0: piece j = KL = 512 / ROWS k lines; lane l -> line KL j + l / (64 / KL), position l % (64 / KL).
template <int ROWS, int KC>
struct Operand {
  static constexpr int NL = ROWS / 64;
  static constexpr int KL = 512 / ROWS;      // KC = 0: k lines per piece (A: 2, B: 4)
  static constexpr int LPL = 64 / KL;        // KC = 0: lanes (16-byte chunks) per k line
  i32x4 rs;
  unsigned voff[NL];   // identity map: complete per-lane offset; remapped k lines: the column part only
  unsigned roff[NL];   // remapped k lines: byte offset of the piece's current storage row, carried from tile to tile with 32-bit adds
  int rem[NL];         //                   (the first version redid (k / group) * gstride + k % group + off with a 64-bit multiply
  int soff, step, ld, group, gstride, off, dm;  //                 per piece and tile: the weight gradients of W_hh ran 1.23 us per k tile against 0.9)
  unsigned c0, c1;
  bool ident;
  __device__ __forceinline__ void init(const bf16_t* src, int ld_, int group_, int gstride_, int off_, int row0, int rows, int k0, int wv,
                                       int lane) {
    ld = ld_; group = group_; gstride = gstride_; off = off_;
    const uintptr_t sa_ = reinterpret_cast<uintptr_t>(src);  // raw buffer: base, stride 0, num_records 2^31 - 1 bytes
    rs = i32x4{(int)(unsigned)sa_, (int)((sa_ >> 32) & 0xffffu), 0x7FFFFFFF, 0x00020000};
    ident = KC || (group == 0x7FFFFFFF && off == 0);
    dm = RBK % group;
    c0 = 2u * (unsigned)ld * (unsigned)((RBK / group) * gstride + dm);  // row(k + 64) - row(k) without a group boundary beyond the whole ones
    c1 = 2u * (unsigned)ld * (unsigned)(gstride - group);               // ... one more boundary
#pragma unroll
    for (int n = 0; n < NL; ++n) {
      const int j = wv + 8 * n;
      if (KC) {
        const int r = row0 + 8 * j + (lane >> 3), c = (lane & 7) ^ ((4 * j + (lane >> 4)) & 7);
        voff[n] = r < rows ? (unsigned)((remap_row(r, group, gstride, off) * ld + 8 * c) * 2) : OOB;
        roff[n] = 0u; rem[n] = 0;
      } else {
        const int kk = KL * j + lane / LPL, h = (kk & 3) | (((kk >> 3) & 1) << 2);
        const int c = (lane % LPL) ^ (2 * h), col = row0 + 8 * c;
        voff[n] = col < rows ? (unsigned)(2 * col) + (ident ? (unsigned)(2 * kk * ld) : 0u) : OOB;
        rem[n] = (k0 + kk) % group;
        roff[n] = ident ? 0u : 2u * (unsigned)(((long)((k0 + kk) / group) * gstride + rem[n] + off) * ld);
      }
    }
    soff = __builtin_amdgcn_readfirstlane(KC ? 2 * k0 : (ident ? 2 * k0 * ld : 0));
    step = __builtin_amdgcn_readfirstlane(KC ? 2 * RBK : 2 * RBK * ld);
  }
  // request piece n of the current k tile into the operand's image at LDS byte address `img` (wave-uniform); next(): step to the
  // following tile once all NL pieces are out
  __device__ __forceinline__ void piece(int n, unsigned img, int wv) {
    unsigned o = voff[n];
    if (!KC && !ident) {  // wave-uniform
      if (o != OOB) o += roff[n];
      roff[n] += c0; rem[n] += dm;
      if (rem[n] >= group) { rem[n] -= group; roff[n] += c1; }
    }
    buf_dma16(rs, o, soff, __builtin_amdgcn_readfirstlane(img + (unsigned)(wv + 8 * n) * 1024u));
  }
  __device__ __forceinline__ void next() {
    if (ident) soff += step;
  }
  __device__ __forceinline__ void issue(unsigned img, int wv) {
#pragma unroll
    for (int n = 0; n < NL; ++n) piece(n, img, wv);
    next();
  }
};

// this lane's operand read offsets (bytes inside the operand's image); everything else of an address is an immediate
template <int ROWS, int KC>
struct Frag {
  unsigned o[4];  // KC = 1: o[k2] (k2 = 0, 1) for row tile 0; KC = 0: o[t] for row tile t at k2 = 0, low half
  static constexpr int LINE = ROWS * 2;
  __device__ __forceinline__ void init(int wbase, int lane) {
    const int g = lane >> 4, li = lane & 15;
    if (KC) {
      const int f = (li >> 1) & 7;
      o[0] = (unsigned)((wbase + li) * 128 + 16 * (g ^ f));
      o[1] = (unsigned)((wbase + li) * 128 + 16 * ((4 + g) ^ f));
      o[2] = o[3] = 0;
    } else {
      const int q = (lane >> 2) & 3, pp = lane & 3, h = q | ((g & 1) << 2), s0 = (wbase >> 4) ^ (h & 4);
#pragma unroll
      for (int t = 0; t < 4; ++t) o[t] = (unsigned)((8 * g + q) * LINE + 32 * (s0 + (t ^ (h & 3))) + 8 * pp);
    }
  }
  __device__ __forceinline__ s16x8 read(const char* img, int t, int k2) const {
    if (KC) return *reinterpret_cast<const s16x8*>(img + o[k2] + 2048 * t);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const char* a0 = img + o[t] + 32 * k2 * LINE;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * LINE));
    return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
};

// The k loop of one output tile: acc += A[m0.., k tiles kt0 .. kt0 + nk) * B[.., n0..] of the operand pair p.A + ea * sa, p.B + eb * sb.
// Called by all 512 threads; returns behind a barrier that follows the last LDS read (the ring can be refilled at once).
//
// Ping-pong: waves 0-3 (`early`) and 4-7 (`late`) -- one of each per SIMD -- run the SAME program one barrier apart.  A tile
// is two phases with a barrier after each: [read tile t's fragments, request tile t + 2, wait for the own share of tile t + 1]
// and [multiply tile t]; while one group multiplies, the other reads and requests.  (First version: all eight waves read,
// requested and multiplied in lockstep -- the barrier starts them together -- and the three costs simply added up: 120 tiles
// of a d W_ih slice 62 us with neither DMA nor MFMA, +25 us of DMA issue, +41 us of MFMA = 129 us.  In-process A/B of the
// stagger, tools/gemm_bf16_bench.py diag: input projection 60 against 70 - 85 us, d layer_in 52 / 59, weight gradient 131 / 133.)
// Hazards: tile t is read by the early group in phase A(t) and by the late group in phase B(t); tile t + 2 goes into the
// stage of tile t - 1 (last read in phase B(t - 1)) and is requested in A(t) / B(t); a wave waits for its share of tile
// t + 1 (requested a whole tile earlier) before the barrier that ends its read phase, i.e. before A(t + 1) starts.
template <int AKC, int BKC>
__device__ __forceinline__ void ring_mainloop(const GemmBfParams& p, long ea, int m0, int n0, int kt0, int nk, char* rlds, unsigned lds0,
                                              f32x4 (&acc)[4][4]) {
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wv >> 1) * 64, wn = (wv & 1) * 64;
  Operand<RBM, AKC> da;
  Operand<RBN, BKC> db;
  Frag<RBM, AKC> fa_;
  Frag<RBN, BKC> fb_;
  fa_.init(wm, lane);
  fb_.init(wn, lane);
  constexpr int LPW = Operand<RBM, AKC>::NL + Operand<RBN, BKC>::NL;  // DMA instructions per wave and k tile (6)
  auto issue = [&](int buf) {
    da.issue(lds0 + (unsigned)(buf * STAGE_BYTES), wv);
    db.issue(lds0 + (unsigned)(buf * STAGE_BYTES + A_BYTES), wv);
  };
  s16x8 fa[2][4], fb[2][4];
  const bool late = wv >= 4 && !(p.flags & 1024);  // wave-uniform (1024: diagnostic, no stagger)
  // the six DMA requests of a wave are spread between its fragment reads: a request holds the wave's issue for as long as the
  // CU's address path is busy with the other waves' requests (48 KB per tile at 64 B/clk), the reads issued before it land meanwhile
  auto load = [&](const char* st, int refill) {
    const char* As = st;
    const char* Bs = st + A_BYTES;
    const bool dma = refill >= 0 && !(p.flags & 256);
    const unsigned ia = lds0 + (unsigned)(refill * STAGE_BYTES), ib = ia + A_BYTES;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k2 = u >> 2, t = u & 3;
      fa[k2][t] = fa_.read(As, t, k2);
      fb[k2][t] = fb_.read(Bs, t, k2);
      if (dma) {
        if (u == 0) da.piece(0, ia, wv);
        if (u == 1) da.piece(1, ia, wv);
        if (u == 2) db.piece(0, ib, wv);
        if (u == 3) da.piece(2, ia, wv);
        if (u == 4) da.piece(3, ia, wv);
        if (u == 5) db.piece(1, ib, wv);
      }
    }
    if (dma) {
      da.next();
      db.next();
    }
  };
  auto mult = [&]() {
    if (p.flags & 512) {  // diagnostic: no MFMAs
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int a = 0; a < 4; ++a) asm volatile("" ::"v"(fa[k2][a]), "v"(fb[k2][a]));
      return;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = mfma_bf16(fa[k2][a], fb[k2][b], acc[a][b]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto phase_barrier = [&]() {  // (register-only MFMAs would move across a bare barrier)
    __builtin_amdgcn_sched_barrier(0);
    raw_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  da.init(static_cast<const bf16_t*>(p.A) + ea * p.sa, p.lda, p.a_group, p.a_gstride, p.a_off, m0, p.M, kt0 * RBK, wv, lane);
  db.init(static_cast<const bf16_t*>(p.B) + ea * p.sb, p.ldb, p.b_group, p.b_gstride, p.b_off, n0, p.N, kt0 * RBK, wv, lane);
  if (nk > 0) issue(0);
  if (nk > 1) {
    issue(1);
    vmcnt_wait<LPW>();
  } else {
    vmcnt_wait<0>();
  }
  phase_barrier();  // tile 0 is in LDS
  if (late) phase_barrier();
  for (int t0 = 0; t0 < nk; t0 += RST) {
#pragma unroll
    for (int s_ = 0; s_ < RST; ++s_) {
      const int t = t0 + s_;
      if (t < nk) {
        load(rlds + s_ * STAGE_BYTES, t + 2 < nk ? (s_ + 2) % RST : -1);
        if (t + 2 < nk) vmcnt_wait<LPW>();  // tile t + 1: at most tile t + 2's LPW instructions of this wave stay in flight
        else vmcnt_wait<0>();
        phase_barrier();
        mult();
        phase_barrier();
      }
    }
  }
  if (!late) phase_barrier();
}

// workgroups are dealt to the 8 XCDs round-robin: give every XCD a contiguous range of work (neighbours share operand panels in
// its L2); bijective for any grid size
__device__ __forceinline__ int xcd_contiguous_id() {
  const int nwg = gridDim.x, id = blockIdx.x, xcd = id & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
}

// acc (D row = 4 g + r, column = li of the wave's 64 x 64 sub-tile) -> C.  Called behind the main loop (no DMA in flight, every LDS
// read done): a tile whose 128 columns are all there goes through an LDS image so that a row leaves as one 512-byte run.
__device__ __forceinline__ void ring_epilogue(const GemmBfParams& p, f32x4 (&acc)[4][4], float* C, const float* bias, int m0, int n0,
                                              bool add_bias, bool atomic, char* rlds) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, li = lane & 15;
  const int wm = (wv >> 1) * 64, wn = (wv & 1) * 64;
  const bool accumulate = p.flags & 1;
  if (!add_bias) bias = nullptr;
  if (!atomic && n0 + RBN <= p.N && (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0) {
    float* stage = reinterpret_cast<float*>(rlds);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = wn + 16 * b + li;
      const float bv = bias ? bias[n0 + col] : 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) stage[(wm + 16 * a + 4 * g + r) * LDC + col] = acc[a][b][r] + bv;
    }
    __syncthreads();
    const int c4 = (tid & 31) * 4, r0 = tid >> 5;
#pragma unroll 4
    for (int it = 0; it < RBM / 16; ++it) {
      const int row = r0 + 16 * it;
      if (m0 + row < p.M) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&stage[row * LDC + c4]);
        f32x4* dst = reinterpret_cast<f32x4*>(C + (long)(m0 + row) * p.ldc + n0 + c4);
        if (accumulate) v += *dst;
        *dst = v;
      }
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int n = n0 + wn + 16 * b + li;
    if (n >= p.N) continue;
    const float bv = bias ? bias[n] : 0.f;
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
}

template <int AKC, int BKC>
__global__ __launch_bounds__(512, 1) void gemm_bf16_ring_kernel(GemmBfParams p, int gx, int gy, int kcat) {
  extern __shared__ __attribute__((aligned(16))) char rlds[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)rlds;
  const int wg = (p.flags & 2048) ? (int)blockIdx.x : xcd_contiguous_id();  // (2048: diagnostic, tiles as dispatched)
  const int bx = wg % gx, by = (wg / gx) % gy, bz = wg / (gx * gy);
  const int batch = bz / p.splits, split = bz % p.splits;
  const int m0 = by * RBM, n0 = bx * RBN;
  const int nkt = p.K / RBK;
  const int per = (nkt + p.splits - 1) / p.splits;
  const int kt0 = split * per, kt1 = min(nkt, kt0 + per), nk = max(0, kt1 - kt0);

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // kcat > 1: that many (A, B) pairs, strides sa / sb apart, are multiplied into ONE C (d layer_in = dG_f W_f + dG_r W_r
  // without atomics or a cleared C); otherwise the pair of this batch entry
  for (int kc = 0; kc < kcat; ++kc) ring_mainloop<AKC, BKC>(p, kcat > 1 ? kc : batch, m0, n0, kt0, nk, rlds, lds0, acc);

  // ---- epilogue
  ring_epilogue(p, acc, p.C + (kcat > 1 ? 0 : batch * p.sc), p.bias ? p.bias + batch * p.sbias : nullptr, m0, n0, split == 0,
                (p.flags & 4) || p.splits > 1, rlds);
}

template <int AKC, int BKC>
int launch_ring(const GemmBfParams& p, int batch, int kcat, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_ring_kernel<AKC, BKC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS_BYTES) != hipSuccess)
      return SS_ERR_LAUNCH;
    attr_set = true;
  }
  const int gx = ceil_div(p.N, RBN), gy = ceil_div(p.M, RBM);
  const int nz = (kcat > 1 ? 1 : batch) * p.splits;
  hipLaunchKernelGGL((gemm_bf16_ring_kernel<AKC, BKC>), dim3((unsigned)(gx * gy * nz)), dim3(512), LDS_BYTES, s, p, gx, gy, kcat);
  return ss_launch_status();
}

// ---- Weight gradients: several k-major problems (tiny outputs, K = B T) in ONE launch, K spread evenly over the chip.
// The work of the group is the list of its (problem, batch entry, output tile, k tile) units in that order; workgroup w takes
// units [w U, (w + 1) U), U = ceil(units / workgroups): a contiguous run of k tiles of one output tile, crossing into the next
// tile at most a few times ("stream-K").  Every run leaves its raw accumulators (16 x 16 bytes per lane, coalesced) in a slab of
// the workspace -- slot (tile, w - first workgroup of the tile) -- and a second launch adds each tile's slabs into the gradient.
// Why not K slices + float atomics (the form the register-staged kernel keeps): the atomic epilogue of a 256 x 128 tile took
// 27 us per workgroup, as long as 30 k tiles (device-scope float atomics are performed beyond the XCD's L2), and whole slices
// per workgroup quantise badly over 256 CUs (144 tiles x 120 k tiles: 1, 2 or 3 slices all end near 80 - 120 tile times for
// an ideal of 67.5).
constexpr int GROUP_MAX = 8;
constexpr int SLAB_FLOATS = RBM * RBN;
struct RingGroup {
  GemmBfParams p[GROUP_MAX];
  int gx[GROUP_MAX], gy[GROUP_MAX], batch[GROUP_MAX], nkt[GROUP_MAX];
  int ubase[GROUP_MAX + 1];  // first unit of problem j
  int tbase[GROUP_MAX + 1];  // first tile of problem j
  int n, U, maxc;            // units per workgroup; slots per tile
  int whole;                 // 1: one workgroup per output tile, all of K, C += straight from the registers (no slabs, no reduce)
  float* ws;
};

__global__ __launch_bounds__(512, 1) void gemm_bf16_ring_group_kernel(RingGroup gg) {
  extern __shared__ __attribute__((aligned(16))) char rlds[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)rlds;
  const int tid = threadIdx.x;
  const int w = xcd_contiguous_id();
  if (gg.whole) {
    // Enough output tiles to fill the chip: workgroup w = tile w, whole K.  Neighbouring tiles (one XCD: xcd_contiguous_id) walk K
    // together and share their operand panels in that XCD's L2 -- the K ranges of the stream-K form below are staggered, every
    // workgroup then streams its own 3 MB from beyond L2 (845 MB per launch for 95 MB of operands: bound by exactly that).
    int j = 0;
#pragma unroll
    for (int q = 1; q < GROUP_MAX; ++q)
      if (q < gg.n && w >= gg.tbase[q]) j = q;
    GemmBfParams p = gg.p[0];
    int gx = gg.gx[0], gy = gg.gy[0], nkt = gg.nkt[0], tb = gg.tbase[0];
#pragma unroll
    for (int q = 1; q < GROUP_MAX; ++q)
      if (q == j) { p = gg.p[q]; gx = gg.gx[q]; gy = gg.gy[q]; nkt = gg.nkt[q]; tb = gg.tbase[q]; }
    const int tile = w - tb, bx = tile % gx, by = (tile / gx) % gy, bi = tile / (gx * gy);
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    ring_mainloop<0, 0>(p, bi, by * RBM, bx * RBN, 0, nkt, rlds, lds0, acc);
    ring_epilogue(p, acc, p.C + bi * p.sc, nullptr, by * RBM, bx * RBN, false, false, rlds);
    return;
  }
  int u = w * gg.U;
  const int u1 = min(gg.ubase[gg.n], u + gg.U);
  while (u < u1) {  // wave-uniform
    int j = 0;
#pragma unroll
    for (int q = 1; q < GROUP_MAX; ++q)
      if (q < gg.n && u >= gg.ubase[q]) j = q;
    // (a dynamically indexed kernel-argument struct would go through scratch: select the problem's fields one by one)
    GemmBfParams p = gg.p[0];
    int gx = gg.gx[0], gy = gg.gy[0], nkt = gg.nkt[0], ub = gg.ubase[0], tb = gg.tbase[0];
#pragma unroll
    for (int q = 1; q < GROUP_MAX; ++q)
      if (q == j) { p = gg.p[q]; gx = gg.gx[q]; gy = gg.gy[q]; nkt = gg.nkt[q]; ub = gg.ubase[q]; tb = gg.tbase[q]; }
    const int local = u - ub, tile = local / nkt, kt0 = local - tile * nkt;
    const int nk = min(nkt - kt0, u1 - u);
    const int bx = tile % gx, by = (tile / gx) % gy, bi = tile / (gx * gy);
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    ring_mainloop<0, 0>(p, bi, by * RBM, bx * RBN, kt0, nk, rlds, lds0, acc);
    const int w_lo = (ub + tile * nkt) / gg.U;  // first workgroup with a unit of this tile
    f32x4* slab = reinterpret_cast<f32x4*>(gg.ws + ((long)(tb + tile) * gg.maxc + (w - w_lo)) * SLAB_FLOATS) + tid;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) slab[(a * 4 + b) * 512] = acc[a][b];
    u += nk;
  }
}

// C[tile] += the tile's slabs.  One thread owns one accumulator quad (a, b, lane) of one output tile.
__global__ __launch_bounds__(512) void gemm_bf16_ring_group_reduce_kernel(RingGroup gg) {
  const int q = blockIdx.x & 15, gt = blockIdx.x >> 4;
  int j = 0;
#pragma unroll
  for (int k = 1; k < GROUP_MAX; ++k)
    if (k < gg.n && gt >= gg.tbase[k]) j = k;
  const GemmBfParams& p = gg.p[j];
  const int tile = gt - gg.tbase[j], gx = gg.gx[j], gy = gg.gy[j], nkt = gg.nkt[j];
  const int bx = tile % gx, by = (tile / gx) % gy, bi = tile / (gx * gy);
  const int ustart = gg.ubase[j] + tile * nkt;
  const int cnt = (ustart + nkt - 1) / gg.U - ustart / gg.U + 1;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, li = lane & 15;
  const f32x4* src = reinterpret_cast<const f32x4*>(gg.ws + (long)gt * gg.maxc * SLAB_FLOATS) + q * 512 + tid;
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < cnt; ++c) sum += src[(long)c * (SLAB_FLOATS / 4)];
  const int a = q >> 2, b = q & 3;
  const int col = bx * RBN + (wv & 1) * 64 + 16 * b + li;
  if (col >= p.N) return;
  float* C = p.C + bi * p.sc;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = by * RBM + (wv >> 1) * 64 + 16 * a + 4 * g + r;
    if (row < p.M) C[(long)row * p.ldc + col] += sum[r];
  }
}

}  // namespace ring

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

// ---- grouped weight-gradient GEMMs (ring kernel, stream-K over the group; see gemm_bf16_ring_group_kernel)
static const bool ss_gemm_bf16_group_split = getenv("SS_GEMM_BF16_GROUP_SPLIT") != nullptr;  // diagnostic: always the stream-K form
static int ring_group_cus() { return ss_device_cus(); }
static int ring_group_prepare(const ss_gemm_problem* pr, int n, float* ws, ring::RingGroup* out, int* wgs, long* floats) {
  SS_REQUIRE(pr && n >= 1 && n <= ring::GROUP_MAX, SS_ERR_ARG);
  ring::RingGroup& gg = *out;
  gg.n = n;
  gg.ws = ws;
  gg.ubase[0] = gg.tbase[0] = 0;
  int nkt_max = 1;
  for (int j = 0; j < n; ++j) {
    const ss_gemm_problem& q = pr[j];
    SS_REQUIRE(q.A && q.B && q.C && q.M > 0 && q.N > 0 && q.K > 0 && q.batch > 0 && q.a_group > 0 && q.b_group > 0, SS_ERR_ARG);
    SS_REQUIRE(!q.a_kcontig && !q.b_kcontig && q.K % ring::RBK == 0, SS_ERR_UNSUPPORTED);
    SS_REQUIRE(q.lda % 8 == 0 && q.ldb % 8 == 0 && q.stride_a % 8 == 0 && q.stride_b % 8 == 0 && q.M <= q.lda && q.N <= q.ldb, SS_ERR_ARG);
    SS_REQUIRE((reinterpret_cast<uintptr_t>(q.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(q.B) & 15) == 0 && q.ldc >= q.N, SS_ERR_ARG);
    GemmBfParams& p = gg.p[j];
    p.M = q.M; p.N = q.N; p.K = q.K;
    p.A = q.A; p.lda = q.lda; p.a_group = q.a_group; p.a_gstride = q.a_gstride; p.a_off = q.a_off;
    p.B = q.B; p.ldb = q.ldb; p.b_group = q.b_group; p.b_gstride = q.b_gstride; p.b_off = q.b_off;
    p.C = q.C; p.ldc = q.ldc; p.bias = nullptr; p.flags = 8 | 1; p.splits = 1;
    p.sa = q.stride_a; p.sb = q.stride_b; p.sc = q.stride_c; p.sbias = 0;
    gg.gx[j] = ceil_div(q.N, ring::RBN); gg.gy[j] = ceil_div(q.M, ring::RBM); gg.batch[j] = q.batch; gg.nkt[j] = q.K / ring::RBK;
    const int tiles = gg.gx[j] * gg.gy[j] * q.batch;
    gg.tbase[j + 1] = gg.tbase[j] + tiles;
    gg.ubase[j + 1] = gg.ubase[j] + tiles * gg.nkt[j];
    if (gg.nkt[j] > nkt_max) nkt_max = gg.nkt[j];
  }
  for (int j = n; j < ring::GROUP_MAX; ++j) { gg.ubase[j + 1] = gg.ubase[n]; gg.tbase[j + 1] = gg.tbase[n]; gg.gx[j] = gg.gy[j] = gg.nkt[j] = 1; }
  const int units = gg.ubase[n];
  const int W = units < ring_group_cus() ? units : ring_group_cus();
  gg.U = ceil_div(units, W);
  gg.maxc = (nkt_max - 1) / gg.U + 2;
  *wgs = ceil_div(units, gg.U);
  *floats = (long)gg.tbase[n] * gg.maxc * ring::SLAB_FLOATS;
  // at least half a chip of output tiles (and no more than a chip): one workgroup per tile, no K split
  gg.whole = (2 * gg.tbase[n] >= ring_group_cus() && gg.tbase[n] <= ring_group_cus() && !ss_gemm_bf16_group_split) ? 1 : 0;
  if (gg.whole) *wgs = gg.tbase[n];
  return SS_OK;
}

extern "C" int ss_gemm_bf16_splitk_group_ws_floats(const ss_gemm_problem* problems, int n, long* floats) {
  SS_REQUIRE(floats, SS_ERR_ARG);
  ring::RingGroup gg;
  int wgs = 0;
  return ring_group_prepare(problems, n, nullptr, &gg, &wgs, floats);
}

extern "C" int ss_gemm_bf16_splitk_group(const ss_gemm_problem* problems, int n, float* ws, long ws_floats, ss_stream_t stream) {
  SS_REQUIRE(ws && (reinterpret_cast<uintptr_t>(ws) & 15) == 0, SS_ERR_ARG);
  ring::RingGroup gg;
  int wgs = 0;
  long floats = 0;
  const int st = ring_group_prepare(problems, n, ws, &gg, &wgs, &floats);
  if (st != SS_OK) return st;
  SS_REQUIRE(gg.whole || ws_floats >= floats, SS_ERR_ARG);  // (one workgroup per tile over all of K stores no slabs)
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(ring::gemm_bf16_ring_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            ring::LDS_BYTES) != hipSuccess)
      return SS_ERR_LAUNCH;
    attr_set = true;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(ring::gemm_bf16_ring_group_kernel, dim3((unsigned)wgs), dim3(512), ring::LDS_BYTES, s, gg);
  if (ss_launch_status() != SS_OK) return SS_ERR_LAUNCH;
  if (gg.whole) return SS_OK;
  hipLaunchKernelGGL(ring::gemm_bf16_ring_group_reduce_kernel, dim3((unsigned)(gg.tbase[n] * 16)), dim3(512), 0, s, gg);
  return ss_launch_status();
}

static const bool ss_gemm_bf16_no_ring = getenv("SS_GEMM_BF16_NO_RING") != nullptr;

extern "C" int ss_gemm_bf16_batched(int a_kcontig, int b_kcontig, int M, int N, int K, const void* A, int lda, int a_group,
                                    int a_gstride, int a_off, const void* B, int ldb, int b_group, int b_gstride, int b_off,
                                    float* C, int ldc, const float* bias, int flags, int splits, int batch, long stride_a,
                                    long stride_b, long stride_c, long stride_bias, ss_stream_t stream) {
  SS_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0 && splits > 0, SS_ERR_ARG);
  SS_REQUIRE(a_group > 0 && b_group > 0, SS_ERR_ARG);
  // bit0 accumulate, bit2 atomics, bit3 operands are bf16 in HBM, bit4 the `batch` (A, B) pairs are summed into ONE C (K concatenated)
  SS_REQUIRE(!(flags & ~(29 | 768 | 1024 | 2048)), SS_ERR_UNSUPPORTED);  // (256 / 512: diagnostic)
  SS_REQUIRE(!(flags & 16) || ((flags & 8) && K % 64 == 0 && splits == 1 && !(flags & 4)), SS_ERR_UNSUPPORTED);
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
  // bf16 operands, whole 64-deep k tiles, an output worth a 256 x 128 tile: the ring kernel (SS_GEMM_BF16_NO_RING: diagnostic)
  if (src16 && K % 64 == 0 && (M >= 128 || (flags & 16)) && !ss_gemm_bf16_no_ring) {
    const int kcat = (flags & 16) ? batch : 1;
    if (a_kcontig && b_kcontig) return ring::launch_ring<1, 1>(p, batch, kcat, s);
    if (a_kcontig) return ring::launch_ring<1, 0>(p, batch, kcat, s);
    if (b_kcontig) return ring::launch_ring<0, 1>(p, batch, kcat, s);
    return ring::launch_ring<0, 0>(p, batch, kcat, s);
  }
  return src16 ? launch_gemm_bf16<1>(a_kcontig, b_kcontig, p, grid, s) : launch_gemm_bf16<0>(a_kcontig, b_kcontig, p, grid, s);
}
