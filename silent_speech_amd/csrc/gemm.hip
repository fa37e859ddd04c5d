// f32 MFMA GEMM for the dense contractions of the path: GRU input projections, their data/weight
// gradients, and the classifier head.  C[M,N] (+)= op(A)[M,K] * op(B)[K,N] (+ bias[N]) (ReLU).
//
// Replaces the aten::linear / addmm calls inside nn.GRU and nn.Linear
// (/root/reference/train_model_official.py:261-267, 271-277) and their autograd.
//
// 128x64x16 block tile, 256 threads = 4 waves as 2(M) x 2(N), wave tile 64x32 = 4x2 tiles of
// v_mfma_f32_16x16x4_f32 (exact f32).  An operand whose K index is contiguous in memory is kept [row][k] in
// LDS and read back with one ds_read_b128 per four k-steps, an operand whose M/N index is contiguous is kept
// [k][row] and read with ds_read_b32.  Inside a 16-wide k tile the MFMA slot (kk, g) carries k = 4g + kk for
// BOTH operands, which is what lets the [row][k] form use a single 16-byte read.
//
// Two mainloops share the epilogue: gemm_f32_kernel stages global -> registers -> LDS (any alignment, ragged
// shapes, column sums of A riding along: the head's small GEMMs); gemm_dma_kernel feeds a ring of four k
// tiles by LDS-DMA (every GEMM of the GRU layers; see the comment above it).  Split-K goes through a scratch
// buffer and a reduce pass; several split-K problems can share one launch (gemm_dma_group_kernel).
#include <stdlib.h>

// diagnostic build: rows of the stamp table in dispatch order over the 3-D grid
#define SS_STAMP_ROW ((int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)))
#include "ss_common.h"

STAMP_TABLE(ss_debug_stamps_gemm)

namespace {

constexpr int BM = 128, BN = 64, BK = 16;
constexpr int LDK = BK + 4;    // [row][k] image, row stride 20 floats (80 B, 16-B aligned)
constexpr int LDM_A = BM + 4;  // [k][row] image strides: == 4 (mod 8) so k rows 4 apart land 16 banks apart
constexpr int LDM_B = BN + 4;

struct RowMap {
  int G, S, off;
  __device__ __forceinline__ long operator()(int r) const {
    if (G == 0x7fffffff) return r + off;  // identity map (wave-uniform branch): no integer division on the fetch path
    return (long)(r / G) * S + (r % G) + off;
  }
};

struct GemmParams {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K;
  int lda, ldb, ldc;
  RowMap ra, rb;
  float* asum; // optional: asum[m] += sum_k A[k][m] (A stored [k][m]); the bias gradient rides on the dW GEMM
  int ksplit;  // K elements per blockIdx.z slice (multiple of BK)
  int nz;      // K slices per problem; blockIdx.z = batch * nz + slice
  long sA, sB, sC, sBias, sAsum;  // element strides between the problems of a batch
  int kcat;    // >1 (flags bit4): that many (A, B) pairs, strides sA / sB apart, are multiplied into ONE C -- the K loop runs
               // through them one after the other (d layer_in = dG_f W_f + dG_r W_r without atomics or a cleared C)
  int flags;   // bit0: accumulate into C (plain RMW when nz==1, atomics otherwise); bit1: ReLU; bit2: always atomic;
               // bit3: C is a split-K workspace -- every workgroup leaves its accumulators there as they lie in its
               // registers (16 bytes per lane, fully coalesced) and splitk_reduce_kernel folds the slices into the real C
};

// Which tile of which launch grid a workgroup works on: blockIdx / gridDim for a plain launch, looked up for a grouped one.
struct TileId { int bx, by, bz, gx, gy; };

// Shared epilogue: acc[mt][nt] of the wave's 64x32 sub-tile -> C (plain / accumulate / ReLU / float atomics) or, with
// flags bit3, into the split-K workspace as the accumulators lie in the registers.
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[4][2], int m0, int n0, int zs, int wm, int wn,
                                              int i, int g, float* stage /* BM x (BN + 4) floats of LDS, or NULL */,
                                              const TileId& id) {
  if (p.flags & 8) {
    // plain 16-byte stores of the raw accumulators instead of 32 float atomics per lane
    f32x4* w = reinterpret_cast<f32x4*>(p.C) + (((long)id.bz * id.gy + id.by) * id.gx + id.bx) * (8 * 256) + threadIdx.x;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) w[(mt * 2 + nt) * 256] = acc[mt][nt];
    return;
  }
  const bool accumulate = p.flags & 1, relu = p.flags & 2;
  const bool atomic = p.nz > 1 || (p.flags & 4);
  auto emit = [&](const f32x4& av, int mt, int nt) {
    int col = n0 + wn * 32 + nt * 16 + i;
    if (col >= p.N) return;
    float bv = (p.bias && zs == 0) ? p.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int row = m0 + wm * 64 + mt * 16 + 4 * g + r;
      if (row >= p.M) continue;
      float v = av[r] + bv;
      float* dst = p.C + (long)row * p.ldc + col;
      if (atomic) {
        atomicAdd(dst, v);
      } else {
        if (accumulate) v += *dst;
        if (relu) v = relu_f(v);
        *dst = v;
      }
    }
  };
  // Plain stores of a tile whose 64 columns are all there: through LDS, so that a row leaves as one 256-byte run (16 lanes x
  // 16 bytes) instead of 64-byte pieces from the MFMA layout -- the input projections write 70 MB per launch and were bound
  // by exactly that.
  if (stage && !atomic && n0 + BN <= p.N && (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0) {
    constexpr int LDS_C = BN + 4;
    __syncthreads();  // the last k tile has been read by everybody
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int col = wn * 32 + nt * 16 + i;
        const float bv = (p.bias && zs == 0) ? p.bias[n0 + col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) stage[(wm * 64 + mt * 16 + 4 * g + r) * LDS_C + col] = acc[mt][nt][r] + bv;
      }
    __syncthreads();
    const int c4 = (threadIdx.x & 15) * 4, r0 = threadIdx.x >> 4;
#pragma unroll
    for (int it = 0; it < BM / 16; ++it) {
      const int row = r0 + 16 * it;
      if (m0 + row < p.M) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&stage[row * LDS_C + c4]);
        f32x4* dst = reinterpret_cast<f32x4*>(p.C + (long)(m0 + row) * p.ldc + n0 + c4);
        if (accumulate) v += *dst;
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = relu_f(v[e]);
        }
        *dst = v;
      }
    }
    return;
  }
  if (p.nz > 1) {
    // split-K slices of one tile reach their epilogues together: each starts at a different row group so that their
    // atomics meet on different lines (wave-uniform rotation; costs a register-indexed read of the accumulators)
    const int rot = zs & 3;
#pragma unroll
    for (int mt_ = 0; mt_ < 4; ++mt_) {
      const int mt = (mt_ + rot) & 3;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 av = mt == 0 ? acc[0][nt] : mt == 1 ? acc[1][nt] : mt == 2 ? acc[2][nt] : acc[3][nt];
        emit(av, mt, nt);
      }
    }
  } else {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) emit(acc[mt][nt], mt, nt);
  }
}

// One operand tile, staged global -> registers (fetch) -> LDS (store) so the loads of tile k+1 are in
// flight while tile k is multiplied.  ROWS = BM or BN.
// KCONTIG: memory is [row][k] (row-major over the GEMM's M or N index) -> LDS image [row][LDK]
// else   : memory is [k][row]                                          -> LDS image [k][ROWS+4]
template <int ROWS, bool KCONTIG>
struct Stager {
  static constexpr int NV = ROWS * 4 / 256;
  f32x4 v[NV];

  __device__ __forceinline__ void fetch(const float* __restrict__ P, int ld, const RowMap& rm, int row0, int nrows, int k0,
                                        int kend, bool vec_ok) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int q = tid + it * 256;
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (KCONTIG) {
        const int r = q >> 2, kq = q & 3;  // thread -> (row, 4-wide k group)
        const int gr = row0 + r, gk = k0 + 4 * kq;
        if (gr < nrows) {
          const float* src = P + rm(gr) * ld + gk;
          if (vec_ok && gk + 3 < kend) {
            x = *reinterpret_cast<const f32x4*>(src);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (gk + e < kend) x[e] = src[e];
          }
        }
      } else {
        const int k = q / (ROWS / 4), rq = q % (ROWS / 4);  // thread -> (k row, 4-wide row group)
        const int gk = k0 + k, gr = row0 + 4 * rq;
        if (gk < kend) {
          const float* src = P + rm(gk) * ld + gr;
          if (vec_ok && gr + 3 < nrows) {
            x = *reinterpret_cast<const f32x4*>(src);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (gr + e < nrows) x[e] = src[e];
          }
        }
      }
      v[it] = x;
    }
  }

  __device__ __forceinline__ void store(float* __restrict__ lds) const {
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int q = tid + it * 256;
      if (KCONTIG) {
        *reinterpret_cast<f32x4*>(&lds[(q >> 2) * LDK + 4 * (q & 3)]) = v[it];
      } else {
        constexpr int LD = ROWS + 4;
        *reinterpret_cast<f32x4*>(&lds[(q / (ROWS / 4)) * LD + 4 * (q % (ROWS / 4))]) = v[it];
      }
    }
  }
};

constexpr int KSUB = 1;  // 16-wide k tiles per barrier (2 measured slower: the doubled LDS halves the co-resident workgroups)

template <bool A_KCONTIG, bool B_KCONTIG>
__global__ __launch_bounds__(256, 5) void gemm_f32_kernel(GemmParams p) {  // 5 waves per SIMD: <= 96 VGPRs, 5 x 29 KB of LDS per CU
  constexpr int A_SZ = A_KCONTIG ? BM * LDK : BK * LDM_A;
  constexpr int B_SZ = B_KCONTIG ? BN * LDK : BK * LDM_B;
  constexpr int STAGE = KSUB * (A_SZ + B_SZ);
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int bi = blockIdx.z / p.nz, zs = blockIdx.z - bi * p.nz;
  p.A += bi * p.sA; p.B += bi * p.sB;
  if (!(p.flags & 8)) p.C += bi * p.sC;
  if (p.bias) p.bias += bi * p.sBias;
  if (p.asum) p.asum += bi * p.sAsum;
  const int kbeg = zs * p.ksplit;
  const int kend = min(p.K, kbeg + p.ksplit);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int i = lane & 15, g = lane >> 4;

  const bool a_vec = ((p.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0);
  const bool b_vec = ((p.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.B) & 15) == 0);

  f32x4 acc[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float asum = 0.f;  // column sums of the [k][row] A tile (bias gradient), thread t < BM owns row m0 + t
  const bool want_asum = (!A_KCONTIG) && p.asum && blockIdx.x == 0;

  STAMP_ENTRY;
  STAMP_DECL;
  Stager<BM, A_KCONTIG> sa[KSUB];
  Stager<BN, B_KCONTIG> sb[KSUB];
#pragma unroll
  for (int u = 0; u < KSUB; ++u) {
    sa[u].fetch(p.A, p.lda, p.ra, m0, p.M, kbeg + u * BK, kend, a_vec);
    sb[u].fetch(p.B, p.ldb, p.rb, n0, p.N, kbeg + u * BK, kend, b_vec);
  }
#pragma unroll
  for (int u = 0; u < KSUB; ++u) {
    sa[u].store(lds + u * (A_SZ + B_SZ));
    sb[u].store(lds + u * (A_SZ + B_SZ) + A_SZ);
  }
  __syncthreads();
  STAMP(15);

  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK * KSUB) {
    const bool more = k0 + BK * KSUB < kend;
    if (more) {  // next stage's global loads fly while this one is multiplied
#pragma unroll
      for (int u = 0; u < KSUB; ++u) {
        sa[u].fetch(p.A, p.lda, p.ra, m0, p.M, k0 + (KSUB + u) * BK, kend, a_vec);
        sb[u].fetch(p.B, p.ldb, p.rb, n0, p.N, k0 + (KSUB + u) * BK, kend, b_vec);
      }
    }
#pragma unroll
    for (int u = 0; u < KSUB; ++u) {
      const float* As = lds + cur * STAGE + u * (A_SZ + B_SZ);
      const float* Bs = As + A_SZ;
      float a[4][4], b[2][4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        int r = wm * 64 + mt * 16 + i;
        if (A_KCONTIG) {
          f32x4 v = *reinterpret_cast<const f32x4*>(&As[r * LDK + 4 * g]);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) a[mt][kk] = v[kk];
        } else {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) a[mt][kk] = As[(4 * g + kk) * LDM_A + r];
        }
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        int c = wn * 32 + nt * 16 + i;
        if (B_KCONTIG) {
          f32x4 v = *reinterpret_cast<const f32x4*>(&Bs[c * LDK + 4 * g]);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) b[nt][kk] = v[kk];
        } else {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) b[nt][kk] = Bs[(4 * g + kk) * LDM_B + c];
        }
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma16(a[mt][kk], b[nt][kk], acc[mt][nt]);
      if (!A_KCONTIG) {
        if (want_asum && threadIdx.x < BM) {
#pragma unroll
          for (int k = 0; k < BK; ++k) asum += As[k * LDM_A + threadIdx.x];
        }
      }
    }
    STAMP(1);
    if (more) {
      float* nxt = lds + (cur ^ 1) * STAGE;
#pragma unroll
      for (int u = 0; u < KSUB; ++u) {
        sa[u].store(nxt + u * (A_SZ + B_SZ));
        sb[u].store(nxt + u * (A_SZ + B_SZ) + A_SZ);
      }
    }
    STAMP(2);
    __syncthreads();
    STAMP(3);
    cur ^= 1;
  }

  if (!A_KCONTIG) {
    if (want_asum && threadIdx.x < BM && m0 + (int)threadIdx.x < p.M) atomicAdd(&p.asum[m0 + threadIdx.x], asum);
  }

  gemm_epilogue(p, acc, m0, n0, zs, wm, wn, i, g, nullptr,
                TileId{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.y});
  STAMP(4);
  STAMP_FLUSH();
}

// ---------------------------------------------------------------------------------------------------------------------
// The same 128x64x16 tile fed by LDS-DMA through a ring of DSTAGES k tiles.
//
// Measured on the shapes of the path (K slices of 16..120 k tiles, 1..3 workgroups per CU): the register-staged kernel
// above spends ~2800 cycles per k tile on a CU it has to itself -- one full memory latency per tile, prefetch distance 1 --
// against 1024 cycles of MFMA work.  Here global_load_lds_dwordx4 writes the tiles straight into LDS, DSTAGES - 1 tiles
// ahead, with no VGPRs and no LDS store instruction on the way; the wave waits with vmcnt(3 * (DSTAGES - 2)), so only
// the oldest tile in flight has to have landed.
//
// The DMA puts lane l's 16 bytes at (wave-uniform base) + 16 l, so the LDS image is fixed and the SOURCE addresses carry the
// swizzle that keeps the MFMA operand reads conflict-free without padding:
//   [k][row] operand  : image [16 k][ROWS] floats; 16-byte unit u of k row k holds row unit (u - 4 * ((k >> 2) & 1)):
//                       the ds_read_b32 of lane groups g and g + 1 (k rows 4 apart) land 16 banks apart
//   [row][k] operand  : image [ROWS][16 k]; unit s of row r holds k unit s ^ kc_swz(r), kc_swz(r) = (-(r >> 2)) & 3.  A
//                       ds_read_b128 is served in groups of 16 lanes that are NOT 16 consecutive rows of one k unit: rows 0-3 and
//                       12-15 at unit g together with rows 4-11 at unit g ^ 1 (MI355X_MICROARCH.md, LDS table).  The rows with equal
//                       r & 3 share four 16-byte slots of the 256-byte bank row, so the group is conflict-free iff
//                       {f(0), f(3), 1 ^ f(1), 1 ^ f(2)} are distinct (f = swizzle of r >> 2): f = (0, 3, 2, 1).  Rounds 1-2 used
//                       f = (0, 1, 2, 3), right for 16 consecutive lanes and 2-way conflicted on the hardware's groups
//                       (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.46 / 0.33 in the input-projection / d layer_in GEMMs).
// Rows past M / N are clamped to the last valid ones (they only feed accumulators nobody stores); a ragged last k tile goes
// through registers with zero fill.
__device__ __forceinline__ constexpr int kc_swz(int r) { return (0 - (r >> 2)) & 3; }
constexpr int DSTAGES = 4;
constexpr int D_A = BM * BK, D_B = BN * BK, D_STAGE = D_A + D_B;  // floats per stage: 12 KB

template <int ROWS, bool KC>
struct DmaOperand {
  static constexpr int U = ROWS / 4;    // 16-byte units per k row of the [k][row] image
  static constexpr int NL = ROWS / 64;  // 1 KB wave loads per wave and k tile
  // Running source pointer of this lane's slot.  linear (wave-uniform): the next tile is `step` floats further on -- always
  // for a [row][k] operand, and for a [k][row] operand whose k rows are stored in order.  Otherwise (the (b,t) -> (b,t-1)
  // pairing of d W_hh) the storage k row of the current tile is q * rm.S + r (+ rm.off, folded into src), r < rm.G.
  const float* src[NL];
  int q[NL], r[NL];
  RowMap rm;
  long step;
  int ld;
  bool linear;

  __device__ __forceinline__ void init(const float* P, int ld_, const RowMap& rm_, int row0, int nrows, int kbeg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    rm = rm_;
    ld = ld_;
    linear = KC || rm.G == 0x7fffffff;
    step = KC ? BK : (long)BK * ld;
#pragma unroll
    for (int n = 0; n < NL; ++n) {
      const int sl = 64 * (wave + 4 * n) + lane;  // this lane's 16-byte slot of the stage image
      if (KC) {
        const int rr = sl >> 2, ku = (sl & 3) ^ kc_swz(rr);
        const int gr = min(row0 + rr, nrows - 1);
        src[n] = P + rm(gr) * ld + kbeg + 4 * ku;
        q[n] = r[n] = 0;
      } else {
        const int k = sl / U, u = sl % U, ru = (u - 4 * ((k >> 2) & 1)) & (U - 1);
        const int gk = kbeg + k;
        src[n] = P + min(row0 + 4 * ru, nrows - 4) + (long)rm.off * ld;
        q[n] = linear ? 0 : gk / rm.G;
        r[n] = gk - q[n] * rm.G;
        if (linear) src[n] += (long)gk * ld;
      }
    }
  }
  __device__ __forceinline__ const float* cur(int n) const {
    return linear ? src[n] : src[n] + ((long)q[n] * rm.S + r[n]) * ld;
  }
  __device__ __forceinline__ void issue(unsigned lds_byte) const {
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < NL; ++n) ss_dma16(cur(n), lds_byte + (wave + 4 * n) * 1024);
  }
  __device__ __forceinline__ void advance() {
    if (linear) {
#pragma unroll
      for (int n = 0; n < NL; ++n) src[n] += step;
    } else {
#pragma unroll
      for (int n = 0; n < NL; ++n) {
        r[n] += BK;
        while (r[n] >= rm.G) { r[n] -= rm.G; ++q[n]; }
      }
    }
  }
  // ragged last tile (kvalid < 16 k left): same image, through registers, zeros past the end
  __device__ __forceinline__ void tail(float* img, int kvalid) const {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < NL; ++n) {
      const int sl = 64 * (wave + 4 * n) + lane;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (KC) {
        const int rr = sl >> 2, ku = (sl & 3) ^ kc_swz(rr);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * ku + e < kvalid) v[e] = src[n][e];
      } else {
        if (sl / U < kvalid) v = *reinterpret_cast<const f32x4*>(cur(n));
      }
      *reinterpret_cast<f32x4*>(img + 4 * sl) = v;
    }
  }
};

template <int N>
__device__ __forceinline__ void ss_vmcnt_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void ss_raw_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <bool A_KC, bool B_KC>
__device__ __forceinline__ void gemm_dma_body(GemmParams p, const TileId id, float* dlds /* the ring, at LDS address 0 */) {
  const int m0 = id.by * BM, n0 = id.bx * BN;
  const int bi = id.bz / p.nz, zs = id.bz - bi * p.nz;  // (kcat > 1: the launch has one problem, bi == 0)
  p.A += bi * p.sA; p.B += bi * p.sB;
  if (!(p.flags & 8)) p.C += bi * p.sC;
  if (p.bias) p.bias += bi * p.sBias;
  const int kbeg = zs * p.ksplit;
  const int kend = min(p.K, kbeg + p.ksplit);
  const int nfull = (kend - kbeg) / BK, rem = (kend - kbeg) - nfull * BK;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int i = lane & 15, g = lane >> 4;

  DmaOperand<BM, A_KC> da;
  DmaOperand<BN, B_KC> db;
  constexpr int LPW = DmaOperand<BM, A_KC>::NL + DmaOperand<BN, B_KC>::NL;  // DMA instructions per wave and k tile

  // this lane's operand read offsets inside a stage (floats); the rest of every address is an immediate
  int offA[4], offB[2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
    offA[mt] = A_KC ? (wm * 64 + mt * 16 + i) * BK + 4 * (g ^ kc_swz(i))
                    : 4 * g * BM + (((16 * wm + 4 * mt + (i >> 2) + 4 * (g & 1)) & 31) << 2) + (i & 3);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
    offB[nt] = D_A + (B_KC ? (wn * 32 + nt * 16 + i) * BK + 4 * (g ^ kc_swz(i))
                           : 4 * g * BN + (((8 * wn + 4 * nt + (i >> 2) + 4 * (g & 1)) & 15) << 2) + (i & 3));

  f32x4 acc[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int buf) {
    da.issue((unsigned)(buf * D_STAGE * 4));
    db.issue((unsigned)((buf * D_STAGE + D_A) * 4));
    da.advance();
    db.advance();
  };
  // `refill`: buffer to start the next DMA into (-1: none), issued between the two halves of the MFMA block -- a wave
  // alone on its SIMD has nothing else to cover the ~100 cycles of address arithmetic and M0 traffic
  auto compute = [&](const float* st, int refill) {
    if (refill >= 0) issue(refill);
    float a[4][4], b[2][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (A_KC) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(st + offA[mt]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) a[mt][kk] = v[kk];
      } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) a[mt][kk] = st[offA[mt] + kk * BM];
      }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      if (B_KC) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(st + offB[nt]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) b[nt][kk] = v[kk];
      } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) b[nt][kk] = st[offB[nt] + kk * BN];
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma16(a[mt][kk], b[nt][kk], acc[mt][nt]);
    }
  };

  for (int kc = 0; kc < p.kcat; ++kc) {
  if (kc) ss_raw_barrier();  // the ring is about to be refilled: everybody is done with the previous pair's last tiles
  da.init(p.A + kc * p.sA, p.lda, p.ra, m0, p.M, kbeg);
  db.init(p.B + kc * p.sB, p.ldb, p.rb, n0, p.N, kbeg);
#pragma unroll
  for (int s_ = 0; s_ < DSTAGES - 1; ++s_)
    if (s_ < nfull) issue(s_);
  for (int t0 = 0; t0 < nfull; t0 += DSTAGES) {
#pragma unroll
    for (int s_ = 0; s_ < DSTAGES; ++s_) {
      const int t = t0 + s_;
      if (t < nfull) {
        // tile t has landed once at most the DSTAGES - 2 younger tiles are still in flight (the pipeline tail just drains)
        if (t + DSTAGES - 2 < nfull) ss_vmcnt_wait<(DSTAGES - 2) * LPW>();
        else ss_vmcnt_wait<0>();
        ss_raw_barrier();  // everybody's share of tile t is in LDS, and everybody is done reading tile t - 1 ...
        compute(dlds + s_ * D_STAGE, t + DSTAGES - 1 < nfull ? (s_ + DSTAGES - 1) % DSTAGES : -1);  // ... whose buffer the new tile takes
      }
    }
  }
  if (rem) {
    __syncthreads();
    da.tail(dlds, rem);
    db.tail(dlds + D_A, rem);
    __syncthreads();
    compute(dlds, -1);
  }
  }  // kcat
  static_assert(BM * (BN + 4) <= DSTAGES * D_STAGE, "the output tile is staged in the k-tile ring");
  gemm_epilogue(p, acc, m0, n0, zs, wm, wn, i, g, dlds, id);
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 3) void gemm_dma_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float dlds[];  // DSTAGES x (A image, B image), at LDS address 0
  gemm_dma_body<A_KC, B_KC>(p, TileId{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.y}, dlds);
}

// Several split-K problems of one operand layout in ONE launch (the three weight-gradient GEMMs of a GRU layer): every launch
// boundary on a stream costs the tail of one kernel and the ramp of the next, and these are 30 us kernels.
constexpr int GEMM_GROUP_MAX = 4;
struct GemmGroup {
  GemmParams p[GEMM_GROUP_MAX];
  int first[GEMM_GROUP_MAX + 1];  // workgroups [first[j], first[j+1]) belong to problem j
  int gx[GEMM_GROUP_MAX], gy[GEMM_GROUP_MAX];
  int n;
};

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 3) void gemm_dma_group_kernel(GemmGroup gg) {
  extern __shared__ __attribute__((aligned(16))) float dlds[];
  int j = 0;
#pragma unroll
  for (int q = 1; q < GEMM_GROUP_MAX; ++q)
    if (q < gg.n && (int)blockIdx.x >= gg.first[q]) j = q;
  const int local = blockIdx.x - gg.first[j], gx = gg.gx[j], gy = gg.gy[j];
  const int bx = local % gx, by = (local / gx) % gy, bz = local / (gx * gy);
  gemm_dma_body<A_KC, B_KC>(gg.p[j], TileId{bx, by, bz, gx, gy}, dlds);
}

// C[bi][row][col] += sum over the nz slices of the accumulators the GEMM workgroups left in `ws` (flags bit3).
// One thread owns one accumulator quad (mt, nt, lane) of one output tile: nz coalesced 16-byte reads, four read-modify-writes.
struct ReduceItem { const f32x4* ws; int nz, gx, gy, M, N; float* C; int ldc; long sC; };
struct ReduceGroup {
  ReduceItem it[GEMM_GROUP_MAX];
  int first[GEMM_GROUP_MAX + 1];
  int n;
};
__device__ __forceinline__ void splitk_reduce_body(const ReduceItem& r, int block) {
  const int q = block & 7, tile = block >> 3;
  const int bx = tile % r.gx, by = (tile / r.gx) % r.gy, bi = tile / (r.gx * r.gy);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1, i = lane & 15, g = lane >> 4;
  const int mt = q >> 1, nt = q & 1;
  const f32x4* src = r.ws + ((((long)bi * r.nz) * r.gy + by) * r.gx + bx) * (8 * 256) + q * 256 + tid;
  const long zstride = (long)r.gy * r.gx * (8 * 256);
  // eight slices in flight per thread: the loads are independent, the sum must not become a chain of memory latencies
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f};
  int z = 0;
  for (; z + 8 <= r.nz; z += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(z + u) * zstride];
    s0 += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (z + u < r.nz) ? src[(z + u) * zstride] : f32x4{0.f, 0.f, 0.f, 0.f};
    s0 += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  const int col = bx * BN + wn * 32 + nt * 16 + i;
  if (col >= r.N) return;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int row = by * BM + wm * 64 + mt * 16 + 4 * g + rr;
    if (row < r.M) r.C[bi * r.sC + (long)row * r.ldc + col] += s0[rr];
  }
}
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ReduceItem r) { splitk_reduce_body(r, blockIdx.x); }
__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(ReduceGroup rg) {
  int j = 0;
#pragma unroll
  for (int q = 1; q < GEMM_GROUP_MAX; ++q)
    if (q < rg.n && (int)blockIdx.x >= rg.first[q]) j = q;
  splitk_reduce_body(rg.it[j], blockIdx.x - rg.first[j]);
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight-gradient GEMMs (round 4): C[M][N] += A^T B with both operands k-major ([k][row]), M x N small (the GRU's 576 x 384,
// 384 x 192, 192 x 192), K = B T huge.  The 128 x 64 tiles above read dG once per 64 output columns and the layer input once per
// 128 output rows: 342 MB per launch for 59 MB of operands, and 768 workgroups of 13 k tiles each leave
// 56 MB of slabs.  Here: 192 x 192 output tiles, 512 threads = 8 waves as 4 (M) x 2 (N), wave tile 48 x 96 = 3 x 6 MFMA tiles
// (72 accumulator registers), one workgroup per CU (ring of four 24 KB stages), K split so that the workgroups of ALL problems of
// the group together fill the chip once: 18 tiles x 14 slices at config 2.  dG is then read twice (N = 384) or once, the layer
// input three times; 36 LDS reads feed 72 MFMAs per k tile (24 / 32 above).  Slabs as before (raw accumulators, coalesced) + one
// reduce launch.  Same DMA ring, same source-side rotation of the [k][row] image (unit u of k row k holds row unit
// u - 4 ((k >> 2) & 1) mod 48: the k rows of lane groups g and g + 1 land 16 banks apart).
namespace wide {
constexpr int WT = 192, WNT = 512, WNWV = 8;
constexpr int W_U = WT / 4;                          // 16-byte units per k row
constexpr int W_PIECES = WT * BK / 4 / 64;           // 1 KB DMA pieces per operand and k tile: 12
constexpr int W_STAGE = 2 * WT * BK;                 // floats per stage (A image, B image): 24 KB
constexpr int W_LPW = 2 * W_PIECES / WNWV;           // DMA instructions per wave and k tile: 3
constexpr int W_Q = 18;                              // accumulator quads per thread (3 x 6)
constexpr int W_SLAB = WT * WT;                      // floats a workgroup leaves in the workspace
static_assert(2 * W_PIECES % WNWV == 0, "pieces deal evenly over the waves");

struct WideProblem {
  const float *A, *B;
  float* C;
  int M, N, K, lda, ldb, ldc;
  RowMap ra, rb;
  int ksplit, nz, gx, gy, batch;
  long sA, sB, sC;
  long ws_off;  // floats in front of this problem's slabs
};
struct WideGroup {
  WideProblem p[GEMM_GROUP_MAX];
  int first[GEMM_GROUP_MAX + 1];   // GEMM workgroups [first[j], first[j+1]) belong to problem j
  int rfirst[GEMM_GROUP_MAX + 1];  // reduce workgroups
  float* ws;
  int n;
};

// The k loop of both wide kernels: all eight waves in lockstep -- wait for the own DMA pieces of tile t, barrier, refill of the
// buffer tile t - 1 left, fragments of tile t, 72 MFMAs.  MEASURED AND NOT USED: the bf16 ring kernel's ping-pong (waves 4..7 one
// barrier behind waves 0..3, one half reading fragments / issuing DMA while the other multiplies; two barriers per tile, hazards
// worked out and the tests green): layer 1's weight gradients 112.8 -> 130.7 us, layer 0's 82.8 -> 94.9, the input projections
// 0.1045 -> 0.1082 ms per step.  With f32 MFMAs the half that reads is starved by the half that multiplies (a saturated
// v_mfma_f32_16x16x4_f32 stream leaves the SIMD's other wave one vector instruction per ~180 cycles, docs/LAB_NOTES.md 8c-2 (6)), so its
// "read phase" stretches over the partner's whole MFMA burst instead of hiding under it; in lockstep both halves read with
// nothing to contend with and then share the pipe.
#define WIDE_K_LOOP                                                                                                      \
  {                                                                                                                      \
    _Pragma("unroll") for (int s_ = 0; s_ < DSTAGES - 1; ++s_) if (s_ < nfull) issue(s_);                                \
    for (int t0 = 0; t0 < nfull; t0 += DSTAGES) {                                                                        \
      _Pragma("unroll") for (int s_ = 0; s_ < DSTAGES; ++s_) {                                                           \
        const int t = t0 + s_;                                                                                           \
        if (t < nfull) {                                                                                                 \
          if (t + DSTAGES - 2 < nfull) ss_vmcnt_wait<(DSTAGES - 2) * W_LPW>(); else ss_vmcnt_wait<0>();                  \
          ss_raw_barrier();                                                                                              \
          if (t + DSTAGES - 1 < nfull) issue((s_ + DSTAGES - 1) % DSTAGES);                                              \
          read(dlds + s_ * W_STAGE);                                                                                     \
          mma();                                                                                                         \
        }                                                                                                                \
      }                                                                                                                  \
    }                                                                                                                    \
  }

// one wave's three DMA pieces of a k tile: piece q = wave + 8 n of the 24 (A: 0..11, B: 12..23)
struct WidePieces {
  const float* src[W_LPW];
  int q[W_LPW], r[W_LPW];   // k row = q * S + r for a remapped operand (see DmaOperand)
  int G[W_LPW], S[W_LPW], ld[W_LPW];
  bool lin[W_LPW];
  __device__ __forceinline__ void init(const WideProblem& P, const float* A, const float* B, int m0, int n0, int kbeg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < W_LPW; ++n) {
      const int piece = wave + WNWV * n;
      const bool isb = piece >= W_PIECES;
      const int sl = (piece - (isb ? W_PIECES : 0)) * 64 + lane;
      const int k = sl / W_U, u = sl % W_U;
      int ru = u - 4 * ((k >> 2) & 1);
      ru += ru < 0 ? W_U : 0;
      const RowMap& rm = isb ? P.rb : P.ra;
      const int nrows = isb ? P.N : P.M, row0 = isb ? n0 : m0;
      ld[n] = isb ? P.ldb : P.lda;
      lin[n] = rm.G == 0x7fffffff;
      G[n] = rm.G; S[n] = rm.S;
      const int gk = kbeg + k;
      src[n] = (isb ? B : A) + min(row0 + 4 * ru, nrows - 4) + (long)rm.off * ld[n];
      q[n] = lin[n] ? 0 : gk / rm.G;
      r[n] = gk - q[n] * rm.G;
      if (lin[n]) src[n] += (long)gk * ld[n];
    }
  }
  __device__ __forceinline__ const float* cur(int n) const { return lin[n] ? src[n] : src[n] + ((long)q[n] * S[n] + r[n]) * ld[n]; }
  __device__ __forceinline__ void issue(unsigned stage_byte) const {
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < W_LPW; ++n) ss_dma16(cur(n), stage_byte + (wave + WNWV * n) * 1024);
  }
  __device__ __forceinline__ void advance() {
#pragma unroll
    for (int n = 0; n < W_LPW; ++n) {
      if (lin[n]) src[n] += (long)BK * ld[n];
      else {
        r[n] += BK;
        while (r[n] >= G[n]) { r[n] -= G[n]; ++q[n]; }
      }
    }
  }
  // ragged last tile (kvalid < 16 k rows left): the same image through registers, zeros past the end
  __device__ __forceinline__ void tail(float* img, int kvalid) const {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < W_LPW; ++n) {
      const int piece = wave + WNWV * n;
      const int sl = (piece % W_PIECES) * 64 + lane;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (sl / W_U < kvalid) v = *reinterpret_cast<const f32x4*>(cur(n));
      *reinterpret_cast<f32x4*>(img + piece * 256 + 4 * lane) = v;
    }
  }
};

__global__ __launch_bounds__(WNT) void gemm_wide_group_kernel(WideGroup gg) {
  extern __shared__ __attribute__((aligned(16))) float dlds[];  // DSTAGES x (A image, B image), at LDS address 0
  int j = 0;
#pragma unroll
  for (int q = 1; q < GEMM_GROUP_MAX; ++q)
    if (q < gg.n && (int)blockIdx.x >= gg.first[q]) j = q;
  const WideProblem& P = gg.p[j];
  const int local = blockIdx.x - gg.first[j];
  const int bx = local % P.gx, by = (local / P.gx) % P.gy, bz = local / (P.gx * P.gy);
  const int bi = bz / P.nz, zs = bz - bi * P.nz;
  const int m0 = by * WT, n0 = bx * WT;
  const int kbeg = zs * P.ksplit, kend = min(P.K, kbeg + P.ksplit);
  const int nfull = (kend - kbeg) / BK, rem = (kend - kbeg) - nfull * BK;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid >> 1, wn = wid & 1, i = lane & 15, g = lane >> 4;

  WidePieces dp;
  dp.init(P, P.A + bi * P.sA, P.B + bi * P.sB, m0, n0, kbeg);
  int offA[3], offB[6];
#pragma unroll
  for (int mt = 0; mt < 3; ++mt) {
    const int m = wm * 48 + mt * 16 + i;
    offA[mt] = 4 * g * WT + 4 * (((m >> 2) + 4 * (g & 1)) % W_U) + (m & 3);
  }
#pragma unroll
  for (int nt = 0; nt < 6; ++nt) {
    const int nn = wn * 96 + nt * 16 + i;
    offB[nt] = WT * BK + 4 * g * WT + 4 * (((nn >> 2) + 4 * (g & 1)) % W_U) + (nn & 3);
  }
  f32x4 acc[3][6];
#pragma unroll
  for (int mt = 0; mt < 3; ++mt)
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int buf) {
    dp.issue((unsigned)(buf * W_STAGE * 4));
    dp.advance();
  };
  float a[3][4], b[6][4];
  auto read = [&](const float* st) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) a[mt][kk] = st[offA[mt] + kk * WT];
#pragma unroll
      for (int nt = 0; nt < 6; ++nt) b[nt][kk] = st[offB[nt] + kk * WT];
    }
  };
  auto mma = [&]() {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) acc[mt][nt] = mfma16(a[mt][kk], b[nt][kk], acc[mt][nt]);
  };
  WIDE_K_LOOP
  if (rem) {
    __syncthreads();
    dp.tail(dlds, rem);
    __syncthreads();
    read(dlds);
    mma();
  }
  // raw accumulators as they lie in the registers: 18 coalesced 16-byte stores per thread
  f32x4* w = reinterpret_cast<f32x4*>(gg.ws + P.ws_off) + (long)local * (W_Q * WNT) + threadIdx.x;
#pragma unroll
  for (int mt = 0; mt < 3; ++mt)
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) w[(mt * 6 + nt) * WNT] = acc[mt][nt];
}

// C[bi][row][col] += sum over the nz slices: one workgroup per (problem, batch entry, tile, accumulator quad)
__global__ __launch_bounds__(WNT) void gemm_wide_reduce_kernel(WideGroup gg) {
  int j = 0;
#pragma unroll
  for (int q = 1; q < GEMM_GROUP_MAX; ++q)
    if (q < gg.n && (int)blockIdx.x >= gg.rfirst[q]) j = q;
  const WideProblem& P = gg.p[j];
  const int blk = blockIdx.x - gg.rfirst[j];
  const int q = blk % W_Q, tile = blk / W_Q;  // tile = (bi * gy + by) * gx + bx
  const int bx = tile % P.gx, by = (tile / P.gx) % P.gy, bi = tile / (P.gx * P.gy);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1, i = lane & 15, g = lane >> 4;
  const int mt = q / 6, nt = q % 6;
  const long tiles = (long)P.gx * P.gy;
  const f32x4* src = reinterpret_cast<const f32x4*>(gg.ws + P.ws_off) + (((long)bi * P.nz * tiles + by * P.gx + bx) * W_Q + q) * WNT + tid;
  const long zstride = tiles * W_Q * WNT;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f};
  int z = 0;
  for (; z + 8 <= P.nz; z += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(z + u) * zstride];
    s0 += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (z + u < P.nz) ? src[(z + u) * zstride] : f32x4{0.f, 0.f, 0.f, 0.f};
    s0 += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  const int col = bx * WT + wn * 96 + nt * 16 + i;
  if (col >= P.N) return;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int row = by * WT + wm * 48 + mt * 16 + 4 * g + rr;
    if (row < P.M) P.C[bi * P.sC + (long)row * P.ldc + col] += s0[rr];
  }
}

// ---- the same tile for the GRU input projections: C[M][N] = A[M][K] B[N][K]^T + bias, both operands [row][k] (k contiguous), M = B T
// rows.  128 x 64 tiles made 1 080 workgroups of the 7 680 x 576 x 2-direction problem -- 1.4 rounds of the chip at three per CU --
// with 24 LDS reads per 32 MFMAs; 192 x 192 tiles make 240 workgroups, ONE round at one per CU, and a lane's four k-steps of an
// operand row are one ds_read_b128 (9 LDS reads per 72 MFMAs).  [row][16 k] images, unit s of row r holds k unit s ^ kc_swz(r)
// (the source-side swizzle of the DMA kernels above); the output tile leaves through the ring's LDS in two 96-column halves so
// that a row is written as 384 contiguous bytes.
struct WideKcParams {
  const float *A, *B, *bias;
  float* C;
  int M, N, K, lda, ldb, ldc, gx, gy;
  long sA, sB, sC, sBias;
};
struct WidePiecesKc {
  const float* src[W_LPW];
  __device__ __forceinline__ void init(const WideKcParams& P, const float* A, const float* B, int m0, int n0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < W_LPW; ++n) {
      const int piece = wave + WNWV * n;
      const bool isb = piece >= W_PIECES;
      const int sl = (piece - (isb ? W_PIECES : 0)) * 64 + lane;  // 16-byte slot of the [row][16 k] image
      const int rr = sl >> 2, ku = (sl & 3) ^ kc_swz(rr);
      const int nrows = isb ? P.N : P.M, row0 = isb ? n0 : m0;
      const int gr = min(row0 + rr, nrows - 1);  // rows past the end feed accumulators nobody stores
      src[n] = (isb ? B + (long)gr * P.ldb : A + (long)gr * P.lda) + 4 * ku;
    }
  }
  __device__ __forceinline__ void issue(unsigned stage_byte) const {
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < W_LPW; ++n) ss_dma16(src[n], stage_byte + (wave + WNWV * n) * 1024);
  }
  __device__ __forceinline__ void advance() {
#pragma unroll
    for (int n = 0; n < W_LPW; ++n) src[n] += BK;
  }
  __device__ __forceinline__ void tail(float* img, int kvalid) const {  // ragged last tile through registers, zeros past K
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < W_LPW; ++n) {
      const int piece = wave + WNWV * n;
      const int sl = (piece % W_PIECES) * 64 + lane;
      const int ku = (sl & 3) ^ kc_swz(sl >> 2);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * ku + e < kvalid) v[e] = src[n][e];
      *reinterpret_cast<f32x4*>(img + piece * 256 + 4 * lane) = v;
    }
  }
};

__global__ __launch_bounds__(WNT) void gemm_wide_kc_kernel(WideKcParams P) {
  extern __shared__ __attribute__((aligned(16))) float dlds[];
  const int bx = blockIdx.x % P.gx, by = blockIdx.x / P.gx, bi = blockIdx.y;
  const int m0 = by * WT, n0 = bx * WT;
  const int nfull = P.K / BK, rem = P.K - nfull * BK;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid >> 1, wn = wid & 1, i = lane & 15, g = lane >> 4;
  WidePiecesKc dp;
  dp.init(P, P.A + bi * P.sA, P.B + bi * P.sB, m0, n0);
  int offA[3], offB[6];
#pragma unroll
  for (int mt = 0; mt < 3; ++mt) offA[mt] = (wm * 48 + mt * 16 + i) * BK + 4 * (g ^ kc_swz(i));
#pragma unroll
  for (int nt = 0; nt < 6; ++nt) offB[nt] = WT * BK + (wn * 96 + nt * 16 + i) * BK + 4 * (g ^ kc_swz(i));
  f32x4 acc[3][6];
#pragma unroll
  for (int mt = 0; mt < 3; ++mt)
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto issue = [&](int buf) {
    dp.issue((unsigned)(buf * W_STAGE * 4));
    dp.advance();
  };
  f32x4 a[3], b[6];
  auto read = [&](const float* st) {
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(st + offA[mt]);
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) b[nt] = *reinterpret_cast<const f32x4*>(st + offB[nt]);
  };
  auto mma = [&]() {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) acc[mt][nt] = mfma16(a[mt][kk], b[nt][kk], acc[mt][nt]);
  };
  WIDE_K_LOOP
  if (rem) {
    __syncthreads();
    dp.tail(dlds, rem);
    __syncthreads();
    read(dlds);
    mma();
  }
  // output tile through LDS, one 96-column half (the waves with wn == h) at a time: rows leave as 24 x 16-byte stores
  constexpr int LDS_C = 96 + 4;
  static_assert(WT * LDS_C <= DSTAGES * W_STAGE, "half an output tile is staged in the k-tile ring");
  float* Cb = P.C + bi * P.sC;
  const float* bias = P.bias ? P.bias + bi * P.sBias : nullptr;
  const bool vec_ok = (P.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(Cb) & 15) == 0;
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    __syncthreads();  // the last k tile (h = 0) / the previous half's rows (h = 1) have been read by everybody
    if (wn == h) {
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) {
          const int col = nt * 16 + i, gc = n0 + h * 96 + col;
          const float bv = (bias && gc < P.N) ? bias[gc] : 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) dlds[(wm * 48 + mt * 16 + 4 * g + r) * LDS_C + col] = acc[mt][nt][r] + bv;
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < WT * 24; q += WNT) {
      const int row = q / 24, c4 = 4 * (q % 24);
      const int gr = m0 + row, gc = n0 + h * 96 + c4;
      if (gr < P.M && gc < P.N) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&dlds[row * LDS_C + c4]);
        float* dst = Cb + (long)gr * P.ldc + gc;
        if (vec_ok && gc + 4 <= P.N) *reinterpret_cast<f32x4*>(dst) = v;
        else
          for (int e = 0; e < 4; ++e)
            if (gc + e < P.N) dst[e] = v[e];
      }
    }
  }
}
}  // namespace wide

int splitk_slices(int K, int splits, int* per_out) {
  const int per = ceil_div(ceil_div(K, splits), BK * KSUB) * BK * KSUB;
  if (per_out) *per_out = per;
  return ceil_div(K, per);
}

}  // namespace

// Floats a split-K workspace needs for ss_gemm_f32_batched(flags bit3) + ss_gemm_splitk_reduce on this problem.
extern "C" int ss_gemm_splitk_ws_floats(int M, int N, int K, int splits, int batch, long* floats) {
  SS_REQUIRE(M > 0 && N > 0 && K > 0 && splits >= 1 && batch >= 1 && floats, SS_ERR_ARG);
  *floats = (long)batch * splitk_slices(K, splits, nullptr) * ceil_div(M, BM) * ceil_div(N, BN) * BM * BN;
  return SS_OK;
}

// Second half of a split-K GEMM whose slices were left in `ws` (ss_gemm_f32_batched with flags bit3 and C = ws):
// C[b] += sum of the slices.  Same M, N, K, splits, batch as the GEMM call.
extern "C" int ss_gemm_splitk_reduce(const float* ws, int M, int N, int K, int splits, int batch, float* C, int ldc,
                                     long stride_c, ss_stream_t stream) {
  SS_REQUIRE(ws && C && M > 0 && N > 0 && K > 0 && splits >= 1 && batch >= 1 && ldc >= N, SS_ERR_ARG);
  SS_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0, SS_ERR_ARG);
  const int nz = splitk_slices(K, splits, nullptr), gx = ceil_div(N, BN), gy = ceil_div(M, BM);
  const ReduceItem r{reinterpret_cast<const f32x4*>(ws), nz, gx, gy, M, N, C, ldc, stride_c};
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(batch * gx * gy * 8)), dim3(256), 0, static_cast<hipStream_t>(stream), r);
  return ss_launch_status();
}

static const bool ss_gemm_no_dma = getenv("SS_GEMM_NO_DMA") != nullptr;  // diagnostic: force the register-staged kernel

// Argument checks and kernel parameters shared by the plain and the grouped entry points.
struct GemmLaunch {
  GemmParams p;
  dim3 grid;
  bool dma_ok;
};
static int gemm_prepare(int a_kcontig, int b_kcontig, int M, int N, int K, const float* A, int lda, int a_group, int a_gstride,
                        int a_off, const float* B, int ldb, int b_group, int b_gstride, int b_off, float* C, int ldc,
                        const float* bias, float* a_colsum, int flags, int splits, int batch, long stride_a, long stride_b,
                        long stride_c, long stride_bias, long stride_colsum, GemmLaunch* out) {
  SS_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C && batch >= 1, SS_ERR_ARG);
  SS_REQUIRE(!a_colsum || !a_kcontig, SS_ERR_ARG);
  SS_REQUIRE(splits >= 1 && a_group > 0 && b_group > 0, SS_ERR_ARG);
  // split-K accumulates with atomics: C must already hold the value to add to, and ReLU cannot apply
  SS_REQUIRE((flags & 8) || (splits == 1 && !(flags & 4)) || ((flags & 1) && !(flags & 2)), SS_ERR_ARG);
  // workspace mode: raw accumulators only -- bias, ReLU and accumulation belong to the reduce pass
  SS_REQUIRE(!(flags & 8) || (!(flags & 7) && !bias && (reinterpret_cast<uintptr_t>(C) & 15) == 0), SS_ERR_ARG);
  GemmParams& p = out->p;
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.asum = a_colsum;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.ra = RowMap{a_group, a_gstride, a_off};
  p.rb = RowMap{b_group, b_gstride, b_off};
  p.nz = splitk_slices(K, splits, &p.ksplit);
  p.sA = stride_a; p.sB = stride_b; p.sC = stride_c; p.sBias = stride_bias; p.sAsum = stride_colsum;
  p.flags = flags & 15;
  p.kcat = 1;
  out->grid = dim3(ceil_div(N, BN), ceil_div(M, BM), p.nz * batch);
  SS_REQUIRE(out->grid.z <= 65535, SS_ERR_UNSUPPORTED);
  // the DMA-fed kernel wants 16-byte aligned operands whose [k][row] row counts are multiples of 4; the rest (the head's
  // tiny GEMMs, column sums riding along) stays on the register-staged kernel
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  out->dma_ok = !a_colsum && al16(A) && al16(B) && (lda & 3) == 0 && (ldb & 3) == 0 && (stride_a & 3) == 0 &&
                (stride_b & 3) == 0 && (a_kcontig || (M >= 4 && (M & 3) == 0)) && (b_kcontig || (N >= 4 && (N & 3) == 0)) &&
                p.ksplit >= 4 * BK && !ss_gemm_no_dma;
  return SS_OK;
}

extern "C" int ss_gemm_f32_batched(int a_kcontig, int b_kcontig, int M, int N, int K, const float* A, int lda,
                                   int a_group, int a_gstride, int a_off, const float* B, int ldb, int b_group,
                                   int b_gstride, int b_off, float* C, int ldc, const float* bias, float* a_colsum,
                                   int flags, int splits, int batch, long stride_a, long stride_b, long stride_c,
                                   long stride_bias, long stride_colsum, ss_stream_t stream) {
  GemmLaunch gl;
  const int st = gemm_prepare(a_kcontig, b_kcontig, M, N, K, A, lda, a_group, a_gstride, a_off, B, ldb, b_group, b_gstride, b_off,
                              C, ldc, bias, a_colsum, flags, splits, batch, stride_a, stride_b, stride_c, stride_bias,
                              stride_colsum, &gl);
  if (st != SS_OK) return st;
  GemmParams& p = gl.p;
  dim3 grid = gl.grid, block(256);
  const bool dma_ok = gl.dma_ok;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (flags & 16) {
    // the batch is summed into one C
    SS_REQUIRE(!(flags & 8) && !a_colsum, SS_ERR_ARG);
    if (!dma_ok) {  // register-staged kernel: one launch per pair, the later ones accumulate (stream order)
      for (int b = 0; b < batch; ++b) {
        const int st2 = ss_gemm_f32_batched(a_kcontig, b_kcontig, M, N, K, A + b * stride_a, lda, a_group, a_gstride, a_off,
                                            B + b * stride_b, ldb, b_group, b_gstride, b_off, C, ldc, b ? nullptr : bias, nullptr,
                                            (flags & 15) | (b ? 1 : 0), splits, 1, 0, 0, 0, 0, 0, stream);
        if (st2 != SS_OK) return st2;
      }
      return SS_OK;
    }
    p.kcat = batch;
    grid.z = p.nz;
  }
  // [row][k] x [row][k] with enough 192 x 192 output tiles to fill most of the chip in one round, plain store (+ bias): the
  // wide-tile kernel (the GRU input projections; SS_GEMM_WIDE_KC=0: diagnostic)
  static const bool wide_kc_on = getenv("SS_GEMM_WIDE_KC") == nullptr || atoi(getenv("SS_GEMM_WIDE_KC")) != 0;
  if (dma_ok && wide_kc_on && a_kcontig && b_kcontig && !(flags & 31) && splits == 1 && p.ra.G == 0x7fffffff && p.ra.off == 0 &&
      p.rb.G == 0x7fffffff && p.rb.off == 0 && (K & 3) == 0) {
    const long tiles = (long)ceil_div(M, wide::WT) * ceil_div(N, wide::WT) * batch;
    const int cus = ss_device_cus();
    // ONE round of the chip, at least 80 % full: that is where the gain is (config 2: 240 tiles against 1.4 rounds of 128 x 64
    // ones).  Over several rounds the wide kernel measured 3.5 % SLOWER than the narrow one (shipped shape, 720 tiles = 2.8
    // rounds: gemm_gru_ih 0.335 against 0.323 ms) -- per tile it is no more efficient, it only quantises better.
    if (tiles <= cus && tiles * 10 >= (long)cus * 8 && N >= 96) {
      wide::WideKcParams w;
      w.A = A; w.B = B; w.bias = bias; w.C = C; w.M = M; w.N = N; w.K = K; w.lda = lda; w.ldb = ldb; w.ldc = ldc;
      w.gx = ceil_div(N, wide::WT); w.gy = ceil_div(M, wide::WT);
      w.sA = stride_a; w.sB = stride_b; w.sC = stride_c; w.sBias = stride_bias;
      constexpr size_t wide_lds = (size_t)DSTAGES * wide::W_STAGE * sizeof(float);
      static bool attr_set = false;
      if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(wide::gemm_wide_kc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)wide_lds) != hipSuccess)
          return SS_ERR_LAUNCH;
        attr_set = true;
      }
      hipLaunchKernelGGL(wide::gemm_wide_kc_kernel, dim3((unsigned)(w.gx * w.gy), (unsigned)batch), dim3(wide::WNT), wide_lds, s, w);
      return ss_launch_status();
    }
  }
  if (dma_ok) {
    constexpr size_t lds_bytes = (size_t)DSTAGES * D_STAGE * sizeof(float);
    if (a_kcontig && b_kcontig) hipLaunchKernelGGL((gemm_dma_kernel<true, true>), grid, block, lds_bytes, s, p);
    else if (a_kcontig && !b_kcontig) hipLaunchKernelGGL((gemm_dma_kernel<true, false>), grid, block, lds_bytes, s, p);
    else if (!a_kcontig && b_kcontig) hipLaunchKernelGGL((gemm_dma_kernel<false, true>), grid, block, lds_bytes, s, p);
    else hipLaunchKernelGGL((gemm_dma_kernel<false, false>), grid, block, lds_bytes, s, p);
    return ss_launch_status();
  }
  if (a_kcontig && b_kcontig) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, s, p);
  else if (a_kcontig && !b_kcontig) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, s, p);
  else if (!a_kcontig && b_kcontig) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, s, p);
  return ss_launch_status();
}

// ---- grouped split-K: C_j += A_j^T-ish . B_j for up to GEMM_GROUP_MAX problems, one GEMM launch + one reduce launch
static long group_ws_offset(const ss_gemm_problem* pr, int j) {  // floats in front of problem j's slices
  long off = 0;
  for (int q = 0; q < j; ++q)
    off += (long)pr[q].batch * splitk_slices(pr[q].K, pr[q].splits, nullptr) * ceil_div(pr[q].M, BM) * ceil_div(pr[q].N, BN) * BM * BN;
  return off;
}

// The wide-tile form (namespace wide) of a group: K slices chosen here, from the shapes and the chip's CU count alone, so that the
// workgroups of all problems together fill the chip once (one 512-thread workgroup per CU); `splits` of the records is ignored.
static const bool ss_gemm_dw_wide = getenv("SS_GEMM_DW_WIDE") == nullptr || atoi(getenv("SS_GEMM_DW_WIDE")) != 0;
static bool wide_plan(const ss_gemm_problem* pr, int n, wide::WideGroup* out, long* floats) {
  using namespace wide;
  long work = 0;
  for (int j = 0; j < n; ++j) work += (long)ceil_div(pr[j].M, WT) * ceil_div(pr[j].N, WT) * pr[j].batch * pr[j].K;
  // k rows per workgroup: the chip's share of the work, raised until the workgroups of the group fit the chip in ONE round (a
  // workgroup holds a CU: 264 of them on 256 CUs took twice as long as 252 -- layer 0 of config 2, first version)
  long kt = work / ss_device_cus() > 4 * BK ? work / ss_device_cus() : 4 * BK;
  auto slices = [&](int K, long kt_, int* ksplit) {
    int nz = (int)((K + kt_ / 2) / kt_);
    nz = nz < 1 ? 1 : nz;
    int ks = ceil_div(ceil_div(K, nz), BK) * BK;
    if (ks < 4 * BK) ks = 4 * BK;
    if (ksplit) *ksplit = ks;
    return ceil_div(K, ks);
  };
  for (int it = 0; it < 64; ++it) {
    long wgs = 0;
    for (int j = 0; j < n; ++j) wgs += (long)ceil_div(pr[j].M, WT) * ceil_div(pr[j].N, WT) * pr[j].batch * slices(pr[j].K, kt, nullptr);
    if (wgs <= ss_device_cus()) break;
    kt += kt / 32 + 1;
  }
  long off = 0;
  int first = 0, rfirst = 0;
  for (int j = 0; j < n; ++j) {
    const ss_gemm_problem& q = pr[j];
    WideProblem w;
    w.A = q.A; w.B = q.B; w.C = q.C;
    w.M = q.M; w.N = q.N; w.K = q.K; w.lda = q.lda; w.ldb = q.ldb; w.ldc = q.ldc;
    w.ra = RowMap{q.a_group, q.a_gstride, q.a_off};
    w.rb = RowMap{q.b_group, q.b_gstride, q.b_off};
    w.nz = slices(q.K, kt, &w.ksplit);
    w.gx = ceil_div(q.N, WT); w.gy = ceil_div(q.M, WT); w.batch = q.batch;
    w.sA = q.stride_a; w.sB = q.stride_b; w.sC = q.stride_c;
    w.ws_off = off;
    const int wgs = w.gx * w.gy * w.batch * w.nz;
    off += (long)wgs * W_SLAB;
    if (out) {
      out->p[j] = w;
      out->first[j] = first; out->rfirst[j] = rfirst;
    }
    first += wgs;
    rfirst += w.gx * w.gy * w.batch * W_Q;
  }
  if (out) {
    out->n = n;
    for (int j = n; j <= GEMM_GROUP_MAX; ++j) { out->first[j] = first; out->rfirst[j] = rfirst; }
    for (int j = n; j < GEMM_GROUP_MAX; ++j) out->p[j] = out->p[0];
  }
  if (floats) *floats = off;
  // what the DMA ring and the row clamps need: k-major operands, 16-byte aligned, row counts that are multiples of 4
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  bool ok = ss_gemm_dw_wide && !ss_gemm_no_dma;
  for (int j = 0; j < n; ++j) {
    const ss_gemm_problem& q = pr[j];
    ok = ok && !q.a_kcontig && !q.b_kcontig && al16(q.A) && al16(q.B) && (q.lda & 3) == 0 && (q.ldb & 3) == 0 &&
         (q.stride_a & 3) == 0 && (q.stride_b & 3) == 0 && q.M >= 4 && (q.M & 3) == 0 && q.N >= 4 && (q.N & 3) == 0;
  }
  return ok;
}

extern "C" int ss_gemm_splitk_group_ws_floats(const ss_gemm_problem* problems, int n, long* floats) {
  SS_REQUIRE(problems && floats && n >= 1 && n <= GEMM_GROUP_MAX, SS_ERR_ARG);
  for (int j = 0; j < n; ++j)
    SS_REQUIRE(problems[j].M > 0 && problems[j].N > 0 && problems[j].K > 0 && problems[j].splits >= 1 && problems[j].batch >= 1,
               SS_ERR_ARG);
  // enough for either form of the launch: which one runs is decided from the operands' alignment at call time
  long wide_floats = 0;
  wide_plan(problems, n, nullptr, &wide_floats);
  const long narrow = group_ws_offset(problems, n);
  *floats = wide_floats > narrow ? wide_floats : narrow;
  return SS_OK;
}

extern "C" int ss_gemm_f32_splitk_group(const ss_gemm_problem* problems, int n, float* ws, long ws_floats, int flags, ss_stream_t stream) {
  SS_REQUIRE(problems && ws && n >= 1 && n <= GEMM_GROUP_MAX, SS_ERR_ARG);
  {
    long need = 0;
    const int st = ss_gemm_splitk_group_ws_floats(problems, n, &need);
    if (st != SS_OK) return st;
    SS_REQUIRE(ws_floats >= need, SS_ERR_ARG);
  }
  SS_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0, SS_ERR_ARG);
  {
    wide::WideGroup wg;
    if ((flags & 1) && wide_plan(problems, n, &wg, nullptr)) {
      for (int j = 0; j < n; ++j) SS_REQUIRE(problems[j].C && problems[j].ldc >= problems[j].N, SS_ERR_ARG);
      wg.ws = ws;
      static bool attr_set = false;
      constexpr size_t wide_lds = (size_t)DSTAGES * wide::W_STAGE * sizeof(float);
      if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(wide::gemm_wide_group_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)wide_lds) != hipSuccess)
          return SS_ERR_LAUNCH;
        attr_set = true;
      }
      hipStream_t s = static_cast<hipStream_t>(stream);
      hipLaunchKernelGGL(wide::gemm_wide_group_kernel, dim3((unsigned)wg.first[n]), dim3(wide::WNT), wide_lds, s, wg);
      if (ss_launch_status() != SS_OK) return SS_ERR_LAUNCH;
      hipLaunchKernelGGL(wide::gemm_wide_reduce_kernel, dim3((unsigned)wg.rfirst[n]), dim3(wide::WNT), 0, s, wg);
      return ss_launch_status();
    }
  }
  GemmGroup gg;
  ReduceGroup rg;
  gg.n = rg.n = n;
  gg.first[0] = rg.first[0] = 0;
  bool grouped = true;
  for (int j = 0; j < n; ++j) {
    const ss_gemm_problem& q = problems[j];
    GemmLaunch gl;
    float* wsj = ws + group_ws_offset(problems, j);
    const int st = gemm_prepare(q.a_kcontig, q.b_kcontig, q.M, q.N, q.K, q.A, q.lda, q.a_group, q.a_gstride, q.a_off, q.B, q.ldb,
                                q.b_group, q.b_gstride, q.b_off, wsj, q.N, nullptr, nullptr, 8, q.splits, q.batch, q.stride_a,
                                q.stride_b, 0, 0, 0, &gl);
    if (st != SS_OK) return st;
    SS_REQUIRE(q.C && q.ldc >= q.N, SS_ERR_ARG);
    grouped = grouped && gl.dma_ok && q.a_kcontig == problems[0].a_kcontig && q.b_kcontig == problems[0].b_kcontig;
    gg.p[j] = gl.p;
    gg.gx[j] = gl.grid.x; gg.gy[j] = gl.grid.y;
    gg.first[j + 1] = gg.first[j] + (int)(gl.grid.x * gl.grid.y * gl.grid.z);
    rg.it[j] = ReduceItem{reinterpret_cast<const f32x4*>(wsj), gl.p.nz, (int)gl.grid.x, (int)gl.grid.y, q.M, q.N, q.C, q.ldc, q.stride_c};
    rg.first[j + 1] = rg.first[j] + (int)(q.batch * gl.grid.x * gl.grid.y * 8);
  }
  for (int j = n; j < GEMM_GROUP_MAX; ++j) {
    gg.p[j] = gg.p[0]; gg.gx[j] = gg.gy[j] = 1; gg.first[j + 1] = gg.first[n];
    rg.it[j] = rg.it[0]; rg.first[j + 1] = rg.first[n];
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!grouped) {  // an operand the DMA-fed kernel cannot take: the same work, problem by problem
    for (int j = 0; j < n; ++j) {
      const ss_gemm_problem& q = problems[j];
      float* wsj = ws + group_ws_offset(problems, j);
      int st = ss_gemm_f32_batched(q.a_kcontig, q.b_kcontig, q.M, q.N, q.K, q.A, q.lda, q.a_group, q.a_gstride, q.a_off, q.B,
                                   q.ldb, q.b_group, q.b_gstride, q.b_off, wsj, q.N, nullptr, nullptr, 8, q.splits, q.batch,
                                   q.stride_a, q.stride_b, 0, 0, 0, stream);
      if (st != SS_OK) return st;
      st = ss_gemm_splitk_reduce(wsj, q.M, q.N, q.K, q.splits, q.batch, q.C, q.ldc, q.stride_c, stream);
      if (st != SS_OK) return st;
    }
    return SS_OK;
  }
  constexpr size_t lds_bytes = (size_t)DSTAGES * D_STAGE * sizeof(float);
  const dim3 ggrid((unsigned)gg.first[n]), block(256);
  const bool akc = problems[0].a_kcontig, bkc = problems[0].b_kcontig;
  if (akc && bkc) hipLaunchKernelGGL((gemm_dma_group_kernel<true, true>), ggrid, block, lds_bytes, s, gg);
  else if (akc && !bkc) hipLaunchKernelGGL((gemm_dma_group_kernel<true, false>), ggrid, block, lds_bytes, s, gg);
  else if (!akc && bkc) hipLaunchKernelGGL((gemm_dma_group_kernel<false, true>), ggrid, block, lds_bytes, s, gg);
  else hipLaunchKernelGGL((gemm_dma_group_kernel<false, false>), ggrid, block, lds_bytes, s, gg);
  if (ss_launch_status() != SS_OK) return SS_ERR_LAUNCH;
  hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3((unsigned)rg.first[n]), dim3(256), 0, s, rg);
  return ss_launch_status();
}

extern "C" int ss_gemm_f32(int a_kcontig, int b_kcontig, int M, int N, int K, const float* A, int lda, int a_group,
                           int a_gstride, int a_off, const float* B, int ldb, int b_group, int b_gstride, int b_off,
                           float* C, int ldc, const float* bias, float* a_colsum, int flags, int splits,
                           ss_stream_t stream) {
  return ss_gemm_f32_batched(a_kcontig, b_kcontig, M, N, K, A, lda, a_group, a_gstride, a_off, B, ldb, b_group,
                             b_gstride, b_off, C, ldc, bias, a_colsum, flags, splits, 1, 0, 0, 0, 0, 0, stream);
}
