// bf16-MFMA GEMM of the BASELINE config-5 GRU layers (H = 512): C[M,N] (+)= opA[M,K] * opB[K,N] (+ bias[N]).
//
// Same operand conventions as ss_gemm_f32_batched (include/ss_hotpath.h): f32 operands in HBM, either k-contiguous
// ([row][k]) or k-major ([k][row]) with the storage-row remap that pairs dG[b][t] with h[b][t -+ 1].  The operands are
// rounded to bf16 (nearest even) while they are staged into LDS and multiplied on v_mfma_f32_16x16x32_bf16 with f32
// accumulation; C, the bias and every accumulation (split-K float atomics) stay f32.
//
// 128 x 128 x 32 tiles, 4 waves (2 x 2, 64 x 64 each = 16 accumulators), double-buffered LDS, one barrier per k tile,
// global loads of tile t+1 issued before the MFMAs of tile t.  k-contiguous operands sit in LDS as [row][32 + 8] and
// are read with ds_read_b128 (row stride 80 B = 20 banks: the 16 rows of a fragment cover all 64 banks); k-major
// operands sit as [k][128 + 8] (written with 8-byte stores as they arrive) and are read with the transposing
// ds_read_b64_tr_b16 -- no transpose pass, no strided global loads.
#include "bf16_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDK = BK + 8;    // [row][k] image, elements per row
constexpr int LDR = 128 + 8;   // [k][row] image, elements per k line
constexpr int TILE_ELEMS = (BM * LDK > BK * LDR) ? BM * LDK : BK * LDR;

struct GemmBfParams {
  int M, N, K;
  const void* A; int lda, a_group, a_gstride, a_off;   // f32 or bf16 elements (template)
  const void* B; int ldb, b_group, b_gstride, b_off;
  float* C; int ldc;
  const float* bias;
  int flags, splits;
  long sa, sb, sc, sbias;
};

__device__ __forceinline__ long remap_row(int r, int group, int gstride, int off) {
  return (long)(r / group) * gstride + (r % group) + off;
}

// One operand tile (128 rows x 32 k) from HBM into 4 float4 registers per thread.
// KC = 1: storage [row][k]: thread -> (row = idx / 8, k = 4 (idx % 8)), idx = tid + 256 i.
// KC = 0: storage [k][row]: thread -> (k = idx / 32, row = 4 (idx % 32)).
// BF = 1: the operand already is bf16 in HBM (a copy its producer wrote): 8-byte loads, no conversion -- half the bytes of the
// f32 form, and the GRU-layer GEMMs are bound by exactly those bytes (128 x 128 x 32 tiles of f32 operands: 32 FLOP per byte)
template <int KC, int BF>
struct TileRegs {
  f32x4 f[BF ? 1 : 4];
  uint2 h[BF ? 4 : 1];
  __device__ __forceinline__ void load(const void* __restrict__ srcv, int ld, int group, int gstride, int off, int row0, int rows, int k0,
                                       int k_end, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      long o = -1;
      if (KC) {
        const int r = row0 + (idx >> 3), k = k0 + 4 * (idx & 7);
        if (r < rows && k < k_end) o = remap_row(r, group, gstride, off) * ld + k;
      } else {
        const int k = k0 + (idx >> 5), r = row0 + 4 * (idx & 31);
        if (k < k_end && r < rows) o = remap_row(k, group, gstride, off) * ld + r;
      }
      if (BF) h[i] = o >= 0 ? *reinterpret_cast<const uint2*>(static_cast<const bf16_t*>(srcv) + o) : uint2{0u, 0u};
      else f[i] = o >= 0 ? *reinterpret_cast<const f32x4*>(static_cast<const float*>(srcv) + o) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __device__ __forceinline__ void store(bf16_t* tile, int tid) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      bf16_t* dst = KC ? tile + (idx >> 3) * LDK + 4 * (idx & 7) : tile + (idx >> 5) * LDR + 4 * (idx & 31);
      *reinterpret_cast<uint2*>(dst) = BF ? h[i] : pack_bf16x4(f[i][0], f[i][1], f[i][2], f[i][3]);
    }
  }
};

template <int AKC, int BKC, int ABF, int BBF>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmBfParams p) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[4 * TILE_ELEMS];  // A0 B0 A1 B1
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int batch = blockIdx.z / p.splits, split = blockIdx.z % p.splits;
  const void* A = static_cast<const char*>(p.A) + batch * p.sa * (ABF ? 2 : 4);
  const void* B = static_cast<const char*>(p.B) + batch * p.sb * (BBF ? 2 : 4);
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

  if (kt0 < kt1) {
    TileRegs<AKC, ABF> va;
    TileRegs<BKC, BBF> vb;
    va.load(A, p.lda, p.a_group, p.a_gstride, p.a_off, m0, p.M, kt0 * BK, p.K, tid);
    vb.load(B, p.ldb, p.b_group, p.b_gstride, p.b_off, n0, p.N, kt0 * BK, p.K, tid);
    va.store(lds, tid);
    vb.store(lds + TILE_ELEMS, tid);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const int cur = (kt - kt0) & 1;
      const bf16_t* As = lds + (2 * cur) * TILE_ELEMS;
      const bf16_t* Bs = lds + (2 * cur + 1) * TILE_ELEMS;
      const bool more = kt + 1 < kt1;
      if (more) {
        va.load(A, p.lda, p.a_group, p.a_gstride, p.a_off, m0, p.M, (kt + 1) * BK, p.K, tid);
        vb.load(B, p.ldb, p.b_group, p.b_gstride, p.b_off, n0, p.N, (kt + 1) * BK, p.K, tid);
      }
      s16x8 fa[4], fb[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[t] = AKC ? lds_frag(As + (wm + 16 * t + li) * LDK + 8 * g) : lds_frag_tr(As + wm + 16 * t, LDR, lane);
        fb[t] = BKC ? lds_frag(Bs + (wn + 16 * t + li) * LDK + 8 * g) : lds_frag_tr(Bs + wn + 16 * t, LDR, lane);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = mfma_bf16(fa[a], fb[b], acc[a][b]);
      if (more) {
        va.store(lds + (2 * (cur ^ 1)) * TILE_ELEMS, tid);
        vb.store(lds + (2 * (cur ^ 1) + 1) * TILE_ELEMS, tid);
      }
      __syncthreads();
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
}

}  // namespace

extern "C" int ss_gemm_bf16_batched_ex(int a_kcontig, int b_kcontig, int a_is_bf16, int b_is_bf16, int M, int N, int K, const void* A,
                                       int lda, int a_group, int a_gstride, int a_off, const void* B, int ldb, int b_group,
                                       int b_gstride, int b_off, float* C, int ldc, const float* bias, int flags, int splits, int batch,
                                       long stride_a, long stride_b, long stride_c, long stride_bias, ss_stream_t stream) {
  SS_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0 && splits > 0, SS_ERR_ARG);
  SS_REQUIRE(a_group > 0 && b_group > 0, SS_ERR_ARG);
  SS_REQUIRE(!(flags & ~5), SS_ERR_UNSUPPORTED);                 // bit0 accumulate, bit2 atomics
  SS_REQUIRE(splits == 1 || (flags & 1), SS_ERR_ARG);            // K slices add into a C the caller initialised
  SS_REQUIRE(!(flags & 4) || (flags & 1), SS_ERR_ARG);
  SS_REQUIRE(!b_is_bf16 || a_is_bf16, SS_ERR_UNSUPPORTED);       // built: (f32, f32), (bf16, f32), (bf16, bf16)
  // 16-byte (f32) / 8-byte (bf16) loads along the contiguous dimension of either layout
  SS_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && stride_a % 4 == 0 && stride_b % 4 == 0, SS_ERR_UNSUPPORTED);
  SS_REQUIRE((a_kcontig ? K : M) % 4 == 0 && (b_kcontig ? K : N) % 4 == 0, SS_ERR_UNSUPPORTED);
  SS_REQUIRE((reinterpret_cast<uintptr_t>(A) & (a_is_bf16 ? 7 : 15)) == 0 && (reinterpret_cast<uintptr_t>(B) & (b_is_bf16 ? 7 : 15)) == 0,
             SS_ERR_UNSUPPORTED);
  GemmBfParams p;
  p.M = M; p.N = N; p.K = K;
  p.A = A; p.lda = lda; p.a_group = a_group; p.a_gstride = a_gstride; p.a_off = a_off;
  p.B = B; p.ldb = ldb; p.b_group = b_group; p.b_gstride = b_gstride; p.b_off = b_off;
  p.C = C; p.ldc = ldc; p.bias = bias; p.flags = flags;
  const int nkt = (K + BK - 1) / BK;
  p.splits = splits < nkt ? splits : nkt;
  p.sa = stride_a; p.sb = stride_b; p.sc = stride_c; p.sbias = stride_bias;
  dim3 grid(ceil_div(N, BN), ceil_div(M, BM), batch * p.splits);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define SS_GEMM_CASE(AK, BK_, AB, BB)                                                  \
  if (!!a_kcontig == AK && !!b_kcontig == BK_ && !!a_is_bf16 == AB && !!b_is_bf16 == BB) { \
    hipLaunchKernelGGL((gemm_bf16_kernel<AK, BK_, AB, BB>), grid, dim3(256), 0, s, p);  \
    return ss_launch_status();                                                         \
  }
#define SS_GEMM_LAYOUTS(AB, BB) SS_GEMM_CASE(1, 1, AB, BB) SS_GEMM_CASE(1, 0, AB, BB) SS_GEMM_CASE(0, 1, AB, BB) SS_GEMM_CASE(0, 0, AB, BB)
  SS_GEMM_LAYOUTS(0, 0)
  SS_GEMM_LAYOUTS(1, 0)
  SS_GEMM_LAYOUTS(1, 1)
#undef SS_GEMM_LAYOUTS
#undef SS_GEMM_CASE
  return SS_ERR_UNSUPPORTED;
}

extern "C" int ss_gemm_bf16_batched(int a_kcontig, int b_kcontig, int M, int N, int K, const float* A, int lda, int a_group,
                                    int a_gstride, int a_off, const float* B, int ldb, int b_group, int b_gstride, int b_off,
                                    float* C, int ldc, const float* bias, int flags, int splits, int batch, long stride_a,
                                    long stride_b, long stride_c, long stride_bias, ss_stream_t stream) {
  return ss_gemm_bf16_batched_ex(a_kcontig, b_kcontig, 0, 0, M, N, K, A, lda, a_group, a_gstride, a_off, B, ldb, b_group, b_gstride,
                                 b_off, C, ldc, bias, flags, splits, batch, stride_a, stride_b, stride_c, stride_bias, stream);
}
