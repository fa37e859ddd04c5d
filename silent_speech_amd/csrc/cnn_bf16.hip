// bf16 ROI-CNN forward of BASELINE config 5: geometry and data flow in cnn_bf16.h.
//
//   ss_c5_conv1_fwd      uint8 frame -> exact integer statistics -> (u/255 - mu)/std table -> bf16 haloed image ->
//                        conv 1->16 as a PATCH GEMM: K = 32 = a 4-row x 8-column input patch, M = the 16 patches of one
//                        row pair (96 = 16 x 6 columns), one MFMA per output position of the patch (2 rows x 6 columns)
//                        against a constant sparse weight fragment kept in registers -- 1 operand read per 12 MFMAs,
//                        2x2 max-pool across the accumulators of a lane -> pooled map + argmax bytes
//   ss_c5_conv_fwd       layers 2 and 3: implicit GEMM out of the haloed LDS image (cnn_bf16.h), bias + ReLU + 2x2
//                        max-pool inside a lane (an M tile is 2 rows x 8 columns = 4 pool windows, the 4 rows a lane
//                        holds of a 16x16 result are one window)
//   ss_c5_conv_last_fwd  layer 4 (no pool): bias + ReLU + sign mask + global average + Linear(96 -> E), written
//                        straight into the (B,T,x_dim+E) GRU input (the torch.cat of train_model_official.py:297)
//
// Replaces /root/reference/train_model_official.py:286-291 (normalise) and :212-229 (CNN) for the wider model.
#include <stdlib.h>

#include "cnn_bf16.h"

extern int ss_cnn_max_wgs;  // roi_cnn.hip: test hook, workgroups per launch (0 = one per CU)

namespace {
using namespace c5;

STAMP_TABLE(ss_debug_stamps_c5_fwd)

__device__ __forceinline__ void zero_lds(void* base, int bytes, int tid) {
  uint4* p = reinterpret_cast<uint4*>(base);
  for (int q = tid; q < bytes / 16; q += NT) p[q] = uint4{0u, 0u, 0u, 0u};
}

// ------------------------------------------------------------------------------------------------ conv1
struct Conv1Params {
  const uint8_t* R;   // (N, 96, 96)
  int N, standardize;
  const float *w1, *b1;  // (16,1,3,3), (16)
  bf16_t* a1;         // (N, 48, 48, 16)
  uint8_t* i1;        // (N, 48, 48, 16)
  float* st;          // (N, 2): mean, std of the frame (backward reuses them) or null
};


__global__ __launch_bounds__(NT, 2) void conv1_fwd_kernel(Conv1Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int HP = HW0 / 2;
  bf16_t* img = reinterpret_cast<bf16_t*>(smem);                         // [98][RS0]
  constexpr int o_tab = round_up(98 * RS0 * 2, 16);
  float* s_xn = reinterpret_cast<float*>(smem + o_tab);                  // [256]
  float* s_misc = s_xn + 256;                                            // [64]
  constexpr int o_a1 = o_tab + (256 + 64) * 4;
  bf16_t* oa = reinterpret_cast<bf16_t*>(smem + o_a1);                   // [48][48][16]
  uint8_t* oi = smem + o_a1 + HP * HP * C1 * 2;                          // [48][48][16]
  unsigned* s_red = reinterpret_cast<unsigned*>(s_misc);                 // [NW][2]
  float* s_stat = s_misc + 32;

  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: as an SGPR the tile / row arithmetic that hangs on it runs on the scalar unit
  zero_lds(img, o_tab, tid);
  // constant B fragments: output position q = (oy, ox) of a patch reads patch cell (oy + ky, ox + kx)
  s16x8 bq[12];
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    const int oy = q / 6, ox = q % 6;
    s16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ky = g - oy, kx = j - ox;
      const float w = (ky >= 0 && ky <= 2 && kx >= 0 && kx <= 2) ? p.w1[li * 9 + ky * 3 + kx] : 0.f;
      f[j] = (short)to_bf16(w);
    }
    bq[q] = f;
  }
  const float bias = p.b1[li];
  __syncthreads();

  for (int n = blockIdx.x; n < p.N; n += gridDim.x) {
    // ---- frame bytes, exact statistics (train_model_official.py:286-290)
    const uint4* src = reinterpret_cast<const uint4*>(p.R + (long)n * HW0 * HW0);
    uint4 px[2];
    px[0] = src[tid];
    px[1] = (tid + NT < HW0 * HW0 / 16) ? src[tid + NT] : uint4{0u, 0u, 0u, 0u};
    unsigned su = 0, sq = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const unsigned wds[4] = {px[k].x, px[k].y, px[k].z, px[k].w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const unsigned u = (wds[e] >> (8 * b)) & 255u;
          su += u;
          sq += u * u;
        }
    }
    su = wave_sum_u32(su);
    sq = wave_sum_u32(sq);
    if (lane == 0) { s_red[2 * wv] = su; s_red[2 * wv + 1] = sq; }
    __syncthreads();
    if (tid == 0) {
      unsigned long long tsu = 0, tsq = 0;
      for (int k = 0; k < NW; ++k) { tsu += s_red[2 * k]; tsq += s_red[2 * k + 1]; }
      float mu = 0.f, sd = 1.f;
      if (p.standardize) {
        const double nn = (double)(HW0 * HW0);
        mu = (float)((double)tsu / nn) / 255.0f;
        const double var = ((double)tsq - (double)tsu * (double)tsu / nn) / (nn - 1.0);
        sd = fmaxf((float)(sqrt(var > 0.0 ? var : 0.0) / 255.0), 1e-6f);
      }
      s_stat[0] = mu;
      s_stat[1] = sd;
      if (p.st) { p.st[2 * (long)n] = mu; p.st[2 * (long)n + 1] = sd; }
    }
    __syncthreads();
    if (tid < 256) {
      const float rr = (float)tid / 255.0f;
      s_xn[tid] = p.standardize ? (rr - s_stat[0]) / s_stat[1] : rr;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int q = tid + k * NT;
      if (q < HW0 * HW0 / 16) {
        const int lin = q * 16;
        bf16_t* dst = img + (lin / HW0 + 1) * RS0 + (lin % HW0) + 1;
        const unsigned wds[4] = {px[k].x, px[k].y, px[k].z, px[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int b = 0; b < 4; ++b) dst[4 * e + b] = to_bf16(s_xn[(wds[e] >> (8 * b)) & 255u]);
      }
    }
    __syncthreads();
    // ---- patch GEMM: row pair yp, patch i = li covers haloed rows 2yp..2yp+3, haloed columns 6i..6i+7
    for (int yp = wv; yp < HP; yp += NW) {
      const unsigned* ap = reinterpret_cast<const unsigned*>(img + (2 * yp + g) * RS0 + 6 * li);
      const unsigned a0 = ap[0], a1v = ap[1], a2 = ap[2], a3 = ap[3];
      const s16x8 fa = __builtin_bit_cast(s16x8, uint4{a0, a1v, a2, a3});
      f32x4 acc[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) acc[q] = mfma_bf16(fa, bq[q], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          float best = acc[2 * w][r];
          int bi = 0;
          if (acc[2 * w + 1][r] > best) { best = acc[2 * w + 1][r]; bi = 1; }
          if (acc[6 + 2 * w][r] > best) { best = acc[6 + 2 * w][r]; bi = 2; }
          if (acc[6 + 2 * w + 1][r] > best) { best = acc[6 + 2 * w + 1][r]; bi = 3; }
          const float v = fmaxf(best + bias, 0.f);
          const int o = (yp * HP + 3 * (4 * g + r) + w) * C1 + li;
          oa[o] = to_bf16(v);
          oi[o] = (uint8_t)(v > 0.f ? bi : IDX_DEAD);
        }
    }
    __syncthreads();
    uint4* da = reinterpret_cast<uint4*>(p.a1 + (long)n * HP * HP * C1);
    for (int q = tid; q < HP * HP * C1 * 2 / 16; q += NT) da[q] = reinterpret_cast<const uint4*>(oa)[q];
    uint4* di = reinterpret_cast<uint4*>(p.i1 + (long)n * HP * HP * C1);
    for (int q = tid; q < HP * HP * C1 / 16; q += NT) di[q] = reinterpret_cast<const uint4*>(oi)[q];
    // the next frame's epilogue writes oa / oi only behind the barriers of its own statistics phase
  }
}

constexpr int CONV1_LDS = round_up(98 * RS0 * 2, 16) + (256 + 64) * 4 + 48 * 48 * C1 * 3;

// ------------------------------------------------------------------------------------------------ layers 2..4
struct ConvFwdParams {
  const bf16_t* in;   // (N, H, W, CIN)
  int N;
  const float *w, *b; // (COUT, CIN, 3, 3), (COUT)
  bf16_t* out;        // (N, H/2, W/2, COUT)            pooled layers
  uint8_t* idx;       // (N, H/2, W/2, COUT)
  // last layer
  const float *wfc, *bfc;  // (E, COUT), (E)
  int E;
  float* z;           // row n at z + n * ld_z
  int ld_z;
  uint8_t* mask;      // (N, H*W, COUT) conv output > 0, or null (inference)
  float* feat;        // (N, COUT) averaged features, or null
};

// bf16 weight matrix [COUT][kk = tap*CIN + ci] into LDS
template <int CIN, int COUT>
__device__ __forceinline__ void stage_weights(const float* __restrict__ w, bf16_t* wl, int tid) {
  using WM = Wmat<CIN, COUT>;
  for (int q = tid; q < COUT * WM::KP; q += NT) {
    const int co = q / WM::KP, kk = q % WM::KP, tap = kk / CIN, ci = kk % CIN;
    wl[co * WM::LD + kk] = to_bf16(tap < 9 ? w[(co * CIN + ci) * 9 + tap] : 0.f);
  }
}

// un-haloed NHWC frame (HBM) -> haloed LDS image, 16-byte pieces
template <class IM>
__device__ __forceinline__ void load_image(const bf16_t* __restrict__ src, bf16_t* img, int tid) {
  constexpr int CH = IM::C / 8;  // 16-byte pieces per pixel
  for (int q = tid; q < IM::H * IM::W * CH; q += NT) {
    const int pix = q / CH, c8 = q % CH;
    *reinterpret_cast<uint4*>(img + IM::at(pix / IM::W, pix % IM::W) + 8 * c8) = reinterpret_cast<const uint4*>(src)[q];
  }
}

// element offset of the A fragment of k step s relative to the lane's base (pixel (y-1, x-1), chunk folded into the base)
template <class IM>
__device__ __forceinline__ int koff(int s, int hi) {
  constexpr int CIN = IM::C;
  if (CIN >= 32) {
    const int tap = (32 * s) / CIN, ci0 = (32 * s) % CIN;
    return (tap / 3) * IM::RS + (tap % 3) * IM::PS + ci0;
  }
  int tap = 2 * s + hi;  // 16 channels: a k step is two taps
  if (tap > 8) tap = 8;  // padding half of the last step: any valid address, its weights are zero
  return (tap / 3) * IM::RS + (tap % 3) * IM::PS;
}

template <int CIN, int COUT, int H, int W, bool LAST, int MT, int NTL>
__global__ __launch_bounds__(NT, 2) void conv_fwd_kernel(ConvFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using IM = Img<CIN, H, W>;
  using WM = Wmat<CIN, COUT>;
  constexpr int HO = H / 2, WO = W / 2;
  bf16_t* img = reinterpret_cast<bf16_t*>(smem);
  bf16_t* wl = reinterpret_cast<bf16_t*>(smem + IM::BYTES);
  constexpr int o_out = IM::BYTES + round_up(WM::BYTES, 16);
  bf16_t* oa = reinterpret_cast<bf16_t*>(smem + o_out);                       // pooled: [HO][WO][COUT]
  uint8_t* oi = smem + o_out + (LAST ? 0 : HO * WO * COUT * 2);               // pooled: argmax; last: mask [H*W][COUT]
  constexpr int o_misc = o_out + (LAST ? H * W * COUT : HO * WO * COUT * 3);
  float* s_bias = reinterpret_cast<float*>(smem + round_up(o_misc, 16));      // [COUT]
  float* s_feat = s_bias + COUT;                                              // [COUT]

  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: as an SGPR the tile / row arithmetic that hangs on it runs on the scalar unit
  zero_lds(img, IM::BYTES, tid);
  stage_weights<CIN, COUT>(p.w, wl, tid);
  for (int q = tid; q < COUT; q += NT) s_bias[q] = p.b[q];
  __syncthreads();

  constexpr int MTILES = LAST ? H * W / 16 : HO * (W / 8);
  constexpr int NTILES = COUT / 16;
  static_assert(MTILES % MT == 0 && NTILES % NTL == 0, "unit split");
  constexpr int UNITS = (MTILES / MT) * (NTILES / NTL);
  const int chunk = (CIN >= 32) ? g : (g & 1), hi = (CIN >= 32) ? 0 : (g >> 1);

  BandLoad<IM, H, H> pre;  // the next frame, on its way while this one is computed
  if ((int)blockIdx.x < p.N) pre.issue(p.in + (long)blockIdx.x * H * W * CIN, 0, tid);
  STAMP_ENTRY;
  STAMP_DECL;
  // The pooled map of frame n leaves for HBM at the top of frame n + 1, BEHIND the commit of that frame's prefetched image: vector
  // memory operations retire in order and hipcc cannot count stores issued in a loop, so the commit's wait for its (older) loads is
  // vmcnt(0) -- with the copy-out in front of it that wait sat through the stores' round trip every frame.
  auto copy_out = [&](int n) {
    uint4* da = reinterpret_cast<uint4*>(p.out + (long)n * HO * WO * COUT);
    for (int q = tid; q < HO * WO * COUT * 2 / 16; q += NT) da[q] = reinterpret_cast<const uint4*>(oa)[q];
    uint4* di = reinterpret_cast<uint4*>(p.idx + (long)n * HO * WO * COUT);
    for (int q = tid; q < HO * WO * COUT / 16; q += NT) di[q] = reinterpret_cast<const uint4*>(oi)[q];
  };
  int n_prev = -1;
  for (int n = blockIdx.x; n < p.N; n += gridDim.x) {
    STAMP(15);
    pre.commit(img, 0, tid);
    if (!LAST && n_prev >= 0) copy_out(n_prev);  // (its staging area is rewritten behind the barrier below)
    n_prev = n;
    if (LAST)
      for (int q = tid; q < COUT; q += NT) s_feat[q] = 0.f;
    STAMP(0);
    __syncthreads();
    STAMP(1);
    if (n + (int)gridDim.x < p.N) pre.issue(p.in + (long)(n + gridDim.x) * H * W * CIN, 0, tid);
    STAMP(2);
    for (int u = wv; u < UNITS; u += NW) {
      const int mg = u % (MTILES / MT), ng = u / (MTILES / MT);
      int base[MT];
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        const int mt = mg * MT + a;
        int y, x;
        if (LAST) {
          const int P = 16 * mt + li;
          y = P / W; x = P % W;
        } else {
          const int yp = mt / (W / 8), xt = mt % (W / 8);
          y = 2 * yp + ((li >> 1) & 1); x = 8 * xt + 2 * (li >> 2) + (li & 1);
        }
        base[a] = IM::at(y - 1, x - 1) + 8 * chunk;
      }
      f32x4 acc[MT][NTL];
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NTL; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      // operand fragments one k step ahead of the MFMAs (two register sets, the loop is fully unrolled): left alone hipcc reads a
      // step's fragments, waits, multiplies -- an LDS latency per k step that the SIMD's other wave only partly covers
      s16x8 fa[2][MT], fb[2][NTL];
      auto read_step = [&](int s, s16x8 (&xa)[MT], s16x8 (&xb)[NTL]) {
        const int off = (CIN >= 32) ? koff<IM>(s, 0) : (hi ? koff<IM>(s, 1) : koff<IM>(s, 0));
#pragma unroll
        for (int a = 0; a < MT; ++a) xa[a] = lds_frag(img + base[a] + off);
#pragma unroll
        for (int b = 0; b < NTL; ++b) xb[b] = lds_frag(wl + (16 * (ng * NTL + b) + li) * WM::LD + 32 * s + 8 * g);
      };
      read_step(0, fa[0], fb[0]);
#pragma unroll
      for (int s = 0; s < WM::KSTEPS; ++s) {
        if (s + 1 < WM::KSTEPS) read_step(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
        SS_SCHED_FENCE();
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int b = 0; b < NTL; ++b) acc[a][b] = mfma_bf16(fa[s & 1][a], fb[s & 1][b], acc[a][b]);
        SS_SCHED_FENCE();
      }
      // ---- epilogue
#pragma unroll
      for (int b = 0; b < NTL; ++b) {
        const int co = 16 * (ng * NTL + b) + li;
        const float bias = s_bias[co];
        float fsum = 0.f;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
          const int mt = mg * MT + a;
          const f32x4 v = acc[a][b];
          if (LAST) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float x = v[r] + bias;
              if (p.mask) oi[(16 * mt + 4 * g + r) * COUT + co] = x > 0.f;
              fsum += fmaxf(x, 0.f);
            }
          } else {
            float best = v[0];
            int bi = 0;
            if (v[1] > best) { best = v[1]; bi = 1; }
            if (v[2] > best) { best = v[2]; bi = 2; }
            if (v[3] > best) { best = v[3]; bi = 3; }
            const float x = fmaxf(best + bias, 0.f);
            const int yp = mt / (W / 8), xt = mt % (W / 8);
            const int o = (yp * WO + 4 * xt + g) * COUT + co;
            oa[o] = to_bf16(x);
            oi[o] = (uint8_t)(x > 0.f ? bi : IDX_DEAD);
          }
        }
        if (LAST) {
          fsum += __shfl_xor(fsum, 16, 64);
          fsum += __shfl_xor(fsum, 32, 64);
          if (g == 0) atomicAdd(&s_feat[co], fsum);
        }
      }
    }
    STAMP(3);
    __syncthreads();
    STAMP(4);
    if (LAST) {
      if (p.mask) {
        uint4* dm = reinterpret_cast<uint4*>(p.mask + (long)n * H * W * COUT);
        for (int q = tid; q < H * W * COUT / 16; q += NT) dm[q] = reinterpret_cast<const uint4*>(oi)[q];
      }
      for (int q = tid; q < COUT; q += NT) {
        const float f = s_feat[q] / (float)(H * W);
        s_feat[q] = f;
        if (p.feat) p.feat[(long)n * COUT + q] = f;
      }
      __syncthreads();
      if (p.z) {  // (null: Linear(96 -> E) over all frames is the caller's GEMM on `feat` -- 64 threads walking a row of W_fc each,
                  // per frame, was a chain of global loads with the other 448 threads waiting at the barrier)
        for (int e = tid; e < p.E; e += NT) {
          float o = p.bfc[e];
          for (int c = 0; c < COUT; ++c) o += s_feat[c] * p.wfc[e * COUT + c];
          p.z[(long)n * p.ld_z + e] = o;
        }
        __syncthreads();
      }
    }
    STAMP(5);
  }
  if (!LAST && n_prev >= 0) copy_out(n_prev);
  STAMP_FLUSH();
}

// ------------------------------------------------------------------------------------------------ layer 4, weight-stationary
// conv 64 -> 96 on the 12 x 12 map: 110 KB of bf16 weights for 18 KB of input per frame.  conv_fwd_kernel keeps the weights in LDS,
// which leaves room for ONE frame: load, multiply, mask copy-out and average are phases of one workgroup with barriers between
// them, six of eight waves multiply (11.9 k cycles per frame for 3.9 k cycles of MFMA, stage timers of round 3).  Here the weights
// live in REGISTERS as the A operand of the transposed product D[co][pixel]: a consumer wave holds the 32 output channels of its
// channel group = 2 x 18 fragments (144 registers) for the whole walk and multiplies them with some of the frame's nine pixel
// tiles (16 consecutive pixels each: conv_fwd's A-fragment read as the B operand), 36 MFMAs and 18 fragment reads per tile.  Waves 6 and 7 are PRODUCERS: they write the next frame (loaded a whole frame time earlier) into the other
// image, copy the previous frame's sign mask out and finish its average; one workgroup barrier per frame.
#ifndef SS_L4_NRD
#define SS_L4_NRD 9
#endif
struct ConvLastWsParams {
  const bf16_t* in;   // (N, 12, 12, 64)
  int N;
  const float *w, *b; // (96, 64, 3, 3), (96)
  uint8_t* mask;      // (N, 144, 96) or null
  float* feat;        // (N, 96)
};
constexpr int L4_IMG = Img<C3, 12, 12>::BYTES, L4_MASK = 144 * C4;
constexpr int L4_O_MASK = 2 * L4_IMG, L4_O_FEAT = L4_O_MASK + 2 * L4_MASK;
constexpr int L4_O_BIAS = L4_O_FEAT + 2 * C4 * 4;
constexpr int L4_LDS_RUN = L4_O_BIAS + C4 * 4;
constexpr int CONV_LAST_WS_LDS = (Wmat<C3, C4>::BYTES > L4_LDS_RUN) ? Wmat<C3, C4>::BYTES : L4_LDS_RUN;

__global__ __launch_bounds__(NT, 2) void conv_last_fwd_ws_kernel(ConvLastWsParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using IM = Img<C3, 12, 12>;
  using WM = Wmat<C3, C4>;
  constexpr int NPIX = 144, NCONS = 6, NPT = (NW - NCONS) * 64;  // consumer waves; producer threads
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool consumer = wv < NCONS;
  // consumer roles.  Waves w and w + 4 of a workgroup share a SIMD (the hardware deals waves to SIMDs in a fixed cyclic order), so
  // waves 0 / 4 and 1 / 5 sit together while 2 and 3 each share with a producer: three channel groups over four SIMDs balance as
  //   wave:            0      1      2      3      4      5
  //   channel group:   0      1      0      1      2      2
  //   pixel tiles:   [0,2)  [0,2)  [2,9)  [2,9)  [0,4)  [4,9)      -> 6 / 7 / 7 / 7 tiles of 36 MFMAs per SIMD
  // (halves of the frame per group, 5 + 4 tiles, put 9 tiles on two of the SIMDs: 8.0 k cycles per frame in the stage timers)
  const int cg = wv < 4 ? (wv & 1) : 2;
  const int t0 = wv < 2 ? 0 : wv < 4 ? 2 : wv == 4 ? 0 : 4;
  const int ntile = wv < 2 ? 2 : wv < 4 ? 7 : wv == 4 ? 4 : 5;
  constexpr int MAXT = 7;

  // ---- the weights: staged through LDS once ([co][tap * 64 + ci] bf16), then this wave's 36 A fragments into registers
  stage_weights<C3, C4>(p.w, reinterpret_cast<bf16_t*>(smem), tid);
  __syncthreads();
  s16x8 wf[2][WM::KSTEPS];
  if (consumer) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < WM::KSTEPS; ++s)
        wf[j][s] = lds_frag(reinterpret_cast<const bf16_t*>(smem) + (32 * cg + 16 * j + li) * WM::LD + 32 * s + 8 * g);
  }
  __syncthreads();
  zero_lds(smem, L4_LDS_RUN, tid);
  bf16_t* img0 = reinterpret_cast<bf16_t*>(smem);
  uint8_t* msk0 = smem + L4_O_MASK;
  float* sf0 = reinterpret_cast<float*>(smem + L4_O_FEAT);
  float* s_bias = reinterpret_cast<float*>(smem + L4_O_BIAS);  // (an accumulator starts from its four channels' biases: one 16-byte read)
  __syncthreads();
  if (tid < C4) s_bias[tid] = p.b[tid];

  // this lane's pixel of each of its (up to 5) tiles: B-fragment base = pixel (y - 1, x - 1), channel chunk g
  int base[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    const int P = 16 * (t0 + t) + li, Pc = P < NPIX ? P : NPIX - 1;
    base[t] = IM::at(Pc / 12 - 1, Pc % 12 - 1) + 8 * g;
  }

  // producers: a frame is 144 pixels x 8 pieces of 16 bytes = 1152 pieces, 9 per producer thread
  // (in rounds of NRD pieces: nine quads of staging registers live across a pass beside the consumers' 144 weight registers spilled)
  constexpr int NLD = (NPIX * (C3 / 8) + NPT - 1) / NPT, NRD = SS_L4_NRD;
  static_assert(NLD % NRD == 0, "load rounds");
  const int pt = tid - NCONS * 64;
  auto copy_frame = [&](int n, bf16_t* img) {
    const uint4* src = reinterpret_cast<const uint4*>(p.in + (long)n * NPIX * C3);
#pragma unroll
    for (int rd = 0; rd < NLD / NRD; ++rd) {
      uint4 fr[NRD];
#pragma unroll
      for (int k = 0; k < NRD; ++k) fr[k] = src[pt + (rd * NRD + k) * NPT];
#pragma unroll
      for (int k = 0; k < NRD; ++k) {
        const int q = pt + (rd * NRD + k) * NPT, pix = q >> 3, c8 = q & 7;
        *reinterpret_cast<uint4*>(img + IM::at(pix / 12, pix % 12) + 8 * c8) = fr[k];
      }
    }
  };
  static_assert(NLD * NPT == NPIX * (C3 / 8), "a frame is a whole number of pieces per producer thread");
  __syncthreads();
  // prologue: frame 0 into image 0
  const int n0 = blockIdx.x, stride = gridDim.x;
  if (!consumer && n0 < p.N) copy_frame(n0, img0);
  __syncthreads();

  int it = 0;
  STAMP_ENTRY;
  STAMP_DECL;
  for (int n = n0; n < p.N + stride; n += stride, ++it) {  // one extra pass: the producers finish the last frame
    STAMP(15);
    const int cur = it & 1;
    if (consumer) {
      if (n < p.N) {
        const bf16_t* img = img0 + cur * (L4_IMG / 2);
        uint8_t* msk = msk0 + cur * L4_MASK;
        float fs[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
          if (t < ntile) {  // wave-uniform
            f32x4 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = *reinterpret_cast<const f32x4*>(s_bias + 32 * cg + 16 * j + 4 * g);
            // the pixel fragments run RD k steps ahead of their MFMAs behind scheduling fences: left alone hipcc reads a fragment,
            // waits lgkmcnt(0) and multiplies -- an LDS latency per pair of MFMAs, and a consumer wave has its SIMD (nearly) to itself
            constexpr int RD = 4;
            s16x8 fb[RD];
#pragma unroll
            for (int s = 0; s < RD; ++s) fb[s] = lds_frag(img + base[t] + koff<IM>(s, 0));
#pragma unroll
            for (int s = 0; s < WM::KSTEPS; ++s) {
              SS_SCHED_FENCE();
              acc[0] = mfma_bf16(wf[0][s], fb[s % RD], acc[0]);
              acc[1] = mfma_bf16(wf[1][s], fb[s % RD], acc[1]);
              SS_SCHED_FENCE();
              if (s + RD < WM::KSTEPS) fb[s % RD] = lds_frag(img + base[t] + koff<IM>(s + RD, 0));
            }
            // D row 4 g + r = channel, column li = pixel: four consecutive channels of one pixel per lane
            const int P = 16 * (t0 + t) + li;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              unsigned mb = 0;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float x = acc[j][r];
                mb |= (x > 0.f ? 1u : 0u) << (8 * r);
                fs[j][r] += fmaxf(x, 0.f);
              }
              *reinterpret_cast<unsigned*>(msk + P * C4 + 32 * cg + 16 * j + 4 * g) = mb;
            }
          }
        }
        // the average: sum over the 16 pixels of a lane row (DPP), the two pixel halves meet in LDS
        float* sf = sf0 + cur * C4;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = row_sum(fs[j][r]);
            if (li == 0) atomicAdd(sf + 32 * cg + 16 * j + 4 * g + r, v);
          }
      }
    } else {
      // the previous frame's mask and average leave (its buffers are this pass's `cur ^ 1`)
      if (it > 0) {
        const long np = n - stride;
        float* sf = sf0 + (cur ^ 1) * C4;
        if (p.mask) {
          const uint4* ms = reinterpret_cast<const uint4*>(msk0 + (cur ^ 1) * L4_MASK);
          uint4* dm = reinterpret_cast<uint4*>(p.mask + np * L4_MASK);
          for (int q = pt; q < L4_MASK / 16; q += NPT) dm[q] = ms[q];
        }
        if (pt < C4) {
          p.feat[np * C4 + pt] = sf[pt] * (1.0f / (float)NPIX);
          sf[pt] = 0.f;
        }
      }
      if (n + stride < p.N) copy_frame(n + stride, img0 + (cur ^ 1) * (L4_IMG / 2));
    }
    STAMP(0);  // (thread 0 is a consumer: its multiply phase, then its wait for the producers)
    __syncthreads();
    STAMP(1);
  }
  STAMP_FLUSH();
}

template <int CIN, int COUT, int H, int W, bool LAST>
constexpr int conv_fwd_lds() {
  return Img<CIN, H, W>::BYTES + round_up(Wmat<CIN, COUT>::BYTES, 16) +
         round_up(LAST ? H * W * COUT : (H / 2) * (W / 2) * COUT * 3, 16) + 16 + 2 * COUT * 4;
}

template <class P, class K>
int launch_persistent(K kernel, const P& p, int lds_bytes, int N, hipStream_t s) {
  if (lds_bytes > 160 * 1024) return SS_ERR_UNSUPPORTED;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
    return SS_ERR_LAUNCH;
  const int cap = ss_cnn_max_wgs > 0 ? ss_cnn_max_wgs : ss_device_cus();  // (the cap makes a test walk many frames per workgroup)
  const int grid = N < cap ? N : cap;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(NT), lds_bytes, s, p);
  return ss_launch_status();
}

// ------------------------------------------------------------------------------------------------ conv1 + conv2 fused
// The pooled conv1 map (73.7 KB per frame as bf16, plus 36.9 KB of argmax bytes) is the largest tensor of the net and was 43 % of
// the CNN's HBM traffic when every layer was its own kernel.  Here it is born in conv2's haloed LDS image and never leaves the
// CU: uint8 frame -> statistics -> bf16 image -> conv1 (patch GEMM) + ReLU + pool -> [LDS] -> conv2 + ReLU + pool -> a2, i2.
// The backward pass recomputes it (and conv1's pool winners) from the 9 KB frame: 576 MFMAs per frame (cnn_bf16_bwd.hip).
struct Conv12Params {
  const uint8_t* R;   // (N, 96, 96)
  int N, standardize;
  const float *w1, *b1, *w2, *b2;
  bf16_t* a2;         // (N, 24, 24, 32)
  uint8_t* i2;        // (N, 24, 24, 32)
  float* st;          // (N, 2) mean, std or null
  uint8_t* i1;        // (N, 48, 48, 16) conv1's pool winners, or null: 37 KB per frame for the fused conv2-dgrad / conv1-wgrad kernel,
                      // which spent 9.9 k of its 55 k cycles per frame recomputing them from the frame (stage timers, round 3)
};

__global__ __launch_bounds__(NT, 2) void conv12_fwd_kernel(Conv12Params p) {
  STAMP_ENTRY;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using IM = Img<C1, 48, 48>;
  using WM = Wmat<C1, C2>;
  constexpr int HO = 24, WO = 24, HH = 12;                     // conv2's pooled output, rows per flush
  bf16_t* img = reinterpret_cast<bf16_t*>(smem);               // [98][RS0]
  constexpr int o_tab = round_up(98 * RS0 * 2, 16);
  float* s_xn = reinterpret_cast<float*>(smem + o_tab);        // [256]
  float* s_misc = s_xn + 256;                                  // [64]
  constexpr int o_a1 = o_tab + (256 + 64) * 4;
  bf16_t* a1 = reinterpret_cast<bf16_t*>(smem + o_a1);         // Img<16,48,48>
  bf16_t* wl = reinterpret_cast<bf16_t*>(smem + o_a1 + IM::BYTES);
  constexpr int o_out = o_a1 + IM::BYTES + round_up(WM::BYTES, 16);
  bf16_t* oa = reinterpret_cast<bf16_t*>(smem + o_out);        // [HH][WO][32]
  uint8_t* oi = smem + o_out + HH * WO * C2 * 2;               // [HH][WO][32]
  float* s_bias = reinterpret_cast<float*>(smem + o_out + HH * WO * C2 * 3);  // [32]
  unsigned* s_red = reinterpret_cast<unsigned*>(s_misc);
  float* s_stat = s_misc + 32;

  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: as an SGPR the tile / row arithmetic that hangs on it runs on the scalar unit
  zero_lds(smem, o_tab, tid);
  zero_lds(a1, IM::BYTES, tid);
  stage_weights<C1, C2>(p.w2, wl, tid);
  if (tid < C2) s_bias[tid] = p.b2[tid];
  s16x8 bq[12];
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    const int oy = q / 6, ox = q % 6;
    s16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ky = g - oy, kx = j - ox;
      f[j] = (short)to_bf16((ky >= 0 && ky <= 2 && kx >= 0 && kx <= 2) ? p.w1[li * 9 + ky * 3 + kx] : 0.f);
    }
    bq[q] = f;
  }
  const f32x4 bias1 = *reinterpret_cast<const f32x4*>(p.b1 + 4 * g);  // channels 4g .. 4g+3 (conv1_rows)
  const int chunk = g & 1, hi = g >> 1;
  int lane_s[WM::KSTEPS];  // conv2: this lane's pixel of a 2 x 8 tile + its k chunk + the tap of k step s
#pragma unroll
  for (int s = 0; s < WM::KSTEPS; ++s)
    lane_s[s] = ((li >> 1) & 1) * IM::RS + (2 * (li >> 2) + (li & 1)) * IM::PS + 8 * chunk + (hi ? koff<IM>(s, 1) : koff<IM>(s, 0));
  const int lane_o = g * C2 + li;  // conv2's staged output: window g of a tile, channel li (+ 16 per n tile)
  uint4 px[2];
  auto load_px = [&](int n) {
    const uint4* src = reinterpret_cast<const uint4*>(p.R + (long)n * HW0 * HW0);
    px[0] = src[tid];
    px[1] = (tid + NT < HW0 * HW0 / 16) ? src[tid + NT] : uint4{0u, 0u, 0u, 0u};
  };
  if ((int)blockIdx.x < p.N) load_px(blockIdx.x);
  __syncthreads();
  STAMP_DECL;

  // A half's pooled rows leave for HBM.  The second half's copy is deferred to the top of the NEXT frame, behind the first use of that
  // frame's prefetched bytes: vector memory operations retire in order and hipcc cannot count stores issued in a loop, so the wait
  // for the (older) prefetch is vmcnt(0) -- with the copy-out in front of it that wait sat through the stores' round trip.
  auto copy_half = [&](int n, int half) {
    uint4* da = reinterpret_cast<uint4*>(p.a2 + ((long)n * HO + half * HH) * WO * C2);
    for (int q = tid; q < HH * WO * C2 * 2 / 16; q += NT) da[q] = reinterpret_cast<const uint4*>(oa)[q];
    uint4* di = reinterpret_cast<uint4*>(p.i2 + ((long)n * HO + half * HH) * WO * C2);
    for (int q = tid; q < HH * WO * C2 / 16; q += NT) di[q] = reinterpret_cast<const uint4*>(oi)[q];
  };
  int n_pending = -1;
  for (int n = blockIdx.x; n < p.N; n += gridDim.x) {
    STAMP(15);
    // ---- statistics, table, bf16 image (as conv1_fwd_kernel)
    unsigned su = 0, sq = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const unsigned wds[4] = {px[k].x, px[k].y, px[k].z, px[k].w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const unsigned u = (wds[e] >> (8 * b)) & 255u;
          su += u;
          sq += u * u;
        }
    }
    su = wave_sum_u32(su);
    sq = wave_sum_u32(sq);
    if (lane == 0) { s_red[2 * wv] = su; s_red[2 * wv + 1] = sq; }
    if (n_pending >= 0) {  // the previous frame's second half (its staging area is rewritten three barriers further on)
      copy_half(n_pending, 1);
      n_pending = -1;
    }
    __syncthreads();
    if (tid == 0) {
      unsigned long long tsu = 0, tsq = 0;
      for (int k = 0; k < NW; ++k) { tsu += s_red[2 * k]; tsq += s_red[2 * k + 1]; }
      float mu = 0.f, sd = 1.f;
      if (p.standardize) {
        const double nn = (double)(HW0 * HW0);
        mu = (float)((double)tsu / nn) / 255.0f;
        const double var = ((double)tsq - (double)tsu * (double)tsu / nn) / (nn - 1.0);
        sd = fmaxf((float)(sqrt(var > 0.0 ? var : 0.0) / 255.0), 1e-6f);
      }
      s_stat[0] = mu;
      s_stat[1] = sd;
      if (p.st) { p.st[2 * (long)n] = mu; p.st[2 * (long)n + 1] = sd; }
    }
    __syncthreads();
    if (tid < 256) {
      const float rr = (float)tid / 255.0f;
      s_xn[tid] = p.standardize ? (rr - s_stat[0]) / s_stat[1] : rr;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int q = tid + k * NT;
      if (q < HW0 * HW0 / 16) {
        const int lin = q * 16;
        bf16_t* dst = img + (lin / HW0 + 1) * RS0 + (lin % HW0) + 1;
        const unsigned wds[4] = {px[k].x, px[k].y, px[k].z, px[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int b = 0; b < 4; ++b) dst[4 * e + b] = to_bf16(s_xn[(wds[e] >> (8 * b)) & 255u]);
      }
    }
    __syncthreads();
    STAMP(0);
    if (n + (int)gridDim.x < p.N) load_px(n + gridDim.x);  // the next frame's bytes, under both convolutions
    // ---- conv1 -> conv2's haloed input image
    if (p.i1)  // (wave-uniform; the winner bytes of a lane's four channels leave as one 4-byte global store)
      conv1_rows(img, [&](int q) { return bq[q]; }, bias1, 0, 48, 0, a1, IM::at(0, 0), IM::RS, IM::PS, p.i1 + (long)n * 48 * 48 * C1, wv, g, li);
    else
      conv1_rows(img, [&](int q) { return bq[q]; }, bias1, 0, 48, 0, a1, IM::at(0, 0), IM::RS, IM::PS, nullptr, wv, g, li);
    STAMP(1);
    __syncthreads();
    STAMP(2);
    // ---- conv2 in two halves of 12 pooled rows (the staging area holds one)
    for (int half = 0; half < 2; ++half) {
      constexpr int MT = 3, UNITS = HH * 6 / MT;  // 72 m tiles of 2 rows x 8 columns per half
      for (int u = wv; u < UNITS; u += NW) {
        // a unit = three neighbouring m tiles = half a pooled row: (pooled row, first tile) are SCALAR (wv is an SGPR), the lane's
        // place inside a 2 x 8 tile, its k chunk and the tap of every k step are loop-invariant (lane_s[], made once per kernel):
        // an operand address is scalar + lane_s[s] + an immediate per tile.  (The pixel -> address arithmetic per tile and step
        // was 54 of the unit's 171 vector instructions beside its 30 MFMAs; the kernel is bound by the vector issue port.)
        const int ypl = u >> 1, xt0 = 3 * (u & 1);                 // pooled row inside the half, first m tile of the unit
        const int ubase = IM::at(2 * (half * HH + ypl) - 1, 8 * xt0 - 1);
        f32x4 acc[MT][2];
#pragma unroll
        for (int a = 0; a < MT; ++a) acc[a][0] = acc[a][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < WM::KSTEPS; ++s) {
          s16x8 fa[MT], fb[2];
          const bf16_t* ap = a1 + ubase + lane_s[s];
#pragma unroll
          for (int a = 0; a < MT; ++a) fa[a] = lds_frag(ap + a * 8 * IM::PS);
#pragma unroll
          for (int b = 0; b < 2; ++b) fb[b] = lds_frag(wl + (16 * b + li) * WM::LD + 32 * s + 8 * g);
#pragma unroll
          for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = mfma_bf16(fa[a], fb[b], acc[a][b]);
        }
        bf16_t* oap = oa + (ypl * WO + 4 * xt0) * C2 + lane_o;
        uint8_t* oip = oi + (ypl * WO + 4 * xt0) * C2 + lane_o;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const float bias = s_bias[16 * b + li];
#pragma unroll
          for (int a = 0; a < MT; ++a) {
            const f32x4 v = acc[a][b];
            float best = v[0];
            int bi = 0;
            if (v[1] > best) { best = v[1]; bi = 1; }
            if (v[2] > best) { best = v[2]; bi = 2; }
            if (v[3] > best) { best = v[3]; bi = 3; }
            const float x = fmaxf(best + bias, 0.f);
            oap[a * 4 * C2 + 16 * b] = to_bf16(x);
            oip[a * 4 * C2 + 16 * b] = (uint8_t)(x > 0.f ? bi : IDX_DEAD);
          }
        }
      }
      STAMP(3);
      __syncthreads();
      STAMP(4);
      if (half == 0) {
        copy_half(n, 0);
        STAMP(5);
        __syncthreads();  // the second half's epilogues rewrite the staging area
        STAMP(6);
      } else {
        n_pending = n;  // leaves at the top of the next frame, behind the wait for that frame's prefetched bytes (see copy_half)
      }
    }
  }
  if (n_pending >= 0) copy_half(n_pending, 1);
  STAMP_FLUSH();
}

constexpr int CONV12_LDS = round_up(98 * RS0 * 2, 16) + (256 + 64) * 4 + Img<C1, 48, 48>::BYTES + round_up(Wmat<C1, C2>::BYTES, 16) +
                           12 * 24 * C2 * 3 + C2 * 4;

}  // namespace

extern "C" int ss_c5_conv12_fwd_i1(const uint8_t* R, int N, int standardize, const float* w1, const float* b1, const float* w2,
                                   const float* b2, uint16_t* a2, uint8_t* i2, float* st, uint8_t* i1, ss_stream_t stream) {
  SS_REQUIRE(R && w1 && b1 && w2 && b2 && a2 && i2 && N > 0, SS_ERR_ARG);
  SS_REQUIRE((reinterpret_cast<uintptr_t>(i1) & 15) == 0, SS_ERR_ARG);
  Conv12Params p{R, N, standardize, w1, b1, w2, b2, a2, i2, st, i1};
  return launch_persistent(conv12_fwd_kernel, p, CONV12_LDS, N, static_cast<hipStream_t>(stream));
}
extern "C" int ss_c5_conv12_fwd(const uint8_t* R, int N, int standardize, const float* w1, const float* b1, const float* w2,
                                const float* b2, uint16_t* a2, uint8_t* i2, float* st, ss_stream_t stream) {
  return ss_c5_conv12_fwd_i1(R, N, standardize, w1, b1, w2, b2, a2, i2, st, nullptr, stream);
}

static const bool ss_c5_last_ws = !(getenv("SS_C5_LAST_WS") && getenv("SS_C5_LAST_WS")[0] == '0');

// layer 4 without the Linear: feat (N, 96) f32 = global average of ReLU(conv4), mask (N,144,96) u8 or null
extern "C" int ss_c5_conv_last_fwd_feat(const uint16_t* in, int N, const float* w, const float* b, uint8_t* mask, float* feat,
                                        ss_stream_t stream) {
  SS_REQUIRE(in && w && b && feat && N > 0, SS_ERR_ARG);
  if (ss_c5_last_ws) {  // weight-stationary form (SS_C5_LAST_WS=0: the LDS-resident weights of conv_fwd_kernel)
    ConvLastWsParams q{in, N, w, b, mask, feat};
    return launch_persistent(conv_last_fwd_ws_kernel, q, CONV_LAST_WS_LDS, N, static_cast<hipStream_t>(stream));
  }
  ConvFwdParams p{};
  p.in = in; p.N = N; p.w = w; p.b = b; p.mask = mask; p.feat = feat;
  return launch_persistent(conv_fwd_kernel<C3, C4, 12, 12, true, 3, 3>, p, conv_fwd_lds<C3, C4, 12, 12, true>(), N,
                           static_cast<hipStream_t>(stream));
}

extern "C" int ss_c5_conv1_fwd(const uint8_t* R, int N, int standardize, const float* w1, const float* b1, uint16_t* a1,
                               uint8_t* i1, float* st, ss_stream_t stream) {
  SS_REQUIRE(R && w1 && b1 && a1 && i1 && N > 0, SS_ERR_ARG);
  Conv1Params p{R, N, standardize, w1, b1, a1, i1, st};
  return launch_persistent(conv1_fwd_kernel, p, CONV1_LDS, N, static_cast<hipStream_t>(stream));
}

// layer = 2: (N,48,48,16) -> (N,24,24,32);  layer = 3: (N,24,24,32) -> (N,12,12,64)
extern "C" int ss_c5_conv_fwd(int layer, const uint16_t* in, int N, const float* w, const float* b, uint16_t* out, uint8_t* idx,
                              ss_stream_t stream) {
  SS_REQUIRE(in && w && b && out && idx && N > 0, SS_ERR_ARG);
  ConvFwdParams p{};
  p.in = in; p.N = N; p.w = w; p.b = b; p.out = out; p.idx = idx;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (layer == 2) return launch_persistent(conv_fwd_kernel<C1, C2, 48, 48, false, 3, 2>, p, conv_fwd_lds<C1, C2, 48, 48, false>(), N, s);
  if (layer == 3) return launch_persistent(conv_fwd_kernel<C2, C3, 24, 24, false, 9, 2>, p, conv_fwd_lds<C2, C3, 24, 24, false>(), N, s);
  return SS_ERR_UNSUPPORTED;
}

extern "C" int ss_c5_conv_last_fwd(const uint16_t* in, int N, const float* w, const float* b, const float* wfc, const float* bfc,
                                   int E, float* z, int ld_z, uint8_t* mask, float* feat, ss_stream_t stream) {
  SS_REQUIRE(in && w && b && wfc && bfc && z && N > 0 && E > 0 && ld_z >= E, SS_ERR_ARG);
  SS_REQUIRE((mask == nullptr) == (feat == nullptr), SS_ERR_ARG);
  ConvFwdParams p{};
  p.in = in; p.N = N; p.w = w; p.b = b; p.wfc = wfc; p.bfc = bfc; p.E = E; p.z = z; p.ld_z = ld_z; p.mask = mask; p.feat = feat;
  return launch_persistent(conv_fwd_kernel<C3, C4, 12, 12, true, 3, 3>, p, conv_fwd_lds<C3, C4, 12, 12, true>(), N,
                           static_cast<hipStream_t>(stream));
}
