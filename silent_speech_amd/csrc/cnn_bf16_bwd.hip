// bf16 ROI-CNN backward of BASELINE config 5 (geometry: cnn_bf16.h; forward: cnn_bf16.hip).
//
// Per layer L (input a_in with CIN channels at H x W, output COUT channels) two persistent kernels:
//   wgrad  d W[co][ci][tap] = sum_pixels dy[p][co] * a_in[p + tap][ci]   M = co, N = ci, K = 32 pixels per MFMA.
//          Both operands are pixel-major in LDS (the contraction index is the SLOW one), so both fragments come from
//          ds_read_b64_tr_b16: every lane hands in the address of one pixel's 4-channel piece, the hardware
//          transposes.  A pixel's address is computed per lane, so the 32 pixels of a k step need no common stride
//          (rows of 12, 24 or 48 pixels, haloed on the a_in side).  The 9 taps share the dy fragment; the weight
//          gradient lives in registers for the whole walk over the frames and leaves as float atomics once.
//          d b[co] = sum_pixels dy[p][co] rides on the same data path (one more MFMA per k step against a fragment
//          of ones).
//   dgrad  d a_in = conv_transpose(dy, W): the forward routine on the haloed dy image with the flipped, transposed
//          weights [ci][tap'][co].  Stored UNMASKED: the ReLU of the layer below is applied when ITS backward
//          expands the gradient through its pool (argmax byte 4 = dead window).
// dy is rebuilt in LDS from the pooled-grid gradient + the argmax bytes (pooled layers) or from d feat + the sign mask
// (last layer).  Large maps are processed in row bands so that dy (dense, bf16) fits beside the other operand.
//   conv1 wgrad: 1 input channel -- a GEMM on the pooled grid against the 4 x 4 input patch under every pool window.
#include <stdlib.h>

#include "cnn_bf16.h"

extern int ss_cnn_max_wgs;  // roi_cnn.hip: test hook, workgroups per launch (0 = one per CU)

namespace {
using namespace c5;

STAMP_TABLE(ss_debug_stamps_c5_bwd)

__device__ __forceinline__ void zero_lds(void* base, int bytes, int tid) {
  uint4* p = reinterpret_cast<uint4*>(base);
  for (int q = tid; q < bytes / 16; q += NT) p[q] = uint4{0u, 0u, 0u, 0u};
}

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__device__ __forceinline__ s16x8 tr_pair(const bf16_t* lo, const bf16_t* hi) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)lo);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)hi);
  return s16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

struct ConvBwdParams {
  int N;
  const bf16_t* a_in;     // (N, H, W, CIN) input of the layer (forward stash)
  const bf16_t* da_out;   // (N, H/2, W/2, COUT) gradient w.r.t. the pooled output (unmasked)      pooled layers
  const uint8_t* idx;     // (N, H/2, W/2, COUT) argmax bytes
  // last layer: d z (N, E) at ld_dz, fc weights, sign mask, averaged features
  const float* dz; int ld_dz, E;
  const float* wfc;
  const uint8_t* mask;    // (N, H*W, COUT)
  const float* feat;      // (N, COUT)
  const float* dfeat;     // (N, COUT) d z . W_fc of every frame (a GEMM in front of the kernels), or null: made per frame from dz, wfc
  const float* w;         // (COUT, CIN, 3, 3) f32 (dgrad)
  float *g_w, *g_b;       // gradient accumulators (wgrad), +=
  float *g_wfc, *g_bfc;   // (E, COUT), (E)  (last layer's wgrad)
  float* part;            // wgrad: null, or gridDim.x x (COUT*CIN*9) floats -- every workgroup leaves its weight-gradient sums there
                          // with plain stores and ss_c5 wgrad_reduce folds them into g_w (instead of float atomics: below)
  bf16_t* da_in;          // (N, H, W, CIN) (dgrad)
  // recompute form of layer 2's weight gradient (RC): a_in is not stored, it is rebuilt from the frame
  const uint8_t* R;       // (N, 96, 96)
  const float* st;        // (N, 2) mean, std
  int standardize;
  const float *w1, *b1;   // conv1
};

// dense dy rows [y0, y0 + rows) of a pooled layer into an LDS image whose local row 0 is y0 (rows outside the frame: zeros)
//   dst(yl, x) = img + off0 + yl * RS + x * PS
template <int COUT, int H, int W>
__device__ __forceinline__ void expand_dy(const bf16_t* __restrict__ da, const uint8_t* __restrict__ idx, bf16_t* img, int off0, int RS,
                                          int PS, int y0, int rows, int tid) {
  constexpr int WO = W / 2, CH = COUT / 8;
  for (int q = tid; q < rows * WO * CH; q += NT) {
    const int c8 = q % CH, xp = (q / CH) % WO, yl = q / (CH * WO);
    const int y = y0 + yl;
    uint4 v0 = {0u, 0u, 0u, 0u}, v1 = v0;
    if (y >= 0 && y < H) {
      const long src = ((long)(y >> 1) * WO + xp) * COUT + 8 * c8;
      const uint4 dv = *reinterpret_cast<const uint4*>(da + src);
      const uint2 iv = *reinterpret_cast<const uint2*>(idx + src);
      const unsigned d[4] = {dv.x, dv.y, dv.z, dv.w};
      const unsigned e0 = (unsigned)(y & 1) * 2u;
      unsigned o0[4], o1[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // channels 2k, 2k+1
        const unsigned ib = (k < 2 ? iv.x : iv.y) >> (16 * (k & 1));
        const unsigned ia = ib & 255u, ic = (ib >> 8) & 255u;
        const unsigned lo = d[k] & 0xffffu, hi = d[k] & 0xffff0000u;
        o0[k] = (ia == e0 ? lo : 0u) | (ic == e0 ? hi : 0u);
        o1[k] = (ia == e0 + 1u ? lo : 0u) | (ic == e0 + 1u ? hi : 0u);
      }
      v0 = uint4{o0[0], o0[1], o0[2], o0[3]};
      v1 = uint4{o1[0], o1[1], o1[2], o1[3]};
    }
    bf16_t* dst = img + off0 + yl * RS + (2 * xp) * PS + 8 * c8;
    *reinterpret_cast<uint4*>(dst) = v0;
    *reinterpret_cast<uint4*>(dst + PS) = v1;
  }
}

// last layer: dy[P][co] = mask[P][co] ? dfeat[co] / (H*W) : 0, all pixels
template <int COUT, int NPIX>
__device__ __forceinline__ void expand_dy_last(const uint8_t* __restrict__ mask, const float* s_dfeat, bf16_t* img, int off0, int RS,
                                               int PS, int W, int tid) {
  constexpr int CH = COUT / 8;
  for (int q = tid; q < NPIX * CH; q += NT) {
    const int c8 = q % CH, P = q / CH;
    const uint2 mv = *reinterpret_cast<const uint2*>(mask + (long)P * COUT + 8 * c8);
    unsigned o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned mb = (k < 2 ? mv.x : mv.y) >> (16 * (k & 1));
      const float a = (mb & 255u) ? s_dfeat[8 * c8 + 2 * k] : 0.f, b = ((mb >> 8) & 255u) ? s_dfeat[8 * c8 + 2 * k + 1] : 0.f;
      o[k] = pack_bf16(a, b);
    }
    *reinterpret_cast<uint4*>(img + off0 + (P / W) * RS + (P % W) * PS + 8 * c8) = uint4{o[0], o[1], o[2], o[3]};
  }
}

// rows [y0, y0 + rows) of an NHWC frame into a haloed LDS image whose local row -1 is y0 (frame rows outside [0,H): zeros)
template <class IM, int HF>
__device__ __forceinline__ void load_band(const bf16_t* __restrict__ src, bf16_t* img, int y0, int rows, int tid) {
  constexpr int CH = IM::C / 8;
  for (int q = tid; q < rows * IM::W * CH; q += NT) {
    const int c8 = q % CH, x = (q / CH) % IM::W, yl = q / (CH * IM::W);
    const int y = y0 + yl;
    uint4 v = {0u, 0u, 0u, 0u};
    if (y >= 0 && y < HF) v = *reinterpret_cast<const uint4*>(src + ((long)y * IM::W + x) * IM::C + 8 * c8);
    *reinterpret_cast<uint4*>(img + IM::at(yl - 1, x) + 8 * c8) = v;
  }
}

// d feat[co] = sum_e dz[e] wfc[e][co] / (H*W)  (gradient of Linear o global average)
template <int COUT>
__device__ __forceinline__ void last_dfeat(const ConvBwdParams& p, int n, float* s_dz, float* s_dfeat, float inv_hw, int tid) {
  for (int e = tid; e < p.E; e += NT) s_dz[e] = p.dz[(long)n * p.ld_dz + e];
  __syncthreads();
  for (int c = tid; c < COUT; c += NT) {
    float s = 0.f;
    for (int e = 0; e < p.E; ++e) s += s_dz[e] * p.wfc[e * COUT + c];
    s_dfeat[c] = s * inv_hw;
  }
}

// ================================================================================================ wgrad
// BH = rows of a band; WCO x WCI x WK = 8 waves: a wave owns COUT/16/WCO co tiles x CIN/16/WCI ci tiles (all 9 taps) and
// every WK-th 32-pixel k step
// RC (layer 2 only): the layer's input a1 = pool(ReLU(conv1(frame))) is not read from HBM but recomputed per band from the 9 KB
// uint8 frame (conv1 patch GEMM of cnn_bf16.hip: 12 MFMAs per row pair) -- it was the largest tensor of the net.
// MINW: waves per SIMD the register allocation must allow (2 = one workgroup per CU, 4 = two)
// FCX (last layer): d feat arrives ready-made (p.dfeat) and the fc gradients are somebody else's GEMMs.  The stage timers put
// the per-frame "d z -> d feat (96 threads walking 64 rows of W_fc in global memory), fc gradients, dy image" stage at 19.9 k of
// the kernel's 30.4 k cycles per frame: as three small GEMMs over all frames that work leaves the kernel (and its 12 registers
// of fc partial sums leave an instantiation that sat at 256 registers and spilled).
template <int CIN, int COUT, int H, int W, bool LAST, int BH, int WCO, int WCI, int WK, bool RC = false, int MINW = 2, bool FCX = false>
__global__ __launch_bounds__(NT, MINW) void conv_wgrad_kernel(ConvBwdParams p) {
  static_assert(WCO * WCI * WK == NW && H % BH == 0, "wave split");
  static_assert(!RC || (CIN == C1 && H == 48 && W == 48 && !LAST), "recompute form: layer 2");
  STAMP_ENTRY;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using IA = Img<CIN, BH, W, true>;
  constexpr int PSD = COUT + TR_PAD, RSD = W * PSD;      // dense dy band, no halo
  constexpr int DY_BYTES = BH * RSD * 2;
  constexpr int NCO = COUT / 16 / WCO, NCI = CIN / 16 / WCI;
  constexpr int NPIX = BH * W, NCH = (NPIX + 31) / 32;
  bf16_t* dyi = reinterpret_cast<bf16_t*>(smem);
  bf16_t* ai = reinterpret_cast<bf16_t*>(smem + DY_BYTES);
  constexpr int ZB = 256;  // a pixel's worth of zeros (COUT <= 96 channels + the lane's piece): what pixels past the band read
  float* s_dz = reinterpret_cast<float*>(smem + DY_BYTES + IA::BYTES + ZB);  // [64]
  float* s_dfeat = s_dz + 64;                                                // [COUT]
  float* s_feat = s_dfeat + 96;                                              // [COUT]
  // RC: normalised bf16 frame [98][RS0], grey-level table, the 12 conv1 B fragments as [q][lane][8]
  constexpr int o_rc = DY_BYTES + IA::BYTES + ZB + (64 + 96 + 96) * 4;
  bf16_t* ximg = reinterpret_cast<bf16_t*>(smem + o_rc);
  float* s_xn = reinterpret_cast<float*>(smem + o_rc + round_up(98 * RS0 * 2, 16));
  bf16_t* bqs = reinterpret_cast<bf16_t*>(s_xn + 256);

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, li = lane & 15;
  // (wv as an SGPR -- readfirstlane, as in the forward kernels -- was measured here: the last layer's weight gradient 0.32 -> 0.47 ms)
  const int q4 = li >> 2, p4 = li & 3;
  const int wco = wv % WCO, wci = (wv / WCO) % WCI, wk = wv / (WCO * WCI);
  f32x4 bias1 = {0.f, 0.f, 0.f, 0.f};
  uint4 px[2];
  auto load_px = [&](int n) {
    const uint4* src = reinterpret_cast<const uint4*>(p.R + (long)n * HW0 * HW0);
    px[0] = src[tid];
    px[1] = (tid + NT < HW0 * HW0 / 16) ? src[tid + NT] : uint4{0u, 0u, 0u, 0u};
  };
  if (RC) {
    zero_lds(ximg, round_up(98 * RS0 * 2, 16), tid);
    if (wv == 0)
#pragma unroll
      for (int q = 0; q < 12; ++q) *reinterpret_cast<s16x8*>(bqs + (q * 64 + lane) * 8) = conv1_bfrag(p.w1, q, g, li);
    bias1 = *reinterpret_cast<const f32x4*>(p.b1 + 4 * g);  // channels 4g .. 4g+3 (conv1_rows)
    if ((int)blockIdx.x < p.N) load_px(blockIdx.x);
  }
  static_assert((COUT + TR_PAD) * 2 <= ZB, "zero pixel");
  zero_lds(smem, DY_BYTES + IA::BYTES + ZB, tid);

  f32x4 acc[NCO][NCI][9];
  f32x4 accb[NCO];
#pragma unroll
  for (int a = 0; a < NCO; ++a) {
    accb[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NCI; ++b)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[a][b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const s16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  // fc gradients of the last layer: thread-private partial sums over the frames
  constexpr int FCN = (LAST && !FCX) ? (64 * COUT + NT - 1) / NT : 1;
  float fcw[FCN];
  float fcb = 0.f;
#pragma unroll
  for (int k = 0; k < FCN; ++k) fcw[k] = 0.f;
  __syncthreads();

  // This lane's operand addresses of its k steps (two pixels each: rows q4 and q4 + 4 of the lane's 8-pixel group) depend on
  // nothing but the lane and the k step -- the dy and input images are band-local -- so they are made ONCE: the loop below had
  // ~110 vector instructions of pixel -> (row, column) -> address arithmetic per k step beside its 20 - 30 MFMAs, and these
  // kernels are bound by the vector issue port (docs/LAB_NOTES.md 8c)
  // (only where a wave has few k steps per band -- layer 2: three; with nine (layer 3) or five (layer 4) offset sets the register
  // allocation spilled)
  constexpr int NCHW = (NCH + WK - 1) / WK;
  constexpr bool PRE = RC && NCHW <= 3;
  const bf16_t* lds_b = reinterpret_cast<const bf16_t*>(smem);
  const bf16_t* zeros = reinterpret_cast<const bf16_t*>(smem + DY_BYTES + IA::BYTES);
  int o_dy0[PRE ? NCHW : 1], o_dy1[PRE ? NCHW : 1], o_a0[PRE ? NCHW : 1], o_a1[PRE ? NCHW : 1];
  if constexpr (PRE) {
    const int z_off = (DY_BYTES + IA::BYTES) / 2, a_off = DY_BYTES / 2;  // element offsets of `zeros` and `ai`
#pragma unroll
    for (int k = 0; k < NCHW; ++k) {
      const int ch = wk + k * WK;
      const int P0 = 32 * ch + 8 * g + q4, P1 = P0 + 4;
      const bool in0 = P0 < NPIX, in1 = P1 < NPIX;
      const int y0l = P0 / W, x0l = P0 % W, y1l = P1 / W, x1l = P1 % W;
      o_dy0[k] = (in0 ? y0l * RSD + x0l * PSD : z_off) + 4 * p4;
      o_dy1[k] = (in1 ? y1l * RSD + x1l * PSD : z_off) + 4 * p4;
      o_a0[k] = a_off + (in0 ? IA::at(y0l - 1, x0l - 1) : IA::at(0, 0)) + 4 * p4;
      o_a1[k] = a_off + (in1 ? IA::at(y1l - 1, x1l - 1) : IA::at(0, 0)) + 4 * p4;
    }
  }
  // units = (frame, band), band-minor; the operands of unit u + 1 are in flight (registers) while unit u is multiplied
  ExpandLoad<COUT, H, W, BH> pe;
  MaskLoad<COUT, H * W> pm;
  BandLoad<IA, H, BH + 2> pa;
  auto issue = [&](int n, int y0) {
    if (LAST) {
      if (FCX) pm.issue(p.mask + (long)n * H * W * COUT, nullptr, 0, nullptr, tid, p.dfeat + (long)n * COUT);
      else pm.issue(p.mask + (long)n * H * W * COUT, p.dz + (long)n * p.ld_dz, p.E, p.feat + (long)n * COUT, tid);
    } else {
      pe.issue(p.da_out + (long)n * (H / 2) * (W / 2) * COUT, p.idx + (long)n * (H / 2) * (W / 2) * COUT, y0, tid);
    }
    if (!RC) pa.issue(p.a_in + (long)n * H * W * CIN, y0 - 1, tid);
  };
  if ((int)blockIdx.x < p.N) issue(blockIdx.x, 0);
  STAMP_DECL;
  for (int n = blockIdx.x; n < p.N; n += gridDim.x) {
    STAMP(15);
    if (RC) {  // the frame's normalised image (statistics from the forward pass), once per frame
      if (tid < 256) {
        const float rr = (float)tid / 255.0f;
        s_xn[tid] = p.standardize ? (rr - p.st[2 * (long)n]) / p.st[2 * (long)n + 1] : rr;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int q = tid + k * NT;
        if (q < HW0 * HW0 / 16) {
          const int lin = q * 16;
          bf16_t* dst = ximg + (lin / HW0 + 1) * RS0 + (lin % HW0) + 1;
          const unsigned wds[4] = {px[k].x, px[k].y, px[k].z, px[k].w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int b = 0; b < 4; ++b) dst[4 * e + b] = to_bf16(s_xn[(wds[e] >> (8 * b)) & 255u]);
        }
      }
      if (n + (int)gridDim.x < p.N) load_px(n + gridDim.x);
      __syncthreads();
    }
    STAMP(0);
    for (int y0 = 0; y0 < H; y0 += BH) {
      if (LAST && FCX) {
        if (tid < COUT) s_dfeat[tid] = pm.df * (1.0f / (float)(H * W));
        __syncthreads();
        pm.commit(s_dfeat, dyi, 0, RSD, PSD, W, tid);
      } else if (LAST) {
        if (tid < p.E) s_dz[tid] = pm.dz;
        if (tid < COUT) s_feat[tid] = pm.ft;
        __syncthreads();
        for (int c = tid; c < COUT; c += NT) {  // d feat[co] = sum_e dz[e] wfc[e][co] / (H*W)
          float s = 0.f;
          for (int e = 0; e < p.E; ++e) s += s_dz[e] * p.wfc[e * COUT + c];
          s_dfeat[c] = s * (1.0f / (float)(H * W));
        }
#pragma unroll
        for (int k = 0; k < FCN; ++k) {
          const int q = tid + k * NT;
          if (q < p.E * COUT) fcw[k] += s_dz[q / COUT] * s_feat[q % COUT];
        }
        if (tid < p.E) fcb += s_dz[tid];
        __syncthreads();
        pm.commit(s_dfeat, dyi, 0, RSD, PSD, W, tid);
      } else {
        pe.commit(dyi, 0, RSD, PSD, y0, tid);
      }
      STAMP(1);
      if (RC) {
        // a1 rows y0-1 .. y0+BH of the band: recomputed inside the frame, zero outside it
        // (rows y0-1 and y0 ARE rows BH-1 and BH of the band before: moving them instead of recomputing them -- 16 rows = two per wave
        // instead of 18 -- was built and measured: 377.5 -> 407.7 us per launch, the two waves that move rows hold the stage up)
        const int yp0 = y0 > 0 ? y0 - 1 : 0, yp1 = y0 + BH + 1 < H ? y0 + BH + 1 : H;
        conv1_rows(ximg, [&](int q) { return lds_frag(bqs + (q * 64 + lane) * 8); }, bias1, yp0, yp1, y0, ai, IA::at(0, 0), IA::RS,
                   IA::PS, nullptr, wv, g, li);
        if (y0 == 0)
          for (int q = tid; q < W * CIN / 8; q += NT) *reinterpret_cast<uint4*>(ai + IA::at(-1, 0) + 8 * q) = uint4{0u, 0u, 0u, 0u};
        if (y0 + BH == H)
          for (int q = tid; q < W * CIN / 8; q += NT) *reinterpret_cast<uint4*>(ai + IA::at(BH, 0) + 8 * q) = uint4{0u, 0u, 0u, 0u};
      } else {
        pa.commit(ai, -1, tid);
      }
      STAMP(2);
      __syncthreads();
      STAMP(3);
      {  // the next unit's loads fly under the MFMAs below
        const bool last_band = y0 + BH >= H;
        const int nn = last_band ? n + (int)gridDim.x : n;
        if (nn < p.N) issue(nn, last_band ? 0 : y0 + BH);
      }
      STAMP(4);
      auto k_step = [&](const bf16_t* dy0, const bf16_t* dy1, const bf16_t* a0, const bf16_t* a1) {
        s16x8 fa[NCO];
#pragma unroll
        for (int a = 0; a < NCO; ++a) {
          const int co0 = 16 * (wco * NCO + a);
          fa[a] = tr_pair(dy0 + co0, dy1 + co0);
          if (wci == 0) accb[a] = mfma_bf16(fa[a], ones, accb[a]);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int toff = (t / 3) * IA::RS + (t % 3) * IA::PS;
#pragma unroll
          for (int b = 0; b < NCI; ++b) {
            const int ci0 = 16 * (wci * NCI + b);
            const s16x8 fb = tr_pair(a0 + toff + ci0, a1 + toff + ci0);
#pragma unroll
            for (int a = 0; a < NCO; ++a) acc[a][b][t] = mfma_bf16(fa[a], fb, acc[a][b][t]);
          }
        }
      };
      if constexpr (PRE) {
#pragma unroll
        for (int k = 0; k < NCHW; ++k) {
          if (wk + k * WK >= NCH) break;  // wave-uniform
          k_step(lds_b + o_dy0[k], lds_b + o_dy1[k], lds_b + o_a0[k], lds_b + o_a1[k]);
        }
      } else {
        // (carrying the two pixels' (row, column) along with an add and a conditional carry instead of dividing per step was
        // measured: layer 3 0.251 -> 0.259 ms, layer 4 0.326 -> 0.482 -- that instantiation sits at 256 registers and spills)
        for (int ch = wk; ch < NCH; ch += WK) {
          // this lane's two pixels of the k step: rows q4 and q4 + 4 of its 8-pixel group
          const int P0 = 32 * ch + 8 * g + q4, P1 = P0 + 4;
          const bool in0 = P0 < NPIX, in1 = P1 < NPIX;
          const int y0l = P0 / W, x0l = P0 % W, y1l = P1 / W, x1l = P1 % W;
          k_step(in0 ? dyi + y0l * RSD + x0l * PSD + 4 * p4 : zeros + 4 * p4, in1 ? dyi + y1l * RSD + x1l * PSD + 4 * p4 : zeros + 4 * p4,
                 ai + (in0 ? IA::at(y0l - 1, x0l - 1) : IA::at(0, 0)) + 4 * p4, ai + (in1 ? IA::at(y1l - 1, x1l - 1) : IA::at(0, 0)) + 4 * p4);
        }
      }
      STAMP(5);
      __syncthreads();
      STAMP(6);
    }
  }
  STAMP_FLUSH();
  // ---- hand the register-resident gradients over: D row 4 g + r = co, column li = ci.  Always through LDS, in [co][ci][tap]
  // order: (i) the WK waves that share a tile set (K split) are summed there first -- 256 workgroups x 8 waves of float atomics
  // onto the 4.6 k addresses of a small layer serialise at the memory side; (ii) an accumulator register holds elements that lie
  // 36 bytes apart (ci) in 4 rows (co): as global atomics that is 64 scattered dwords per wave instruction, 17x below the rate of
  // 256 contiguous bytes (MI355X_MICROARCH.md, global float atomics: 0.7 ms of the last layer's 1.0 ms).  The image holds
  // COUT/HALVES output channels at a time so that the last layer's 221 KB of gradients fit.
  float* redw = reinterpret_cast<float*>(smem);  // the operand images are dead
  constexpr int HALVES = (COUT * CIN * 9 * 4 > 120 * 1024) ? 2 : 1, CO_H = COUT / HALVES;
  static_assert(CO_H % 16 == 0 && CO_H * CIN * 9 * 4 <= 150 * 1024, "gradient image");
#pragma unroll
  for (int hf = 0; hf < HALVES; ++hf) {
    // The WK waves that share a tile set add their sums one after the other with plain read-modify-writes (an element has one owner
    // lane per pass; the first pass stores, so nothing is cleared).  The first version cleared the image and used LDS float
    // atomics: 110 k ds_add_f32 lane-operations per workgroup took 178 k cycles -- 80 us of the last layer's 228 us launch, found
    // as the launch time at one frame per workgroup (tools/c5_last_bench.py) and placed by a stamp behind the flush.
    __syncthreads();
#pragma unroll
    for (int kp = 0; kp < WK; ++kp) {
      if (wk == kp) {
#pragma unroll
        for (int a = 0; a < NCO; ++a) {
          const int co0 = 16 * (wco * NCO + a) + 4 * g - hf * CO_H;  // first of this lane's 4 rows, relative to the half
          if (co0 >= 0 && co0 < CO_H) {
#pragma unroll
            for (int b = 0; b < NCI; ++b) {
              const int ci = 16 * (wci * NCI + b) + li;
#pragma unroll
              for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  float* d = redw + ((co0 + r) * CIN + ci) * 9 + t;
                  *d = kp == 0 ? acc[a][b][t][r] : *d + acc[a][b][t][r];
                }
            }
          }
        }
      }
      __syncthreads();
    }
    if (p.part) {
      // 256 workgroups x 55 k float atomics onto the same 55 k addresses took 113 us of the last layer's 228 us launch (launch
      // time against the number of frames, tools/c5_last_bench.py): plain 16-byte stores + one small reduce launch instead
      f32x4* dst = reinterpret_cast<f32x4*>(p.part + (long)blockIdx.x * (COUT * CIN * 9) + (long)hf * CO_H * CIN * 9);
      for (int q = tid; q < CO_H * CIN * 9 / 4; q += NT) dst[q] = reinterpret_cast<const f32x4*>(redw)[q];
    } else {
      for (int q = tid; q < CO_H * CIN * 9; q += NT) atomicAdd(p.g_w + (long)hf * CO_H * CIN * 9 + q, redw[q]);
    }
  }
  STAMP(10);
  {  // bias gradients: every column of accb holds the same sum
    __syncthreads();
    for (int q = tid; q < COUT; q += NT) redw[q] = 0.f;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < NCO; ++a)
      if (wci == 0 && li == 0)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(redw + 16 * (wco * NCO + a) + 4 * g + r, accb[a][r]);
    __syncthreads();
    for (int q = tid; q < COUT; q += NT) atomicAdd(p.g_b + q, redw[q]);
  }
  STAMP(11);
  if (LAST && !FCX) {
#pragma unroll
    for (int k = 0; k < FCN; ++k) {
      const int q = tid + k * NT;
      if (q < p.E * COUT) atomicAdd(p.g_wfc + q, fcw[k]);
    }
    if (tid < p.E) atomicAdd(p.g_bfc + tid, fcb);
  }
}

// g_w[e] += sum over the workgroups' partial sums (ConvBwdParams::part); one thread per four elements
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int nwg, int total, float* __restrict__ g_w) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (4 * q >= total) return;
  const f32x4* src = reinterpret_cast<const f32x4*>(part) + q;
  const long ws = total / 4;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f};
  int w = 0;
  for (; w + 8 <= nwg; w += 8) {  // eight partials in flight: the sum must not become a chain of memory latencies
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(w + u) * ws];
    s0 += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  for (; w < nwg; ++w) s0 += src[w * ws];
  f32x4* dst = reinterpret_cast<f32x4*>(g_w) + q;
  *dst += s0;
}

template <int CIN, int COUT, int W, int BH, bool RC = false>
constexpr int wgrad_lds() {
  const int operands = BH * W * (COUT + TR_PAD) * 2 + Img<CIN, BH, W, true>::BYTES + 256 + (64 + 96 + 96) * 4 +
                       (RC ? round_up(98 * RS0 * 2, 16) + 256 * 4 + 12 * 64 * 16 : 0);
  const int halves = (COUT * CIN * 9 * 4 > 120 * 1024) ? 2 : 1;
  const int flush = COUT / halves * CIN * 9 * 4;  // the [co][ci][tap] image the gradients leave through
  return operands > flush ? operands : flush;
}

// ================================================================================================ dgrad
// STAGE: the band's result goes through LDS and leaves in 16-byte pieces; without it (last layer: dy + 110 KB of weights
// leave no room) every lane stores its 2-byte results itself -- 18 KB per frame, the smallest map of the net
template <int CIN, int COUT, int H, int W, bool LAST, int BH, int MT, bool STAGE>
__global__ __launch_bounds__(NT, 2) void conv_dgrad_kernel(ConvBwdParams p) {
  static_assert(H % BH == 0 && (BH * W) % 16 == 0, "band split");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using ID = Img<COUT, BH, W>;         // haloed dy band
  using WM = Wmat<COUT, CIN>;          // [ci][tap' * COUT + co]
  bf16_t* dyi = reinterpret_cast<bf16_t*>(smem);
  bf16_t* wl = reinterpret_cast<bf16_t*>(smem + ID::BYTES);
  constexpr int o_out = ID::BYTES + round_up(WM::BYTES, 16);
  bf16_t* oa = reinterpret_cast<bf16_t*>(smem + o_out);                         // [BH*W][CIN]
  float* s_dz = reinterpret_cast<float*>(smem + o_out + (STAGE ? BH * W * CIN * 2 : 0));  // [64]
  float* s_dfeat = s_dz + 64;                                                   // [96]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, li = lane & 15;
  // (wv as an SGPR -- readfirstlane, as in the forward kernels -- was measured here: the last layer's weight gradient 0.32 -> 0.47 ms)

  zero_lds(dyi, ID::BYTES, tid);
  for (int q = tid; q < CIN * WM::KP; q += NT) {
    const int ci = q / WM::KP, kk = q % WM::KP, tap = kk / COUT, co = kk % COUT;
    wl[ci * WM::LD + kk] = to_bf16(tap < 9 ? p.w[((long)co * CIN + ci) * 9 + (8 - tap)] : 0.f);
  }
  __syncthreads();
  constexpr int MTILES = BH * W / 16, NTILES = CIN / 16;
  static_assert(MTILES % MT == 0, "unit split");
  constexpr int UNITS = (MTILES / MT) * NTILES;

  ExpandLoad<COUT, H, W, BH + 2> pe;
  MaskLoad<COUT, H * W> pm;
  auto issue = [&](int n, int y0) {
    if (LAST) {
      if (p.dfeat) pm.issue(p.mask + (long)n * H * W * COUT, nullptr, 0, nullptr, tid, p.dfeat + (long)n * COUT);
      else pm.issue(p.mask + (long)n * H * W * COUT, p.dz + (long)n * p.ld_dz, p.E, nullptr, tid);
    } else {
      pe.issue(p.da_out + (long)n * (H / 2) * (W / 2) * COUT, p.idx + (long)n * (H / 2) * (W / 2) * COUT, y0 - 1, tid);
    }
  };
  if ((int)blockIdx.x < p.N) issue(blockIdx.x, 0);
  bool out_pending = false;
  int out_n = 0, out_y0 = 0;
  STAMP_ENTRY;
  STAMP_DECL;
  for (int n = blockIdx.x; n < p.N; n += gridDim.x) {
    for (int y0 = 0; y0 < H; y0 += BH) {
      STAMP(15);
      if (LAST) {
        if (p.dfeat) {  // wave-uniform: d z . W_fc of every frame was made by a GEMM (see conv_wgrad_kernel, FCX)
          if (tid < COUT) s_dfeat[tid] = pm.df * (1.0f / (float)(H * W));
        } else {
          if (tid < p.E) s_dz[tid] = pm.dz;
          __syncthreads();
          for (int c = tid; c < COUT; c += NT) {
            float s = 0.f;
            for (int e = 0; e < p.E; ++e) s += s_dz[e] * p.wfc[e * COUT + c];
            s_dfeat[c] = s * (1.0f / (float)(H * W));
          }
        }
        __syncthreads();
        pm.commit(s_dfeat, dyi, ID::at(0, 0), ID::RS, ID::PS, W, tid);
      } else {
        pe.commit(dyi, ID::at(-1, 0), ID::RS, ID::PS, y0 - 1, tid);
      }
      // the previous band's result leaves HERE, behind the commit of this band's prefetched operands: vector memory operations
      // retire in order and hipcc cannot count stores issued in a loop, so the commit's wait for its (older) loads is vmcnt(0) --
      // with the copy-out in front of it that wait sat through the stores' round trip every band
      if (STAGE && out_pending) {
        uint4* dst = reinterpret_cast<uint4*>(p.da_in + ((long)out_n * H + out_y0) * W * CIN);
        for (int q = tid; q < BH * W * CIN * 2 / 16; q += NT) dst[q] = reinterpret_cast<const uint4*>(oa)[q];
        out_pending = false;
      }
      STAMP(0);
      __syncthreads();
      STAMP(1);
      {
        const bool last_band = y0 + BH >= H;
        const int nn = last_band ? n + (int)gridDim.x : n;
        if (nn < p.N) issue(nn, last_band ? 0 : y0 + BH);
      }
      STAMP(2);
      for (int u = wv; u < UNITS; u += NW) {
        const int mg = u % (MTILES / MT), nt = u / (MTILES / MT);
        int base[MT];
#pragma unroll
        for (int a = 0; a < MT; ++a) {
          const int P = 16 * (mg * MT + a) + li;
          base[a] = ID::at(P / W - 1, P % W - 1) + 8 * g;
        }
        f32x4 acc[MT];
#pragma unroll
        for (int a = 0; a < MT; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < WM::KSTEPS; ++s) {
          const int tap = (32 * s) / COUT, c0 = (32 * s) % COUT;
          const int off = (tap / 3) * ID::RS + (tap % 3) * ID::PS + c0;
          const s16x8 fb = lds_frag(wl + (16 * nt + li) * WM::LD + 32 * s + 8 * g);
#pragma unroll
          for (int a = 0; a < MT; ++a) acc[a] = mfma_bf16(lds_frag(dyi + base[a] + off), fb, acc[a]);
        }
        // (the transposed product -- four channels of one pixel per lane, one 8-byte store: what the fused conv2 kernel below does --
        // was measured here: layer 3 0.260 -> 0.270 ms, layer 4 0.234 -> 0.242: at 32 / 64 channels the 8-byte stores of 16 pixels
        // fall on 2 / 1 bank groups of the staging image)
        bf16_t* od = STAGE ? oa : p.da_in + ((long)n * H + y0) * W * CIN;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) od[(16 * (mg * MT + a) + 4 * g + r) * CIN + 16 * nt + li] = to_bf16(acc[a][r]);
      }
      STAMP(3);
      __syncthreads();
      STAMP(4);
      if (STAGE) { out_pending = true; out_n = n; out_y0 = y0; }  // (the staging area is rewritten behind the next band's first barrier)
      STAMP(5);
    }
  }
  if (STAGE && out_pending) {
    uint4* dst = reinterpret_cast<uint4*>(p.da_in + ((long)out_n * H + out_y0) * W * CIN);
    for (int q = tid; q < BH * W * CIN * 2 / 16; q += NT) dst[q] = reinterpret_cast<const uint4*>(oa)[q];
  }
  STAMP_FLUSH();
}

// ------------------------------------------------------------------------------------------------ layer 4 dgrad, weight-stationary
// d a3 = conv_transpose(dy4, W4) on the 12 x 12 map.  conv_dgrad_kernel keeps the 110 KB of flipped weights in LDS beside ONE dy
// image: load, mask expansion, multiply and store are phases of one workgroup (13.4 k cycles per frame for 3.9 k of MFMA, stage
// timers of round 3).  Here (as conv_last_fwd_ws_kernel, cnn_bf16.hip) the weights live in registers as the A operand of the
// transposed product D[ci][pixel]: wave (c, h) holds input-channel tile c = 16 rows x 27 k steps (108 registers) for the whole walk
// and multiplies it with the pixel tiles of half h of the frame (5 + 4 tiles: waves w and w + 4 share a SIMD, 243 MFMAs per SIMD
// and frame); a lane ends up with four consecutive channels of one pixel = one 8-byte store.  The freed LDS holds TWO dy images:
// every wave expands the next frame's sign mask (loaded at the top of the pass) into the other image behind its multiplications;
// one barrier per frame.  d feat rows travel two passes ahead through a ring of three LDS rows.
struct ConvLastDgradWsParams {
  int N;
  const float* dfeat;   // (N, 96) d z . W_fc (unscaled)
  const uint8_t* mask;  // (N, 144, 96)
  const float* w;       // (96, 64, 3, 3)
  bf16_t* da_in;        // (N, 12, 12, 64)
};
constexpr int L4D_IMG = Img<C4, 12, 12>::BYTES;
constexpr int L4D_O_DF = 2 * L4D_IMG;
constexpr int L4D_LDS_RUN = L4D_O_DF + 3 * C4 * 4;
constexpr int CONV_LAST_DGRAD_WS_LDS = (Wmat<C4, C3>::BYTES > L4D_LDS_RUN) ? Wmat<C4, C3>::BYTES : L4D_LDS_RUN;

__global__ __launch_bounds__(NT, 2) void conv_last_dgrad_ws_kernel(ConvLastDgradWsParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using ID = Img<C4, 12, 12>;
  using WM = Wmat<C4, C3>;  // [ci][tap' * 96 + co], 27 k steps
  constexpr int NPIX = 144, W = 12;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cit = wv & 3, ph = wv >> 2;
  const int t0 = ph ? 5 : 0, ntile = ph ? 4 : 5;

  // ---- flipped, transposed weights through LDS once (conv_dgrad_kernel's image), this wave's 27 A fragments into registers
  {
    bf16_t* wl = reinterpret_cast<bf16_t*>(smem);
    for (int q = tid; q < C3 * WM::KP; q += NT) {
      const int ci = q / WM::KP, kk = q % WM::KP, tap = kk / C4, co = kk % C4;
      wl[ci * WM::LD + kk] = to_bf16(tap < 9 ? p.w[((long)co * C3 + ci) * 9 + (8 - tap)] : 0.f);
    }
  }
  __syncthreads();
  s16x8 wf[WM::KSTEPS];
#pragma unroll
  for (int s = 0; s < WM::KSTEPS; ++s) wf[s] = lds_frag(reinterpret_cast<const bf16_t*>(smem) + (16 * cit + li) * WM::LD + 32 * s + 8 * g);
  __syncthreads();
  zero_lds(smem, L4D_LDS_RUN, tid);
  bf16_t* dy0 = reinterpret_cast<bf16_t*>(smem);
  float* sdf = reinterpret_cast<float*>(smem + L4D_O_DF);  // [3][96]: d feat / 144 of frames i, i + 1, i + 2
  constexpr float inv_hw = 1.0f / (float)NPIX;

  int base[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const int P = 16 * (t0 + t) + li, Pc = P < NPIX ? P : NPIX - 1;
    base[t] = ID::at(Pc / W - 1, Pc % W - 1) + 8 * g;
  }
  MaskLoad<C4, NPIX> pm;
  const int n0 = blockIdx.x, stride = gridDim.x;
  auto frame = [&](int k) { return n0 + (long)k * stride; };  // k-th frame of this workgroup
  float dfreg = 0.f;
  __syncthreads();
  // prologue: d feat rows of frames 0 and 1 into the ring, frame 2's on its way, frame 0's dy image
  if (tid < C4) {
    if (frame(0) < p.N) sdf[tid] = p.dfeat[frame(0) * C4 + tid] * inv_hw;
    if (frame(1) < p.N) sdf[C4 + tid] = p.dfeat[frame(1) * C4 + tid] * inv_hw;
    if (frame(2) < p.N) dfreg = p.dfeat[frame(2) * C4 + tid];
  }
  if (frame(0) < p.N) pm.issue(p.mask + frame(0) * NPIX * C4, nullptr, 0, nullptr, tid);
  __syncthreads();
  if (frame(0) < p.N) pm.commit(sdf, dy0, ID::at(0, 0), ID::RS, ID::PS, W, tid);
  __syncthreads();

  for (int it = 0; frame(it) < p.N; ++it) {
    const long n = frame(it);
    const int cur = it & 1;
    // top of the pass: frame it + 2's d feat row into the ring (read by the expansion of pass it + 1, a barrier later), frame
    // it + 3's on its way; the next frame's sign mask on its way (expanded behind the multiplications below)
    if (tid < C4) {
      if (frame(it + 2) < p.N) sdf[((it + 2) % 3) * C4 + tid] = dfreg * inv_hw;
      if (frame(it + 3) < p.N) dfreg = p.dfeat[frame(it + 3) * C4 + tid];
    }
    const bool more = frame(it + 1) < p.N;
    if (more) pm.issue(p.mask + frame(it + 1) * NPIX * C4, nullptr, 0, nullptr, tid);
    const bf16_t* dyi = dy0 + cur * (L4D_IMG / 2);
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      if (t < ntile) {  // wave-uniform
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        constexpr int RD = 4;  // fragment reads RD k steps ahead of their MFMAs (see conv_last_fwd_ws_kernel)
        s16x8 fb[RD];
        auto koff = [&](int s) {
          const int tap = (32 * s) / C4, c0 = (32 * s) % C4;
          return (tap / 3) * ID::RS + (tap % 3) * ID::PS + c0;
        };
#pragma unroll
        for (int s = 0; s < RD; ++s) fb[s] = lds_frag(dyi + base[t] + koff(s));
#pragma unroll
        for (int s = 0; s < WM::KSTEPS; ++s) {
          SS_SCHED_FENCE();
          acc = mfma_bf16(wf[s], fb[s % RD], acc);
          SS_SCHED_FENCE();
          if (s + RD < WM::KSTEPS) fb[s % RD] = lds_frag(dyi + base[t] + koff(s + RD));
        }
        // D row 4 g + r = input channel, column li = pixel: four consecutive channels of one pixel
        const int P = 16 * (t0 + t) + li;
        *reinterpret_cast<uint2*>(p.da_in + (n * NPIX + P) * C3 + 16 * cit + 4 * g) = pack_bf16x4(acc[0], acc[1], acc[2], acc[3]);
      }
    }
    if (more) pm.commit(sdf + ((it + 1) % 3) * C4, dy0 + (cur ^ 1) * (L4D_IMG / 2), ID::at(0, 0), ID::RS, ID::PS, W, tid);
    __syncthreads();
  }
}

template <int CIN, int COUT, int W, int BH, bool STAGE>
constexpr int dgrad_lds() {
  return Img<COUT, BH, W>::BYTES + round_up(Wmat<COUT, CIN>::BYTES, 16) + (STAGE ? BH * W * CIN * 2 : 0) + (64 + 96) * 4;
}

// ================================================================================================ conv1 wgrad
struct Conv1BwdParams {
  const uint8_t* R;
  int N, standardize;
  const float* st;       // (N, 2) mean, std from the forward
  const bf16_t* da1;     // (N, 48, 48, 16) unmasked
  const uint8_t* i1;     // (N, 48, 48, 16), or null: conv1's pool winners are recomputed from the frame (needs w1, b1)
  const float *w1, *b1;
  float *g_w1, *g_b1;    // (16, 1, 3, 3), (16)
};

// conv1's weight gradient on the POOLED grid: the pool routes the gradient of a pooled pixel q to ONE of the four conv1
// outputs of its window (slot e = argmax byte), so
//   d W1[c][ky][kx] = sum_e sum_q [i1[q][c] == e] da1[q][c] * xn[2 yq + (e >> 1) + ky - 1][2 xq + (e & 1) + kx - 1]
// is four GEMMs with K = pooled pixels (a quarter of the conv1 grid):  D_e[c][v] = sum_q A_e[q][c] * V[q][v],
//   A_e = da1 masked to slot e (four bf16 images [q][16 channels], written once per band while the gradient map arrives),
//   V[q][v = 4 vy + vx] = xn[2 yq - 1 + vy][2 xq - 1 + vx], the 4 x 4 input patch under the window,
// and d W1[c][ky][kx] = sum_e D_e[c][4 ((e >> 1) + ky) + (e & 1) + kx].  Both operands are k-major, so both come from
// ds_read_b64_tr_b16; a row of V is four CONSECUTIVE pixels of one image row = one 8-byte piece, handed in per lane.
// The piece starts at haloed column 2 xq: 8-byte aligned for even xq only, so a second copy of the image, shifted by two
// pixels, serves the odd ones.  288 MFMAs per frame instead of 83 k FMAs per lane-frame on the VALU (measured: 1.79 ms per
// step for 4 % of the backward MACs).
constexpr int C1_BR = 16;                       // pooled rows per band
constexpr int C1_XS = 104;                      // row stride of the two haloed 98 x 98 images (elements)
constexpr int C1_IMG = 98 * C1_XS;              // elements per image
constexpr int C1_AE = C1_BR * 48 * C1;          // elements per masked image of a band
constexpr int CONV1_WGRAD_LDS = 2 * C1_IMG * 2 + 4 * C1_AE * 2 + 256 * 4 + 16 * 16 * 4 + C1_BR * 48 * C1;

__global__ __launch_bounds__(NT, 2) void conv1_wgrad_kernel(Conv1BwdParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int HP = HW0 / 2;
  bf16_t* imgE = reinterpret_cast<bf16_t*>(smem);           // imgE[r][c] = xn haloed (r, c)
  bf16_t* imgO = imgE + C1_IMG;                             // imgO[r][c] = xn haloed (r, c + 2)
  bf16_t* ae = imgO + C1_IMG;                               // [4][C1_BR * 48][16]
  float* s_xn = reinterpret_cast<float*>(ae + 4 * C1_AE);   // [256]
  float* s_red = s_xn + 256;                                // [16 c][16]: 9 weights + 1 bias per channel
  uint8_t* ibl = reinterpret_cast<uint8_t*>(s_red + 16 * 16);  // [C1_BR][48][16] recomputed pool winners of the band
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, li = lane & 15;
  // (wv as an SGPR -- readfirstlane, as in the forward kernels -- was measured here: the last layer's weight gradient 0.32 -> 0.47 ms)
  const int q4 = li >> 2, p4 = li & 3;
  const bool rc = p.i1 == nullptr;
  static_assert(C1_XS == RS0, "conv1_rows reads the image with row stride RS0");
  s16x8 bq[12];
  float bias1 = 0.f;
  if (rc) {
#pragma unroll
    for (int q = 0; q < 12; ++q) bq[q] = conv1_bfrag(p.w1, q, g, li);
    bias1 = p.b1[li];
  }
  zero_lds(smem, 2 * C1_IMG * 2, tid);
  f32x4 acc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  float gb[8];  // bias gradient: this thread always stages the same 8 channels (item parity = channel half)
#pragma unroll
  for (int k = 0; k < 8; ++k) gb[k] = 0.f;
  __syncthreads();

  // Everything a band needs from HBM is requested one band ahead into registers: the frame bytes of the NEXT frame during this
  // frame's last band, the gradient rows (and stored pool winners) of the NEXT band before this band's MFMAs.  (First version:
  // loads at the point of use -- a memory latency per band and per frame with nothing else to do, 44 k cycles per frame for
  // 3.5 k cycles of MFMA.)
  constexpr int NB = C1_BR * HP * 2 / NT;  // staging items (8 channels of a pooled pixel) per thread and band
  static_assert(NB * NT == C1_BR * HP * 2, "band items");
  uint4 px[2];
  auto load_px = [&](int n) {
    const uint4* src = reinterpret_cast<const uint4*>(p.R + (long)n * HW0 * HW0);
    px[0] = src[tid];
    px[1] = (tid + NT < HW0 * HW0 / 16) ? src[tid + NT] : uint4{0u, 0u, 0u, 0u};
  };
  uint4 dpre[NB];
  uint2 ipre[NB];
  auto issue_band = [&](int n, int r0) {
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int q = tid + k * NT, half = q & 1, pp = q >> 1;
      const long so = ((long)n * HP * HP + (long)r0 * HP + pp) * C1 + 8 * half;
      dpre[k] = *reinterpret_cast<const uint4*>(p.da1 + so);
      ipre[k] = rc ? uint2{0u, 0u} : *reinterpret_cast<const uint2*>(p.i1 + so);
    }
  };
  if ((int)blockIdx.x < p.N) {
    load_px(blockIdx.x);
    issue_band(blockIdx.x, 0);
  }

  for (int n = blockIdx.x; n < p.N; n += gridDim.x) {
    if (tid < 256) {
      const float rr = (float)tid / 255.0f;
      s_xn[tid] = p.standardize ? (rr - p.st[2 * (long)n]) / p.st[2 * (long)n + 1] : rr;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int q = tid + k * NT;
      if (q < HW0 * HW0 / 16) {
        const int lin = q * 16, r = lin / HW0 + 1, c0 = lin % HW0 + 1;
        const unsigned wds[4] = {px[k].x, px[k].y, px[k].z, px[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const bf16_t xv = to_bf16(s_xn[(wds[e] >> (8 * b)) & 255u]);  // bf16 as in the forward kernel's image
            const int c = c0 + 4 * e + b;
            imgE[r * C1_XS + c] = xv;
            if (c >= 2) imgO[r * C1_XS + c - 2] = xv;
          }
      }
    }
    for (int r0 = 0; r0 < HP; r0 += C1_BR) {
      const bool last_band = r0 + C1_BR >= HP;
      if (rc) {  // conv1 again for the band's row pairs: only the pool winners are kept
        __syncthreads();  // the image is complete (first band) / the previous band's bytes have been consumed
        conv1_winners(imgE, [&](int q) { return bq[q]; }, bias1, r0, r0 + C1_BR, r0, ibl, wv, g, li);
        __syncthreads();
      }
      // ---- the band's gradients, split by window slot
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int q = tid + k * NT, half = q & 1, pp = q >> 1;
        const uint4 dv = dpre[k];
        const uint2 iv = rc ? *reinterpret_cast<const uint2*>(ibl + pp * C1 + 8 * half) : ipre[k];
        const unsigned d[4] = {dv.x, dv.y, dv.z, dv.w};
        unsigned o[4][4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const unsigned ib = (kk < 2 ? iv.x : iv.y) >> (16 * (kk & 1));
          const unsigned ia = ib & 255u, ic = (ib >> 8) & 255u;
          const unsigned lo = d[kk] & 0xffffu, hi = d[kk] & 0xffff0000u;
          if (ia < 4u) gb[2 * kk] += __uint_as_float(lo << 16);
          if (ic < 4u) gb[2 * kk + 1] += __uint_as_float(hi);
#pragma unroll
          for (unsigned e = 0; e < 4; ++e) o[e][kk] = (ia == e ? lo : 0u) | (ic == e ? hi : 0u);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
          *reinterpret_cast<uint4*>(ae + e * C1_AE + pp * C1 + 8 * half) = uint4{o[e][0], o[e][1], o[e][2], o[e][3]};
      }
      __syncthreads();
      {  // the next band's (frame's) operands fly under the MFMAs below
        const int nn = last_band ? n + (int)gridDim.x : n;
        if (nn < p.N) {
          issue_band(nn, last_band ? 0 : r0 + C1_BR);
          if (last_band) load_px(nn);
        }
      }
      for (int ch = wv; ch < C1_BR * HP / 32; ch += NW) {
        const int P0 = 32 * ch + 8 * g + q4, P1 = P0 + 4;  // this lane's two pooled pixels of the k step
        const int y0 = r0 + P0 / HP, x0 = P0 % HP, y1 = r0 + P1 / HP, x1 = P1 % HP;
        // patch row vy = p4 of the window: haloed image row 2 yq + p4, haloed columns 2 xq .. 2 xq + 3
        const bf16_t* v0 = ((x0 & 1) ? imgO - 2 : imgE) + (2 * y0 + p4) * C1_XS + 2 * x0;
        const bf16_t* v1 = ((x1 & 1) ? imgO - 2 : imgE) + (2 * y1 + p4) * C1_XS + 2 * x1;
        const s16x8 fb = tr_pair(v0, v1);
        const bf16_t* a0 = ae + P0 * C1 + 4 * p4;
        const bf16_t* a1 = ae + P1 * C1 + 4 * p4;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = mfma_bf16(tr_pair(a0 + e * C1_AE, a1 + e * C1_AE), fb, acc[e]);
      }
      __syncthreads();
    }
  }
  // ---- D_e[c = 4 g + r][v = li]: fold the slots into the 3 x 3 taps, the waves through LDS, one atomic per element
  for (int q = tid; q < 16 * 16; q += NT) s_red[q] = 0.f;
  __syncthreads();
  const int vy = li >> 2, vx = li & 3;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ky = vy - (e >> 1), kx = vx - (e & 1);
    if (ky >= 0 && ky <= 2 && kx >= 0 && kx <= 2)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&s_red[(4 * g + r) * 16 + 3 * ky + kx], acc[e][r]);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) atomicAdd(&s_red[(8 * (tid & 1) + k) * 16 + 9], gb[k]);
  __syncthreads();
  if (tid < 16 * 16) {
    const int c = tid >> 4, k = tid & 15;
    if (k < 9) atomicAdd(p.g_w1 + c * 9 + k, s_red[tid]);
    else if (k == 9) atomicAdd(p.g_b1 + c, s_red[tid]);
  }
}

// ================================================================================================ conv2 dgrad + conv1 wgrad, fused
// d a1 = conv_transpose(dy2, W2) is the largest gradient map of the net (73.7 KB per frame as bf16): as two kernels it was written
// to HBM by conv2's data gradient and read back by conv1's weight gradient, 1.13 GB per step for a tensor nobody else wants.
// Here a band of 8 rows of d a1 is born in LDS (conv_dgrad_kernel's routine) and consumed on the spot by conv1's weight gradient
// on the pooled grid (conv1_wgrad_kernel's routine: pool winners recomputed from the frame, four slot-masked images, 4 x 4
// patches by transposing reads).  The band itself never exists as an image: the data gradient's epilogue routes every value
// straight into the slot-masked images.  Optionally (tests) the band is also stored, bit-identical to conv_dgrad_kernel's.
struct Conv2DgradW1Params {
  int N;
  const bf16_t* da2;     // (N, 24, 24, 32) gradient w.r.t. the pooled conv2 output (unmasked)
  const uint8_t* i2;     // (N, 24, 24, 32) conv2's pool winners
  const float* w2;       // (32, 16, 3, 3)
  const uint8_t* R;      // (N, 96, 96)
  const float* st;       // (N, 2) mean, std from the forward
  int standardize;
  const float *w1, *b1;  // conv1 (to recompute its pool winners)
  bf16_t* da1;           // (N, 48, 48, 16) or null
  float *g_w1, *g_b1;
  const uint8_t* i1;     // (N, 48, 48, 16) conv1's pool winners from the forward pass (ss_c5_conv12_fwd_i1), or null: recomputed
};

constexpr int F_BH = 8;                                             // rows of d a1 per band
using F_ID = Img<C2, F_BH, 48>;                                     // haloed dy2 band
using F_WM = Wmat<C2, C1>;                                          // [ci][tap' * 32 + co]
constexpr int F_O_W = F_ID::BYTES;
constexpr int F_O_IMG = F_O_W + round_up(F_WM::BYTES, 16);          // imgE, imgO
constexpr int F_AE = F_BH * 48 * C1;                                // elements per slot-masked image of a band
constexpr int F_O_AE = F_O_IMG + 2 * C1_IMG * 2;
constexpr int F_O_XN = F_O_AE + 4 * F_AE * 2;                       // [256] f32, then s_red [256] f32, then ibl bytes
constexpr int CONV2_DGRAD_W1_LDS = F_O_XN + 256 * 4 + 256 * 4 + F_BH * 48 * C1;
static_assert(CONV2_DGRAD_W1_LDS <= 160 * 1024, "LDS");

__global__ __launch_bounds__(NT, 2) void conv2_dgrad_w1_kernel(Conv2DgradW1Params p) {
  STAMP_ENTRY;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int HP = 48, H = 48, W = 48;
  bf16_t* dyi = reinterpret_cast<bf16_t*>(smem);
  bf16_t* wl = reinterpret_cast<bf16_t*>(smem + F_O_W);
  bf16_t* imgE = reinterpret_cast<bf16_t*>(smem + F_O_IMG);
  bf16_t* imgO = imgE + C1_IMG;
  bf16_t* ae = reinterpret_cast<bf16_t*>(smem + F_O_AE);             // [4 slots][8 * 48 pooled pixels][16 channels]
  float* s_xn = reinterpret_cast<float*>(smem + F_O_XN);
  float* s_red = s_xn + 256;
  uint8_t* ibl = reinterpret_cast<uint8_t*>(s_red + 256);            // [8][48][16] recomputed pool winners of the band
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, li = lane & 15;
  const int q4 = li >> 2, p4 = li & 3;

  s16x8 bq[12];
#pragma unroll
  for (int q = 0; q < 12; ++q) bq[q] = conv1_bfrag(p.w1, q, g, li);
  const float bias_li = p.b1[li];
  zero_lds(dyi, F_ID::BYTES, tid);
  zero_lds(imgE, 2 * C1_IMG * 2, tid);
  for (int q = tid; q < C1 * F_WM::KP; q += NT) {  // flipped, transposed conv2 weights (conv_dgrad_kernel)
    const int ci = q / F_WM::KP, kk = q % F_WM::KP, tap = kk / C2, co = kk % C2;
    wl[ci * F_WM::LD + kk] = to_bf16(tap < 9 ? p.w2[((long)co * C1 + ci) * 9 + (8 - tap)] : 0.f);
  }
  f32x4 acc1[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) acc1[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 gb = {0.f, 0.f, 0.f, 0.f};  // bias gradient of channels 4g .. 4g+3 (the lane's four rows of the transposed product)

  uint4 px[2];
  auto load_px = [&](int n) {
    const uint4* src = reinterpret_cast<const uint4*>(p.R + (long)n * HW0 * HW0);
    px[0] = src[tid];
    px[1] = (tid + NT < HW0 * HW0 / 16) ? src[tid + NT] : uint4{0u, 0u, 0u, 0u};
  };
  ExpandLoad<C2, H, W, F_BH + 2> pe;
  // the band's pool winners (8 rows x 48 pixels x 16 channels = 384 x 16 bytes) ride with the band's other loads when the forward
  // pass left them in HBM
  uint4 iw = {0u, 0u, 0u, 0u};
  auto issue = [&](int n, int y0) {
    pe.issue(p.da2 + (long)n * 24 * 24 * C2, p.i2 + (long)n * 24 * 24 * C2, y0 - 1, tid);
    if (p.i1 && tid < F_BH * 48) iw = reinterpret_cast<const uint4*>(p.i1 + ((long)n * HP + y0) * HP * C1)[tid];
  };
  if ((int)blockIdx.x < p.N) {
    load_px(blockIdx.x);
    issue(blockIdx.x, 0);
  }
  __syncthreads();
  STAMP_DECL;

  // operand addresses of this lane's k steps of conv1's weight gradient, band-relative: made once (see conv_wgrad_kernel)
  constexpr int F_NCH = (F_BH * HP / 32 + NW - 1) / NW;
  int o_v0[F_NCH], o_v1[F_NCH], o_ae0[F_NCH];
#pragma unroll
  for (int k = 0; k < F_NCH; ++k) {
    const int P0 = 32 * (wv + k * NW) + 8 * g + q4, P1 = P0 + 4;
    const int ya = P0 / HP, xa = P0 % HP, yb = P1 / HP, xb = P1 % HP;
    // patch row vy = p4 of the window: haloed image row 2 yq + p4, haloed columns 2 xq .. 2 xq + 3 (odd xq: the shifted copy)
    o_v0[k] = ((xa & 1) ? C1_IMG - 2 : 0) + (2 * ya + p4) * C1_XS + 2 * xa;
    o_v1[k] = ((xb & 1) ? C1_IMG - 2 : 0) + (2 * yb + p4) * C1_XS + 2 * xb;
    o_ae0[k] = P0 * C1 + 4 * p4;
  }
  constexpr int MT = 3, MTILES = F_BH * W / 16;  // 24 m tiles per band: one unit of 3 per wave
  static_assert(MTILES / MT == NW, "one d a1 unit per wave");
  for (int n = blockIdx.x; n < p.N; n += gridDim.x) {
    STAMP(15);
    // ---- the frame's normalised image, twice (conv1_wgrad_kernel).  The last band's MFMAs of the previous frame still read the
    // images: a barrier in front (the frame loop has two barriers per band and none behind the last one)
    __syncthreads();
    if (tid < 256) {
      const float rr = (float)tid / 255.0f;
      s_xn[tid] = p.standardize ? (rr - p.st[2 * (long)n]) / p.st[2 * (long)n + 1] : rr;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int q = tid + k * NT;
      if (q < HW0 * HW0 / 16) {
        const int lin = q * 16, r = lin / HW0 + 1, c0 = lin % HW0 + 1;
        const unsigned wds[4] = {px[k].x, px[k].y, px[k].z, px[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const bf16_t xv = to_bf16(s_xn[(wds[e] >> (8 * b)) & 255u]);
            const int c = c0 + 4 * e + b;
            imgE[r * C1_XS + c] = xv;
            if (c >= 2) imgO[r * C1_XS + c - 2] = xv;
          }
      }
    }
    __syncthreads();  // the images are complete: conv1 reads them below
    STAMP(0);
    for (int y0 = 0; y0 < H; y0 += F_BH) {
      const bool last_band = y0 + F_BH >= H;
      // Two barriers per band.  Before the first: the dy band (from the prefetched registers) and conv1's pool winners of the band's
      // 8 row pairs (one per wave) -- both write what only the phase behind the barrier reads.
#ifdef SS_STAMP
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // diagnostic build only: the wait for the band's loads apart from the expansion
      STAMP(9);
#endif
      pe.commit(dyi, F_ID::at(-1, 0), F_ID::RS, F_ID::PS, y0 - 1, tid);
      STAMP(1);
      if (p.i1) {  // wave-uniform
        if (tid < F_BH * 48) reinterpret_cast<uint4*>(ibl)[tid] = iw;
      } else {
        conv1_winners(imgE, [&](int q) { return bq[q]; }, bias_li, y0, y0 + F_BH, y0, ibl, wv, g, li);
      }
      STAMP(4);
      __syncthreads();
      STAMP(2);
      {
        const int nn = last_band ? n + (int)gridDim.x : n;
        if (nn < p.N) {
          issue(nn, last_band ? 0 : y0 + F_BH);
          if (last_band) load_px(nn);
        }
      }
      // ---- d a1 rows y0 .. y0+7, one unit of 3 m tiles per wave, as a TRANSPOSED product (weights = A operand): the lane ends up
      // with channels 4g .. 4g+3 of pixel 16 (3 wv + a) + li -- and routes them on the spot: the pool winner of each (pixel,
      // channel) picks ONE of the four slot-masked images conv1's weight gradient multiplies, the other three get zeros.  The
      // band of d a1 itself exists in registers only (first version: a bf16 staging image, a separate masking pass over it in
      // two half bands and four more barriers per band -- 20 k of the kernel's 69 k cycles per frame).
      {
        int base[MT];
#pragma unroll
        for (int a = 0; a < MT; ++a) {
          const int P = 16 * (wv * MT + a) + li;
          base[a] = F_ID::at(P / W - 1, P % W - 1) + 8 * g;
        }
        f32x4 acc[MT];
#pragma unroll
        for (int a = 0; a < MT; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < F_WM::KSTEPS; ++s2) {
          const int tap = (32 * s2) / C2, c0 = (32 * s2) % C2;
          const int off = (tap / 3) * F_ID::RS + (tap % 3) * F_ID::PS + c0;
          const s16x8 fb = lds_frag(wl + li * F_WM::LD + 32 * s2 + 8 * g);
#pragma unroll
          for (int a = 0; a < MT; ++a) acc[a] = mfma_bf16(fb, lds_frag(dyi + base[a] + off), acc[a]);
        }
#pragma unroll
        for (int a = 0; a < MT; ++a) {
          const int P = 16 * (wv * MT + a) + li;  // pooled pixel of the band, row-major
          const uint2 ob = pack_bf16x4(acc[a][0], acc[a][1], acc[a][2], acc[a][3]);
          if (p.da1) *reinterpret_cast<uint2*>(p.da1 + (((long)n * H + y0) * W + P) * C1 + 4 * g) = ob;
          const unsigned iv = *reinterpret_cast<const unsigned*>(ibl + P * C1 + 4 * g);  // the four channels' winners
          const unsigned s0 = iv & 255u, s1 = (iv >> 8) & 255u, s2 = (iv >> 16) & 255u, s3 = iv >> 24;
          const unsigned l0 = ob.x & 0xffffu, h0 = ob.x & 0xffff0000u, l1 = ob.y & 0xffffu, h1 = ob.y & 0xffff0000u;
          if (s0 < 4u) gb[0] += __uint_as_float(l0 << 16);
          if (s1 < 4u) gb[1] += __uint_as_float(h0);
          if (s2 < 4u) gb[2] += __uint_as_float(l1 << 16);
          if (s3 < 4u) gb[3] += __uint_as_float(h1);
#pragma unroll
          for (unsigned e = 0; e < 4; ++e)
            *reinterpret_cast<uint2*>(ae + e * F_AE + P * C1 + 4 * g) =
                uint2{(s0 == e ? l0 : 0u) | (s1 == e ? h0 : 0u), (s2 == e ? l1 : 0u) | (s3 == e ? h1 : 0u)};
        }
      }
      STAMP(3);
      __syncthreads();
      STAMP(5);
      // ---- conv1's weight gradient over the band: 12 k steps of 32 pooled pixels (conv1_wgrad_kernel's routine)
#pragma unroll
      for (int k = 0; k < F_NCH; ++k) {
        if (wv + k * NW >= F_BH * HP / 32) break;  // wave-uniform
        const int boff = 2 * y0 * C1_XS;           // the band's rows of the frame images
        const s16x8 fb = tr_pair(imgE + o_v0[k] + boff, imgE + o_v1[k] + boff);
        const bf16_t* a0 = ae + o_ae0[k];
        const bf16_t* a1 = a0 + 4 * C1;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc1[e] = mfma_bf16(tr_pair(a0 + e * F_AE, a1 + e * F_AE), fb, acc1[e]);
      }
      STAMP(8);
      // no barrier here: the next band's commit / winners rewrite the dy band and ibl, which this band read before its second
      // barrier; ae is rewritten behind the next band's FIRST barrier, which every wave reaches after these MFMAs
    }
  }
  STAMP_FLUSH();
  // ---- fold the slots into the 3 x 3 taps, the waves through LDS, one atomic per element (conv1_wgrad_kernel)
  __syncthreads();
  for (int q = tid; q < 16 * 16; q += NT) s_red[q] = 0.f;
  __syncthreads();
  const int vy = li >> 2, vx = li & 3;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ky = vy - (e >> 1), kx = vx - (e & 1);
    if (ky >= 0 && ky <= 2 && kx >= 0 && kx <= 2)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&s_red[(4 * g + r) * 16 + 3 * ky + kx], acc1[e][r]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) atomicAdd(&s_red[(4 * g + r) * 16 + 9], gb[r]);
  __syncthreads();
  if (tid < 16 * 16) {
    const int c = tid >> 4, k = tid & 15;
    if (k < 9) atomicAdd(p.g_w1 + c * 9 + k, s_red[tid]);
    else if (k == 9) atomicAdd(p.g_b1 + c, s_red[tid]);
  }
}

// wgs_per_cu: workgroups the LDS footprint lets a CU hold; the grid is that many times the 256 CUs (a persistent workgroup walks
// its share of the frames), so one workgroup's commit / barrier phases run under another's MFMAs
template <class P, class K>
int launch_persistent(K kernel, const P& p, int lds_bytes, int N, hipStream_t s, int wgs_per_cu = 1) {
  if (lds_bytes > 160 * 1024 / wgs_per_cu) return SS_ERR_UNSUPPORTED;
  if (lds_bytes > 0 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
    return SS_ERR_LAUNCH;
  const int cap = (ss_cnn_max_wgs > 0 ? ss_cnn_max_wgs : ss_device_cus()) * wgs_per_cu;  // (the cap makes a test walk many frames per workgroup)
  const int grid = N < cap ? N : cap;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(NT), lds_bytes, s, p);
  return ss_launch_status();
}

// weight-gradient launch with the partial sums through `part` (may be null: float atomics) + the reduce launch
template <class K>
int launch_wgrad(K kernel, ConvBwdParams& p, int lds_bytes, int N, int total, float* part, long part_floats, hipStream_t s) {
  const int cap = ss_cnn_max_wgs > 0 ? ss_cnn_max_wgs : ss_device_cus();
  const int grid = N < cap ? N : cap;
  p.part = (part && part_floats >= (long)grid * total) ? part : nullptr;
  const int st = launch_persistent(kernel, p, lds_bytes, N, s);
  if (st != SS_OK || !p.part) return st;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div(total / 4, 256)), dim3(256), 0, s, p.part, grid, total, p.g_w);
  return ss_launch_status();
}

}  // namespace

// layer 2: a_in (N,48,48,16), da_out / idx (N,24,24,32);  layer 3: a_in (N,24,24,32), da_out / idx (N,12,12,64)
// part / part_floats (the _ws forms): scratch for the workgroups' partial weight-gradient sums, (workgroups = min(N, CUs)) x
// (COUT * CIN * 9) floats, 16-byte aligned, contents irrelevant; the sums then leave with plain stores and a reduce launch adds them to
// g_w.  Null or too small: float atomics straight onto g_w.
extern "C" int ss_c5_conv_wgrad_ws(int layer, const uint16_t* a_in, const uint16_t* da_out, const uint8_t* idx, int N, float* g_w,
                                   float* g_b, float* part, long part_floats, ss_stream_t stream) {
  SS_REQUIRE(a_in && da_out && idx && g_w && g_b && N > 0, SS_ERR_ARG);
  SS_REQUIRE((reinterpret_cast<uintptr_t>(part) & 15) == 0 && (reinterpret_cast<uintptr_t>(g_w) & 15) == 0, SS_ERR_ARG);
  ConvBwdParams p{};
  p.N = N; p.a_in = a_in; p.da_out = da_out; p.idx = idx; p.g_w = g_w; p.g_b = g_b;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (layer == 2)
    return launch_wgrad(conv_wgrad_kernel<C1, C2, 48, 48, false, 16, 1, 1, 8>, p, wgrad_lds<C1, C2, 48, 16>(), N, C2 * C1 * 9, part, part_floats, s);
  if (layer == 3)
    return launch_wgrad(conv_wgrad_kernel<C2, C3, 24, 24, false, 24, 2, 2, 2>, p, wgrad_lds<C2, C3, 24, 24>(), N, C3 * C2 * 9, part, part_floats, s);
  return SS_ERR_UNSUPPORTED;
}
extern "C" int ss_c5_conv_wgrad(int layer, const uint16_t* a_in, const uint16_t* da_out, const uint8_t* idx, int N, float* g_w,
                                float* g_b, ss_stream_t stream) {
  return ss_c5_conv_wgrad_ws(layer, a_in, da_out, idx, N, g_w, g_b, nullptr, 0, stream);
}

// layer 2's weight gradient with its input recomputed from the frame (no a1 in HBM): R (N,96,96) u8, st (N,2) from the forward
extern "C" int ss_c5_conv2_wgrad_rc_ws(const uint8_t* R, const float* st, int standardize, const float* w1, const float* b1,
                                       const uint16_t* da_out, const uint8_t* idx, int N, float* g_w, float* g_b, float* part,
                                       long part_floats, ss_stream_t stream) {
  SS_REQUIRE(R && st && w1 && b1 && da_out && idx && g_w && g_b && N > 0, SS_ERR_ARG);
  SS_REQUIRE((reinterpret_cast<uintptr_t>(part) & 15) == 0 && (reinterpret_cast<uintptr_t>(g_w) & 15) == 0, SS_ERR_ARG);
  ConvBwdParams p{};
  p.N = N; p.da_out = da_out; p.idx = idx; p.g_w = g_w; p.g_b = g_b; p.R = R; p.st = st; p.standardize = standardize; p.w1 = w1; p.b1 = b1;
  return launch_wgrad(conv_wgrad_kernel<C1, C2, 48, 48, false, 16, 1, 1, 8, true>, p, wgrad_lds<C1, C2, 48, 16, true>(), N, C2 * C1 * 9,
                      part, part_floats, static_cast<hipStream_t>(stream));
}
extern "C" int ss_c5_conv2_wgrad_rc(const uint8_t* R, const float* st, int standardize, const float* w1, const float* b1,
                                    const uint16_t* da_out, const uint8_t* idx, int N, float* g_w, float* g_b, ss_stream_t stream) {
  return ss_c5_conv2_wgrad_rc_ws(R, st, standardize, w1, b1, da_out, idx, N, g_w, g_b, nullptr, 0, stream);
}

// conv2's data gradient and conv1's weight gradient in one kernel: d a1 never reaches HBM (da1 = NULL; a non-NULL da1 also gets the
// map, for tests).  da2 / i2 (N,24,24,32), R (N,96,96), st (N,2) from the forward, w2 (32,16,3,3), w1 / b1 conv1.
// i1 (N,48,48,16): conv1's pool winners as ss_c5_conv12_fwd_i1 left them, or NULL (recomputed from R, w1, b1).
extern "C" int ss_c5_conv2_dgrad_conv1_wgrad_i1(const uint16_t* da2, const uint8_t* i2, int N, const float* w2, const uint8_t* R,
                                                const float* st, int standardize, const float* w1, const float* b1, uint16_t* da1,
                                                float* g_w1, float* g_b1, const uint8_t* i1, ss_stream_t stream) {
  SS_REQUIRE(da2 && i2 && w2 && R && st && w1 && b1 && g_w1 && g_b1 && N > 0, SS_ERR_ARG);
  SS_REQUIRE((reinterpret_cast<uintptr_t>(i1) & 15) == 0, SS_ERR_ARG);
  Conv2DgradW1Params p{N, da2, i2, w2, R, st, standardize, w1, b1, da1, g_w1, g_b1, i1};
  return launch_persistent(conv2_dgrad_w1_kernel, p, CONV2_DGRAD_W1_LDS, N, static_cast<hipStream_t>(stream));
}
extern "C" int ss_c5_conv2_dgrad_conv1_wgrad(const uint16_t* da2, const uint8_t* i2, int N, const float* w2, const uint8_t* R,
                                             const float* st, int standardize, const float* w1, const float* b1, uint16_t* da1,
                                             float* g_w1, float* g_b1, ss_stream_t stream) {
  return ss_c5_conv2_dgrad_conv1_wgrad_i1(da2, i2, N, w2, R, st, standardize, w1, b1, da1, g_w1, g_b1, nullptr, stream);
}

extern "C" int ss_c5_conv_dgrad(int layer, const uint16_t* da_out, const uint8_t* idx, int N, const float* w, uint16_t* da_in,
                                ss_stream_t stream) {
  SS_REQUIRE(da_out && idx && w && da_in && N > 0, SS_ERR_ARG);
  ConvBwdParams p{};
  p.N = N; p.da_out = da_out; p.idx = idx; p.w = w; p.da_in = da_in;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (layer == 2) return launch_persistent(conv_dgrad_kernel<C1, C2, 48, 48, false, 8, 3, true>, p, dgrad_lds<C1, C2, 48, 8, true>(), N, s, 2);
  if (layer == 3) return launch_persistent(conv_dgrad_kernel<C2, C3, 24, 24, false, 4, 2, true>, p, dgrad_lds<C2, C3, 24, 4, true>(), N, s, 2);
  return SS_ERR_UNSUPPORTED;
}

// last layer (conv 64 -> 96 on 12 x 12, global average, Linear(96 -> E)): d z (N,E) -> d W4, d b4, d Wfc, d bfc / d a3
extern "C" int ss_c5_conv_last_wgrad(const uint16_t* a_in, const float* dz, int ld_dz, int E, const float* wfc, const uint8_t* mask,
                                     const float* feat, int N, float* g_w, float* g_b, float* g_wfc, float* g_bfc,
                                     ss_stream_t stream) {
  SS_REQUIRE(a_in && dz && wfc && mask && feat && g_w && g_b && g_wfc && g_bfc && N > 0, SS_ERR_ARG);
  SS_REQUIRE(E > 0 && E <= 64 && ld_dz >= E, SS_ERR_UNSUPPORTED);
  ConvBwdParams p{};
  p.N = N; p.a_in = a_in; p.dz = dz; p.ld_dz = ld_dz; p.E = E; p.wfc = wfc; p.mask = mask; p.feat = feat;
  p.g_w = g_w; p.g_b = g_b; p.g_wfc = g_wfc; p.g_bfc = g_bfc;
  return launch_persistent(conv_wgrad_kernel<C3, C4, 12, 12, true, 12, 2, 4, 1>, p, wgrad_lds<C3, C4, 12, 12>(), N,
                           static_cast<hipStream_t>(stream));
}

static const bool ss_c5_last_ws = !(getenv("SS_C5_LAST_WS") && getenv("SS_C5_LAST_WS")[0] == '0');

// The same two with d feat * 144 = d z . W_fc of every frame ready-made (dfeat (N, 96) f32: one small GEMM in front of them); the fc
// gradients (g_wfc = d z^T . feat, g_bfc = column sums of d z) are then the caller's GEMMs as well.
extern "C" int ss_c5_conv_last_wgrad_df(const uint16_t* a_in, const float* dfeat, const uint8_t* mask, int N, float* g_w, float* g_b,
                                        float* part, long part_floats, ss_stream_t stream) {
  SS_REQUIRE(a_in && dfeat && mask && g_w && g_b && N > 0, SS_ERR_ARG);
  SS_REQUIRE((reinterpret_cast<uintptr_t>(part) & 15) == 0 && (reinterpret_cast<uintptr_t>(g_w) & 15) == 0, SS_ERR_ARG);
  ConvBwdParams p{};
  p.N = N; p.a_in = a_in; p.dfeat = dfeat; p.mask = mask; p.g_w = g_w; p.g_b = g_b;
  return launch_wgrad(conv_wgrad_kernel<C3, C4, 12, 12, true, 12, 2, 4, 1, false, 2, true>, p, wgrad_lds<C3, C4, 12, 12>(), N,
                      C4 * C3 * 9, part, part_floats, static_cast<hipStream_t>(stream));
}
extern "C" int ss_c5_conv_last_dgrad_df(const float* dfeat, const uint8_t* mask, int N, const float* w, uint16_t* da_in,
                                        ss_stream_t stream) {
  SS_REQUIRE(dfeat && mask && w && da_in && N > 0, SS_ERR_ARG);
  if (ss_c5_last_ws) {  // weight-stationary form (SS_C5_LAST_WS=0: the LDS-resident weights of conv_dgrad_kernel)
    ConvLastDgradWsParams q{N, dfeat, mask, w, da_in};
    return launch_persistent(conv_last_dgrad_ws_kernel, q, CONV_LAST_DGRAD_WS_LDS, N, static_cast<hipStream_t>(stream));
  }
  ConvBwdParams p{};
  p.N = N; p.dfeat = dfeat; p.mask = mask; p.w = w; p.da_in = da_in;
  return launch_persistent(conv_dgrad_kernel<C3, C4, 12, 12, true, 12, 3, false>, p, dgrad_lds<C3, C4, 12, 12, false>(), N,
                           static_cast<hipStream_t>(stream));
}

extern "C" int ss_c5_conv_last_dgrad(const float* dz, int ld_dz, int E, const float* wfc, const uint8_t* mask, int N, const float* w,
                                     uint16_t* da_in, ss_stream_t stream) {
  SS_REQUIRE(dz && wfc && mask && w && da_in && N > 0, SS_ERR_ARG);
  SS_REQUIRE(E > 0 && E <= 64 && ld_dz >= E, SS_ERR_UNSUPPORTED);
  ConvBwdParams p{};
  p.N = N; p.dz = dz; p.ld_dz = ld_dz; p.E = E; p.wfc = wfc; p.mask = mask; p.w = w; p.da_in = da_in;
  return launch_persistent(conv_dgrad_kernel<C3, C4, 12, 12, true, 12, 3, false>, p, dgrad_lds<C3, C4, 12, 12, false>(), N,
                           static_cast<hipStream_t>(stream));
}

extern "C" int ss_c5_conv1_wgrad(const uint8_t* R, int N, int standardize, const float* st, const uint16_t* da1, const uint8_t* i1,
                                 const float* w1, const float* b1, float* g_w1, float* g_b1, ss_stream_t stream) {
  SS_REQUIRE(R && st && da1 && g_w1 && g_b1 && N > 0, SS_ERR_ARG);
  SS_REQUIRE(i1 || (w1 && b1), SS_ERR_ARG);  // stored pool winners, or the conv1 parameters to recompute them
  Conv1BwdParams p{R, N, standardize, st, da1, i1, w1, b1, g_w1, g_b1};
  return launch_persistent(conv1_wgrad_kernel, p, CONV1_WGRAD_LDS, N, static_cast<hipStream_t>(stream));
}
