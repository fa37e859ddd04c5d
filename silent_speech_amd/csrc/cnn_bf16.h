// Geometry shared by the bf16 ROI-CNN kernels of BASELINE config 5 (cnn_bf16.hip forward, cnn_bf16_bwd.hip backward):
//   96x96 uint8 frame -> normalise -> conv 1->16 + ReLU + pool -> conv 16->32 + ReLU + pool -> conv 32->64 + ReLU + pool
//   -> conv 64->96 + ReLU -> global average -> Linear(96 -> E)
// i.e. TinyROICNN of /root/reference/train_model_official.py:209-229 with a fourth conv block and wider channels (the
// reference does not define this model: SURVEY.md 8d row 5, "build-defined").
//
// One kernel per layer, one persistent workgroup per CU walking frames; a layer's whole input frame sits in LDS as a
// zero-haloed pixel-major (NHWC) bf16 image, its 3x3 convolution is an implicit GEMM on v_mfma_f32_16x16x32_bf16:
//   M = 16 pixels, N = 16 output channels, K = 32 = (tap, input channel) pairs -- the 8 k values of a lane are 8 consecutive
//   channels of ONE pixel at ONE tap = one aligned ds_read_b128 at (lane base + immediate).
// Between the layers the pooled maps travel through HBM as plain NHWC bf16 (N, H, W, C) plus one byte per pooled element
// for the backward pass: the position 0..3 of the maximum inside its 2x2 window, or 4 when the maximum is not positive
// (ReLU gives no gradient there), so the backward kernels route gradients with one compare.
#pragma once
#include "bf16_common.h"

namespace c5 {

constexpr int C1 = 16, C2 = 32, C3 = 64, C4 = 96;   // channels after conv 1..4
constexpr int HW0 = 96;                              // frame size
constexpr int NT = 512, NW = NT / 64;                // threads / waves of every CNN workgroup
constexpr int IDX_DEAD = 4;                          // argmax byte of a window whose maximum is <= 0

constexpr int round_up(int a, int b) { return (a + b - 1) / b * b; }
// smallest row stride (elements, 2 per dword) >= n whose dword count is == 32 (mod 64): two image rows then sit half a
// bank row apart, so the 2 x 8 pixel tiles of the 16-channel map (32-byte pixels) read conflict-free
constexpr int row_stride_half_bank(int n) {
  int d = (n + 1) / 2;
  d += (32 - d % 64 + 64) % 64;
  return 2 * d;
}

// LDS image of a C-channel H x W map with a one-pixel zero halo.  PS = pixel stride, RS = row stride (elements).
// The strides decide the bank conflicts of the MFMA operand reads, and those are set by the hardware's lane groups, not by 16
// consecutive lanes: a ds_read_b128 is served in four groups of 16 lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_
// MICROARCH.md, LDS table), i.e. pixels 0-3 and 12-15 of an M tile at chunk g together with pixels 4-11 at chunk g + 1.  Round 2's
// PS = C + 8 was laid out for 16 consecutive lanes and cost every A-fragment read 2 - 3x its cycles (SQ_LDS_BANK_CONFLICT 0.5 - 0.6
// of the active LDS cycles in the layer 3 / 4 kernels).  ImgLayout holds, per (channels, width), strides found by enumerating the
// bank of every lane of every read pattern that touches the image (2 x 8-pixel tiles of the pooled forward layers, 16 consecutive
// pixels of the data-gradient kernels and the last layer, the transposing reads of the weight-gradient kernels): zero conflicts
// for the ds_read_b128 patterns.  Default (unlisted shapes): the round-2 rule.
template <int C, int W>
struct ImgLayout {
  static constexpr int PS = C >= 32 ? C + 8 : C;
  static constexpr int RS = C >= 32 ? (W + 2) * PS : row_stride_half_bank((W + 2) * PS);
};
template <> struct ImgLayout<32, 24> { static constexpr int PS = 32, RS = 848; };    // conv3 forward / weight gradient: a2
template <> struct ImgLayout<64, 12> { static constexpr int PS = 80, RS = 1216; };   // conv4 forward: a3 (16 consecutive pixels)
template <> struct ImgLayout<32, 48> { static constexpr int PS = 48, RS = 2400; };   // conv2 data gradient: dy2 band
template <> struct ImgLayout<64, 24> { static constexpr int PS = 80, RS = 2176; };   // conv3 data gradient: dy3 band
template <> struct ImgLayout<96, 12> { static constexpr int PS = 112, RS = 1600; };  // conv4 data gradient: dy4

// TR = true: the image is read by the transposing reads of a weight-gradient kernel (ds_read_b64_tr_b16, two groups of 32 lanes):
// those keep the round-2 strides (measured: conv3's weight gradient 0.25 -> 0.30 ms on the ds_read_b128 layout of its input).
#ifndef SS_TR_PAD
#define SS_TR_PAD 8
#endif
constexpr int TR_PAD = SS_TR_PAD;  // elements added to a pixel of an image (C >= 32) that the transposing reads walk
template <int C_, int H_, int W_, bool TR = false>
struct Img {
  static constexpr int C = C_, H = H_, W = W_;
  static constexpr int PS = TR ? (C >= 32 ? C + TR_PAD : C) : ImgLayout<C, W>::PS;
  static constexpr int RS = TR ? (C >= 32 ? (W + 2) * PS : row_stride_half_bank((W + 2) * PS)) : ImgLayout<C, W>::RS;
  static_assert(PS >= C && PS % 8 == 0 && RS >= (W + 2) * PS && RS % 8 == 0, "image strides");
  static constexpr int ELEMS = (H + 2) * RS;
  static constexpr int BYTES = ELEMS * 2;
  __device__ static constexpr int at(int y, int x) { return (y + 1) * RS + (x + 1) * PS; }  // (y, x) may be -1 .. H / W
};

// weights of a conv layer as the B operand: [n][kk], kk = tap * CK + c, rows padded to a multiple of 32 plus 16: a lane group of
// ds_read_b128 holds rows 0-3, 12-15 at chunk g and rows 4-11 at chunk g + 1, and the 16 reads fall on 16 different 16-byte slots
// of the 256-byte bank row exactly when the row stride is == 16 (mod 32) elements (KP + 8, the round-2 stride, is 2-way conflicted)
template <int CK, int CN>
struct Wmat {
  static constexpr int K = 9 * CK, KP = round_up(K, 32), LD = KP + 16, KSTEPS = KP / 32;
  static constexpr int ELEMS = CN * LD, BYTES = ELEMS * 2;
};

constexpr int RS0 = 104;  // row stride of the 98 x 98 haloed one-channel normalised image (elements)

// The 12 constant B fragments of the conv1 patch GEMM (cnn_bf16.hip): output position q = (oy, ox) of a 4 x 8 patch reads patch
// cell (oy + ky, ox + kx); lane (g = patch row, li = channel) holds the 8 cells of its row.
__device__ __forceinline__ s16x8 conv1_bfrag(const float* __restrict__ w1, int q, int g, int li) {
  const int oy = q / 6, ox = q % 6;
  s16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ky = g - oy, kx = j - ox;
    f[j] = (short)to_bf16((ky >= 0 && ky <= 2 && kx >= 0 && kx <= 2) ? w1[li * 9 + ky * 3 + kx] : 0.f);
  }
  return f;
}

// conv1 of the row pairs [yp0, yp1) of a frame whose normalised bf16 image sits at img ([98][RS0]): pooled ReLU output
// (bf16) into dst(yp, xp)[c], dst(yp, xp) = a1 + off0 + (yp - ypb) * RS + xp * PS; optionally the argmax bytes into
// ib[((yp - ypb) * 48 + xp) * 16 + c].  getb(q) = the 12 constant weight fragments (output position q of a patch), bias4 = the
// biases of channels 4g .. 4g+3.
// The product is taken TRANSPOSED (round 3): the constant weight fragment is the A operand, the patch the B operand -- the same
// register contents, the operands swapped -- so D[row = channel][column = patch]: a lane then holds FOUR CHANNELS of one pooled
// pixel, which leave as one 8-byte LDS store (and one 4-byte store of argmax bytes) after two v_cvt_pk_bf16_f32, where the
// [patch][channel] form wrote twelve 2-byte values per lane with one convert each.  The epilogue is what this routine costs (its
// twelve MFMAs take 192 cycles): 123 -> 74 vector instructions per row pair, and three kernels recompute conv1 with it.
template <class GETB>
__device__ __forceinline__ void conv1_rows(const bf16_t* img, GETB getb, f32x4 bias4, int yp0, int yp1, int ypb, bf16_t* a1, int off0,
                                           int RS, int PS, uint8_t* ib, int wv, int g, int li) {
  for (int yp = yp0 + wv; yp < yp1; yp += NW) {
    const unsigned* ap = reinterpret_cast<const unsigned*>(img + (2 * yp + g) * RS0 + 6 * li);
    const s16x8 fa = __builtin_bit_cast(s16x8, uint4{ap[0], ap[1], ap[2], ap[3]});
    f32x4 acc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = mfma_bf16(getb(q), fa, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
    for (int w = 0; w < 3; ++w) {  // pooled pixel xp = 3 li + w of the row pair; r = channel 4g + r
      f32x4 v;
      unsigned ibytes = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float best = acc[2 * w][r];
        int bi = 0;
        if (ib) {
          if (acc[2 * w + 1][r] > best) { best = acc[2 * w + 1][r]; bi = 1; }
          if (acc[6 + 2 * w][r] > best) { best = acc[6 + 2 * w][r]; bi = 2; }
          if (acc[6 + 2 * w + 1][r] > best) { best = acc[6 + 2 * w + 1][r]; bi = 3; }
        } else {
          best = fmaxf(fmaxf(best, acc[2 * w + 1][r]), fmaxf(acc[6 + 2 * w][r], acc[6 + 2 * w + 1][r]));
        }
        v[r] = fmaxf(best + bias4[r], 0.f);
        if (ib) ibytes |= (unsigned)(v[r] > 0.f ? bi : IDX_DEAD) << (8 * r);
      }
      const int xp = 3 * li + w;
      if (a1) *reinterpret_cast<uint2*>(a1 + off0 + (yp - ypb) * RS + xp * PS + 4 * g) = pack_bf16x4(v[0], v[1], v[2], v[3]);
      if (ib) *reinterpret_cast<unsigned*>(ib + ((yp - ypb) * 48 + xp) * C1 + 4 * g) = ibytes;
    }
  }
}


// The pool winners alone (conv1's weight gradient recomputes them from the frame): the [patch][channel] product of round 2 --
// lane (g, li) holds channel li of the pooled pixels 3 (4g + r) + w -- kept for this use: measured, the transposed form with its
// packed 4-byte stores made the fused conv2-dgrad / conv1-wgrad kernel slower (0.72 -> 0.77 ms).  bias = b1[li].
template <class GETB>
__device__ __forceinline__ void conv1_winners(const bf16_t* img, GETB getb, float bias, int yp0, int yp1, int ypb, uint8_t* ib, int wv,
                                              int g, int li) {
  for (int yp = yp0 + wv; yp < yp1; yp += NW) {
    const unsigned* ap = reinterpret_cast<const unsigned*>(img + (2 * yp + g) * RS0 + 6 * li);
    const s16x8 fa = __builtin_bit_cast(s16x8, uint4{ap[0], ap[1], ap[2], ap[3]});
    f32x4 acc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = mfma_bf16(fa, getb(q), f32x4{0.f, 0.f, 0.f, 0.f});
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
        const int xp = 3 * (4 * g + r) + w;
        ib[((yp - ypb) * 48 + xp) * C1 + li] = (uint8_t)(v > 0.f ? bi : IDX_DEAD);
      }
  }
}


// ---- HBM -> registers -> LDS in two steps: ``issue`` starts the global loads of the NEXT frame / band before the MFMA
// phase of the current one, ``commit`` writes them into the (single-buffered) LDS images behind it.  The per-layer kernels
// are HBM-bound (a layer's whole input and output cross HBM once per frame); with load and compute back to back a
// workgroup idles through every load phase (first version: conv2 forward 0.42 ms for 0.2 ms of HBM time).

// rows [y0, y0 + ROWS) of an NHWC frame (HF rows) -> rows lrow0, lrow0 + 1, ... of a haloed image; rows outside the frame: zeros
template <class IM, int HF, int ROWS>
struct BandLoad {
  static constexpr int CH = IM::C / 8, ITEMS = ROWS * IM::W * CH, NV = (ITEMS + NT - 1) / NT;
  uint4 v[NV];
  __device__ __forceinline__ void issue(const bf16_t* __restrict__ src, int y0, int tid) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int q = tid + k * NT;
      const int c8 = q % CH, x = (q / CH) % IM::W, y = y0 + q / (CH * IM::W);
      uint4 t = {0u, 0u, 0u, 0u};
      if (q < ITEMS && y >= 0 && y < HF) t = *reinterpret_cast<const uint4*>(src + ((long)y * IM::W + x) * IM::C + 8 * c8);
      v[k] = t;
    }
  }
  __device__ __forceinline__ void commit(bf16_t* img, int lrow0, int tid) const {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int q = tid + k * NT;
      if (q < ITEMS) {
        const int c8 = q % CH, x = (q / CH) % IM::W, yl = q / (CH * IM::W);
        *reinterpret_cast<uint4*>(img + IM::at(yl + lrow0, x) + 8 * c8) = v[k];
      }
    }
  }
};

// dense gradient rows [y0, y0 + ROWS) of a pooled layer (H x W before the pool) from the pooled-grid gradient + argmax bytes:
//   dst(yl, x) = img + off0 + yl * RS + x * PS;  rows outside the frame: zeros
template <int COUT, int H, int W, int ROWS>
struct ExpandLoad {
  static constexpr int WO = W / 2, CH = COUT / 8, ITEMS = ROWS * WO * CH, NV = (ITEMS + NT - 1) / NT;
  uint4 d[NV];
  uint2 ix[NV];
  __device__ __forceinline__ void issue(const bf16_t* __restrict__ da, const uint8_t* __restrict__ idx, int y0, int tid) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int q = tid + k * NT;
      const int c8 = q % CH, xp = (q / CH) % WO, y = y0 + q / (CH * WO);
      uint4 t = {0u, 0u, 0u, 0u};
      uint2 u = {0x04040404u, 0x04040404u};
      if (q < ITEMS && y >= 0 && y < H) {
        const long src = ((long)(y >> 1) * WO + xp) * COUT + 8 * c8;
        t = *reinterpret_cast<const uint4*>(da + src);
        u = *reinterpret_cast<const uint2*>(idx + src);
      }
      d[k] = t;
      ix[k] = u;
    }
  }
  __device__ __forceinline__ void commit(bf16_t* img, int off0, int RS, int PS, int y0, int tid) const {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int q = tid + k * NT;
      if (q < ITEMS) {
        const int c8 = q % CH, xp = (q / CH) % WO, yl = q / (CH * WO);
        const unsigned dd[4] = {d[k].x, d[k].y, d[k].z, d[k].w};
        const unsigned e0 = (unsigned)((y0 + yl) & 1) * 2u;
        unsigned o0[4], o1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // channels 2j, 2j+1
          const unsigned ib = (j < 2 ? ix[k].x : ix[k].y) >> (16 * (j & 1));
          const unsigned ia = ib & 255u, ic = (ib >> 8) & 255u;
          const unsigned lo = dd[j] & 0xffffu, hi = dd[j] & 0xffff0000u;
          o0[j] = (ia == e0 ? lo : 0u) | (ic == e0 ? hi : 0u);
          o1[j] = (ia == e0 + 1u ? lo : 0u) | (ic == e0 + 1u ? hi : 0u);
        }
        bf16_t* dst = img + off0 + yl * RS + (2 * xp) * PS + 8 * c8;
        *reinterpret_cast<uint4*>(dst) = uint4{o0[0], o0[1], o0[2], o0[3]};
        *reinterpret_cast<uint4*>(dst + PS) = uint4{o1[0], o1[1], o1[2], o1[3]};
      }
    }
  }
};

// last layer: sign mask of all NPIX pixels (8 bytes per item) + this thread's piece of the d z row and of the averaged features
template <int COUT, int NPIX>
struct MaskLoad {
  static constexpr int CH = COUT / 8, ITEMS = NPIX * CH, NV = (ITEMS + NT - 1) / NT;
  uint2 m[NV];
  float dz, ft, df;
  // dfrow: this frame's row of d feat * (H*W) = d z . W_fc made by a GEMM over all frames (or null: the kernel makes it from dzrow)
  __device__ __forceinline__ void issue(const uint8_t* __restrict__ mask, const float* __restrict__ dzrow, int E,
                                        const float* __restrict__ featrow, int tid, const float* __restrict__ dfrow = nullptr) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int q = tid + k * NT;
      m[k] = q < ITEMS ? *reinterpret_cast<const uint2*>(mask + (long)(q / CH) * COUT + 8 * (q % CH)) : uint2{0u, 0u};
    }
    dz = (dzrow && tid < E) ? dzrow[tid] : 0.f;
    ft = (featrow && tid < COUT) ? featrow[tid] : 0.f;
    df = (dfrow && tid < COUT) ? dfrow[tid] : 0.f;
  }
  // dy[P][co] = mask ? dfeat[co] : 0
  __device__ __forceinline__ void commit(const float* s_dfeat, bf16_t* img, int off0, int RS, int PS, int W, int tid) const {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int q = tid + k * NT;
      if (q < ITEMS) {
        const int c8 = q % CH, P = q / CH;
        unsigned o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned mb = (j < 2 ? m[k].x : m[k].y) >> (16 * (j & 1));
          const float a = (mb & 255u) ? s_dfeat[8 * c8 + 2 * j] : 0.f, b = ((mb >> 8) & 255u) ? s_dfeat[8 * c8 + 2 * j + 1] : 0.f;
          o[j] = pack_bf16(a, b);
        }
        *reinterpret_cast<uint4*>(img + off0 + (P / W) * RS + (P % W) * PS + 8 * c8) = uint4{o[0], o[1], o[2], o[3]};
      }
    }
  }
};

}  // namespace c5
