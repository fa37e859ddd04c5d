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
// PS = C + 8 (C >= 32) keeps 16 consecutive pixels on 16 different 16-byte slots of the 256-byte bank row
// (pixel stride 20 / 36 / 52 dwords, all 4 x odd); the 16-channel map keeps PS = 16 and pads its rows instead.
template <int C_, int H_, int W_>
struct Img {
  static constexpr int C = C_, H = H_, W = W_;
  static constexpr int PS = C >= 32 ? C + 8 : C;
  static constexpr int RS = C >= 32 ? (W + 2) * PS : row_stride_half_bank((W + 2) * PS);
  static constexpr int ELEMS = (H + 2) * RS;
  static constexpr int BYTES = ELEMS * 2;
  __device__ static constexpr int at(int y, int x) { return (y + 1) * RS + (x + 1) * PS; }  // (y, x) may be -1 .. H / W
};

// weights of a conv layer as the B operand: [n][kk], kk = tap * CK + c, rows padded to a multiple of 32 plus 8
template <int CK, int CN>
struct Wmat {
  static constexpr int K = 9 * CK, KP = round_up(K, 32), LD = KP + 8, KSTEPS = KP / 32;
  static constexpr int ELEMS = CN * LD, BYTES = ELEMS * 2;
};

}  // namespace c5
