// Compile-time LDS geometry shared by the ROI-CNN forward and backward kernels.  The kernels are instantiated per
// frame size so every plane / row stride is an immediate in the ds_read/ds_write offsets and every pixel -> (y, x)
// split is a multiply-shift; SS_CNN_SHAPES lists the sizes built (the reference ships 48x96, BASELINE.json's
// synthetic configs use 64x64).
#pragma once

namespace {

constexpr int plane_stride(int n) {  // smallest >= n that is == 18 (mod 32): conflict-free MFMA operand reads
  return n + (18 - n % 32 + 32) % 32;  // both when lanes walk pixels (stride 1) and when they walk planes
}

// one row of the st_feat stash: 24 averaged conv3 features, 24 counts of positive conv3 outputs, the frame's mean and
// standard deviation (the backward kernel would otherwise redo the pixel statistics and their f64 arithmetic per frame), pad
constexpr int ST_FEAT = 52;

template <int H_, int W_>
struct Geom {
  static constexpr int H = H_, W = W_, H2 = H_ / 2, W2 = W_ / 2, H4 = H_ / 4, W4 = W_ / 4;
  static constexpr int HW = H * W, HW2 = H2 * W2, P = H4 * W4;
  static constexpr int XS = W + 2;                              // haloed normalised image, row stride
  static constexpr int S1 = W2 + 2, P1 = plane_stride((H2 + 2) * S1);  // haloed pooled-1 map
  static constexpr int S2 = W4 + 2, P2 = plane_stride((H4 + 2) * S2);  // haloed pooled-2 map
  // pool-1 argmax bytes, plane stride per channel: 16 bytes of padding put the eight planes four LDS banks apart (the backward's S5
  // reads one byte per plane) and keep them 16-byte aligned, so the stash IS the backward kernel's LDS image and arrives by
  // linear LDS-DMA (round 3: the planes went through registers to be laid out one bank apart)
  static constexpr int I1S = HW2 + 16;
  static_assert(H % 4 == 0 && W % 32 == 0 && H2 % 2 == 0 && P % 32 == 0 && HW2 % 32 == 0, "unsupported ROI size");
};

// Per-frame sizes of the six stash arrays as THIS translation unit lays them out: a1 / a2 floats, pool-1 argmax bytes,
// pool-2 argmax bytes, conv3 sign-mask bytes, floats of an st_feat row.  The caller allocates from ss_roi_cnn_stash_size and
// hands the same six numbers to the forward and the backward entry point; each compares them with its own Geom.
constexpr int SS_CNN_STASH_SIZES = 6;
template <class G>
inline void stash_sizes_of(int* s) {
  s[0] = 8 * G::P1; s[1] = 16 * G::P2; s[2] = 8 * G::I1S; s[3] = G::P * 16; s[4] = G::P * 32; s[5] = ST_FEAT;
}
template <class G>
inline bool stash_sizes_match(const int* s) {
  int mine[SS_CNN_STASH_SIZES];
  stash_sizes_of<G>(mine);
  for (int k = 0; k < SS_CNN_STASH_SIZES; ++k)
    if (s[k] != mine[k]) return false;
  return true;
}

}  // namespace

// X(H, W) for every instantiated frame size
#define SS_CNN_SHAPES(X) X(64, 64) X(48, 96) X(32, 32)
