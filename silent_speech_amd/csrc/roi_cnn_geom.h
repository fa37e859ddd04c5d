// LDS geometry shared by the ROI-CNN forward and backward kernels.
#pragma once

namespace {

struct CnnGeom {
  int H, W, H2, W2, H4, W4, P;  // P = H4*W4
  int XS;                       // row stride of the haloed normalised image
  int S1, P1;                   // pooled-1 map: row stride, plane stride (== 18 mod 32)
  int S2, P2;                   // pooled-2 map
  int lds_floats;
};

static inline int plane_stride(int n) {  // smallest >= n that is == 18 (mod 32)
  int r = n % 32;
  int add = (18 - r + 32) % 32;
  return n + add;
}

static inline CnnGeom make_geom(int H, int W) {
  CnnGeom g;
  g.H = H; g.W = W; g.H2 = H / 2; g.W2 = W / 2; g.H4 = H / 4; g.W4 = W / 4; g.P = g.H4 * g.W4;
  g.XS = W + 2;
  g.S1 = g.W2 + 2; g.P1 = plane_stride((g.H2 + 2) * g.S1);
  g.S2 = g.W4 + 2; g.P2 = plane_stride((g.H4 + 2) * g.S2);
  g.lds_floats = (H + 2) * g.XS + 8 * g.P1 + 16 * g.P2 + 512;
  return g;
}

}  // namespace
