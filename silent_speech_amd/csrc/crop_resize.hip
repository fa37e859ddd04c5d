// Mouth ROI: crop box -> BGR2GRAY -> resize to the CNN's input, one workgroup per frame.
//
// Replaces the two OpenCV calls of crop_roi (/root/reference/record_landmarks_official.py:116-118: cvtColor(BGR2GRAY) +
// resize(.., (ROI_W, ROI_H)), default INTER_LINEAR) and crop_roi_gray (/root/reference/live_infer_official.py:184-186:
// INTER_AREA) behind the integer crop box of ss_roi_crop_idx.  OpenCV (opencv-python 4.13.0.90) is not part of the
// reference tree and not installed here: the arithmetic below restates its published 8-bit algorithms
//   gray    : (B*3735 + G*19235 + R*9798 + 2^14) >> 15                                      (imgproc color_rgb, 15-bit)
//   linear  : 11-bit coefficients, horizontal pass in int32, vertical ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2
//   area    : DecimateAlpha tables (float weights over the source cells a destination pixel covers), per source row
//             buf = sum_x S*alpha, sum (+)= beta*buf, round-half-even; integer scales take the box average
//             (2x2: (a+b+c+d+2)>>2); a scale < 1 falls back to the linear pass with area-style coefficients
// and is checked bit for bit against the same restatement in NumPy (oracle/resize_ref.py).  PARITY WITH OPENCV ITSELF IS
// UNPINNED (no cv2 here; wheels may also route resize through IPP): treat it as +-1 grey level until checked.
#include "ss_common.h"

namespace {

__device__ __forceinline__ int gray_at(const uint8_t* __restrict__ f, int w, int y, int x) {
  const uint8_t* p = f + ((long)y * w + x) * 3;
  return (p[0] * 3735 + p[1] * 19235 + p[2] * 9798 + (1 << 14)) >> 15;
}

__device__ __forceinline__ int sat_short(float v) {  // saturate_cast<short>(float): round half to even, clamp
  const int r = (int)rintf(v);
  return r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
}
__device__ __forceinline__ int sat_u8(float v) {
  const int r = (int)rintf(v);
  return r < 0 ? 0 : (r > 255 ? 255 : r);
}

// source index and fraction of the linear pass for destination index d (OpenCV resize(): INTER_LINEAR, or INTER_AREA
// when the axis is enlarged)
__device__ __forceinline__ void linear_coord(int d, int ssize, int dsize, bool area_mode, int& s0, int& a0, int& a1) {
  const double scale = (double)ssize / dsize, inv = (double)dsize / ssize;
  int s;
  float f;
  if (!area_mode) {
    f = (float)((d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f -= s;
  } else {
    s = (int)floor(d * scale);
    f = (float)((d + 1) - (s + 1) * inv);
    f = f <= 0.f ? 0.f : f - floorf(f);
  }
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  s0 = s;
  a0 = sat_short(__fmul_rn(__fsub_rn(1.f, f), 2048.f));
  a1 = sat_short(__fmul_rn(f, 2048.f));
}

struct CropParams {
  const uint8_t* frames;  // (N, h, w, 3) BGR
  const int32_t* box;     // (N, 5) x1, x2, y1, y2, valid
  int N, h, w, RH, RW, interp;  // interp 0 = INTER_LINEAR (recorder), 1 = INTER_AREA (live)
  uint8_t* out;           // (N, RH, RW)
};

__global__ __launch_bounds__(256) void crop_gray_resize_kernel(CropParams p) {
  const int n = blockIdx.x;
  const int32_t* b = p.box + (long)n * 5;
  uint8_t* out = p.out + (long)n * p.RH * p.RW;
  const int x1 = b[0], x2 = b[1], y1 = b[2], y2 = b[3];
  if (!b[4]) {
    for (int q = threadIdx.x; q < p.RH * p.RW; q += 256) out[q] = 0;
    return;
  }
  const uint8_t* f = p.frames + (long)n * p.h * p.w * 3;
  const int sw = x2 - x1, sh = y2 - y1;
  const double scx = (double)sw / p.RW, scy = (double)sh / p.RH;
  const bool area = p.interp == 1 && scx >= 1.0 && scy >= 1.0;
  for (int q = threadIdx.x; q < p.RH * p.RW; q += 256) {
    const int dy = q / p.RW, dx = q - dy * p.RW;
    int res;
    if (!area) {
      const bool am = p.interp == 1;  // INTER_AREA on an enlarged crop: linear pass, area-style coordinates
      int sx, ax0, ax1, sy, by0, by1;
      linear_coord(dx, sw, p.RW, am, sx, ax0, ax1);
      linear_coord(dy, sh, p.RH, am, sy, by0, by1);
      const int sx1 = min(sx + 1, sw - 1), sy1 = min(sy + 1, sh - 1);
      const int S0 = gray_at(f, p.w, y1 + sy, x1 + sx) * ax0 + gray_at(f, p.w, y1 + sy, x1 + sx1) * ax1;
      const int S1 = gray_at(f, p.w, y1 + sy1, x1 + sx) * ax0 + gray_at(f, p.w, y1 + sy1, x1 + sx1) * ax1;
      res = (((by0 * (S0 >> 4)) >> 16) + ((by1 * (S1 >> 4)) >> 16) + 2) >> 2;
      res = res < 0 ? 0 : (res > 255 ? 255 : res);
    } else {
      const int isx = (int)scx, isy = (int)scy;
      if ((double)isx == scx && (double)isy == scy) {  // integer scales: box average
        int sum = 0;
        for (int yy = 0; yy < isy; ++yy)
          for (int xx = 0; xx < isx; ++xx) sum += gray_at(f, p.w, y1 + dy * isy + yy, x1 + dx * isx + xx);
        res = (isx == 2 && isy == 2) ? ((sum + 2) >> 2) : sat_u8(__fmul_rn((float)sum, 1.f / (float)(isx * isy)));
      } else {
        // DecimateAlpha cells of this destination pixel along y and x (doubles for the geometry, float weights)
        const double fy1 = dy * scy, fy2 = fy1 + scy, chh = fmin(scy, sh - fy1);
        int sya = (int)ceil(fy1), syb = min((int)floor(fy2), sh - 1);
        sya = min(sya, syb);
        const double fx1 = dx * scx, fx2 = fx1 + scx, cww = fmin(scx, sw - fx1);
        int sxa = (int)ceil(fx1), sxb = min((int)floor(fx2), sw - 1);
        sxa = min(sxa, sxb);
        const float ax_l = (float)((sxa - fx1) / cww), ax_m = (float)(1.0 / cww);
        const float ax_r = (float)(fmin(fmin(fx2 - sxb, 1.0), cww) / cww);
        const bool has_xl = sxa - fx1 > 1e-3, has_xr = fx2 - sxb > 1e-3;
        auto row = [&](int sy) {  // buf = sum over the source cells of this row, in table order
          float buf = 0.f;  // separate multiply and add (two roundings, as the scalar C++ loop; no FMA contraction)
          if (has_xl) buf = __fadd_rn(buf, __fmul_rn((float)gray_at(f, p.w, y1 + sy, x1 + sxa - 1), ax_l));
          for (int sx = sxa; sx < sxb; ++sx) buf = __fadd_rn(buf, __fmul_rn((float)gray_at(f, p.w, y1 + sy, x1 + sx), ax_m));
          if (has_xr) buf = __fadd_rn(buf, __fmul_rn((float)gray_at(f, p.w, y1 + sy, x1 + sxb), ax_r));
          return buf;
        };
        float sum = 0.f;
        bool first = true;
        auto add = [&](int sy, float beta) {
          const float v = __fmul_rn(beta, row(sy));
          sum = first ? v : __fadd_rn(sum, v);
          first = false;
        };
        if (sya - fy1 > 1e-3) add(sya - 1, (float)((sya - fy1) / chh));
        for (int sy = sya; sy < syb; ++sy) add(sy, (float)(1.0 / chh));
        if (fy2 - syb > 1e-3) add(syb, (float)(fmin(fmin(fy2 - syb, 1.0), chh) / chh));
        res = sat_u8(sum);
      }
    }
    out[q] = (uint8_t)res;
  }
}

}  // namespace

extern "C" int ss_crop_gray_resize(const uint8_t* frames_bgr, int N, int h, int w, const int32_t* box, int roi_h, int roi_w,
                                   int interp, uint8_t* out, ss_stream_t stream) {
  SS_REQUIRE(frames_bgr && box && out && N > 0 && h > 0 && w > 0 && roi_h > 0 && roi_w > 0, SS_ERR_ARG);
  SS_REQUIRE(interp == 0 || interp == 1, SS_ERR_UNSUPPORTED);
  CropParams p;
  p.frames = frames_bgr; p.box = box; p.N = N; p.h = h; p.w = w; p.RH = roi_h; p.RW = roi_w; p.interp = interp; p.out = out;
  hipLaunchKernelGGL(crop_gray_resize_kernel, dim3(N), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return ss_launch_status();
}
