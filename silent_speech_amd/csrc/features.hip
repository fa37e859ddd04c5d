// Per-frame landmark feature fuse and ROI crop-box arithmetic.
//
// Replaces extract_feature / mouth_width_px (/root/reference/record_landmarks_official.py:52-100,
// /root/reference/live_infer_official.py:141-169) and the index arithmetic of crop_roi /
// crop_roi_gray (record_landmarks_official.py:106-114, live_infer_official.py:172-181).
//
// HBM-bound byte/float shuffling: one wavefront per frame, the frame's K points are staged in LDS
// with coalesced loads, reductions use wave shuffles.  The crop box must be BIT-EXACT, and it
// depends on the centre (a float32 mean) and the float64 mouth width, so those two follow NumPy's
// evaluation order literally: the centre is a sequential float32 sum over the K rows (NumPy's
// axis-0 reduction order) followed by one float32 divide, computed redundantly by every lane from
// LDS broadcasts, and all roundings are pinned with the *_rn intrinsics so the compiler cannot
// contract them into FMAs.
#include "ss_common.h"

namespace {

constexpr int FF_WAVES = 4;   // frames per workgroup
constexpr int FF_MAXK = 256;  // landmarks per frame the LDS staging holds

struct FeatParams {
  const float* lm;       // (B,T,K,2)
  const uint8_t* reset;  // (B,T) or null
  int B, T, K, w, h;
  int a_left, a_right, a_up, a_lo;
  int variant;  // 0 recorder, 1 live
  float* X;
  int ldx;
  float* center;   // (B,T,2)
  double* fourth;  // (B,T)
};

// pixel coordinates, centre and scale of one frame from its staged landmarks
struct FrameNorm {
  float cx, cy;    // centre
  double mw;       // mouth width (already rounded through f32 for the live variant)
  float scale32;   // float32(mw + 1e-6)
};

__device__ __forceinline__ FrameNorm frame_norm(const float* __restrict__ pts, int K, float fw, float fh, int a_left,
                                                int a_right, int variant) {
  FrameNorm r;
  float sx = 0.f, sy = 0.f;
  for (int k = 0; k < K; ++k) {  // sequential f32 accumulation, identical in every lane
    sx = __fadd_rn(sx, __fmul_rn(pts[2 * k], fw));
    sy = __fadd_rn(sy, __fmul_rn(pts[2 * k + 1], fh));
  }
  r.cx = __fdiv_rn(sx, (float)K);
  r.cy = __fdiv_rn(sy, (float)K);
  const float lx = pts[2 * a_left], ly = pts[2 * a_left + 1], rx = pts[2 * a_right], ry = pts[2 * a_right + 1];
  if (variant == 0) {
    const double dx = __dsub_rn(__dmul_rn((double)lx, (double)fw), __dmul_rn((double)rx, (double)fw));
    const double dy = __dsub_rn(__dmul_rn((double)ly, (double)fh), __dmul_rn((double)ry, (double)fh));
    r.mw = sqrt(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
  } else {
    const float dx = __fsub_rn(__fmul_rn(lx, fw), __fmul_rn(rx, fw));
    const float dy = __fsub_rn(__fmul_rn(ly, fh), __fmul_rn(ry, fh));
    r.mw = (double)ss_sqrt_rn_f32(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
  }
  r.scale32 = (float)__dadd_rn(r.mw, 1e-6);
  return r;
}

__global__ __launch_bounds__(FF_WAVES * 64) void feature_fuse_kernel(FeatParams p) {
  __shared__ float cur[FF_WAVES][2 * FF_MAXK];
  __shared__ float prv[FF_WAVES][2 * FF_MAXK];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long frame = (long)blockIdx.x * FF_WAVES + wv;
  const long nframes = (long)p.B * p.T;
  const bool active = frame < nframes;
  const int t = active ? (int)(frame % p.T) : 0;
  const int K = p.K;
  const bool has_prev = active && (t > 0) && !(p.reset && p.reset[frame]);
  if (active) {
    const float* src = p.lm + frame * 2 * K;
    for (int q = lane; q < 2 * K; q += 64) {
      cur[wv][q] = src[q];
      if (has_prev) prv[wv][q] = src[q - 2 * K];
    }
  }
  __syncthreads();
  if (!active) return;
  const float fw = (float)p.w, fh = (float)p.h;
  const FrameNorm c = frame_norm(cur[wv], K, fw, fh, p.a_left, p.a_right, p.variant);
  FrameNorm pr = c;
  if (has_prev) pr = frame_norm(prv[wv], K, fw, fh, p.a_left, p.a_right, p.variant);

  float* xr = p.X + frame * p.ldx;
  float vsum = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float nx = __fdiv_rn(__fsub_rn(__fmul_rn(cur[wv][2 * k], fw), c.cx), c.scale32);
    const float ny = __fdiv_rn(__fsub_rn(__fmul_rn(cur[wv][2 * k + 1], fh), c.cy), c.scale32);
    xr[2 * k] = nx;
    xr[2 * k + 1] = ny;
    if (has_prev) {
      const float qx = __fdiv_rn(__fsub_rn(__fmul_rn(prv[wv][2 * k], fw), pr.cx), pr.scale32);
      const float qy = __fdiv_rn(__fsub_rn(__fmul_rn(prv[wv][2 * k + 1], fh), pr.cy), pr.scale32);
      const float dx = nx - qx, dy = ny - qy;
      vsum += ss_sqrt_rn_f32(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    }
  }
  vsum = wave_sum(vsum);
  if (lane == 0) {
    const float ux = __fmul_rn(cur[wv][2 * p.a_up], fw), uy = __fmul_rn(cur[wv][2 * p.a_up + 1], fh);
    const float lx = __fmul_rn(cur[wv][2 * p.a_lo], fw), ly = __fmul_rn(cur[wv][2 * p.a_lo + 1], fh);
    const float dx = __fsub_rn(ux, lx), dy = __fsub_rn(uy, ly);
    const float open_px = ss_sqrt_rn_f32(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    const double aspect = (double)open_px / __dadd_rn(c.mw, 1e-6);
    xr[2 * K] = has_prev ? vsum / (float)K : 0.f;
    xr[2 * K + 1] = open_px;
    xr[2 * K + 2] = (float)c.mw;
    xr[2 * K + 3] = (float)aspect;
    if (p.center) {
      p.center[2 * frame] = c.cx;
      p.center[2 * frame + 1] = c.cy;
    }
    if (p.fourth) p.fourth[frame] = p.variant == 0 ? __dadd_rn(c.mw, 1e-6) : c.mw;
  }
}

// The live chain's front (live_infer_official.py:264-296), one frame for each of n DISTINCT streams: the distance gate
// MOUTH_W_MIN_PX <= mouth_w <= MOUTH_W_MAX_PX decides whether the frame is kept; a kept frame gets extract_feature with the
// stream's velocity state, a dropped one clears that state (``prev_xy = None``, :295-296) so the first frame after re-entry
// has vel = 0.  The state is the previous KEPT frame's raw landmarks (its normalised form is recomputed with that frame's own
// centre and scale, as feature_fuse_kernel does for t - 1) + one flag per stream, both resident in HBM.
struct FeatStreamParams {
  const float* lm;          // (n,K,2)
  const int32_t* ids;       // (n) stream of each frame
  const uint8_t* recording; // (n) or null: 0 = the stream is idle (nothing kept, state untouched -- ``if recording`` is false)
  int n, n_streams, K, w, h;
  int a_left, a_right, a_up, a_lo;
  int variant;
  double band_lo, band_hi;
  float* prev_lm;           // (S,K,2)
  uint8_t* has_prev;        // (S)
  float* X;
  int ldx;
  float* center;            // (n,2)
  double* fourth;           // (n)
  uint8_t* kept;            // (n): 1 = the frame passed the gate and its row of X / center / fourth is valid
};

__global__ __launch_bounds__(FF_WAVES * 64) void feature_fuse_stream_kernel(FeatStreamParams p) {
  __shared__ float cur[FF_WAVES][2 * FF_MAXK];
  __shared__ float prv[FF_WAVES][2 * FF_MAXK];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * FF_WAVES + wv;
  const bool active = i < p.n;
  const int K = p.K;
  const int sid = active ? p.ids[i] : 0;
  const bool rec = active && (!p.recording || p.recording[i]);
  const bool has_prev = rec && p.has_prev[sid];
  if (active) {
    const float* src = p.lm + (long)i * 2 * K;
    const float* psrc = p.prev_lm + (long)sid * 2 * K;
    for (int q = lane; q < 2 * K; q += 64) {
      cur[wv][q] = src[q];
      if (has_prev) prv[wv][q] = psrc[q];
    }
  }
  __syncthreads();
  if (!active) return;
  const float fw = (float)p.w, fh = (float)p.h;
  const FrameNorm c = frame_norm(cur[wv], K, fw, fh, p.a_left, p.a_right, p.variant);
  const bool in_band = c.mw >= p.band_lo && c.mw <= p.band_hi;  // Python: MIN <= mw <= MAX on a float
  const bool keep = rec && in_band;
  float* xr = p.X + (long)i * p.ldx;
  if (!keep) {
    for (int q = lane; q < 2 * K + 4; q += 64) xr[q] = 0.f;
    if (lane == 0) {
      p.kept[i] = 0;
      // live_infer_official.py:294-296 ``else: if recording: prev_xy = None``; the recorder clears it on EVERY frame that is not
      // appended, recording or not (record_landmarks_official.py:199-201 ``else: prev_xy = None``)
      if (rec || p.variant == 0) p.has_prev[sid] = 0;
      p.center[2 * i] = c.cx;
      p.center[2 * i + 1] = c.cy;
      p.fourth[i] = p.variant == 0 ? __dadd_rn(c.mw, 1e-6) : c.mw;
    }
    return;
  }
  FrameNorm pr = c;
  if (has_prev) pr = frame_norm(prv[wv], K, fw, fh, p.a_left, p.a_right, p.variant);
  float vsum = 0.f;
  float* pdst = p.prev_lm + (long)sid * 2 * K;
  for (int k = lane; k < K; k += 64) {
    const float nx = __fdiv_rn(__fsub_rn(__fmul_rn(cur[wv][2 * k], fw), c.cx), c.scale32);
    const float ny = __fdiv_rn(__fsub_rn(__fmul_rn(cur[wv][2 * k + 1], fh), c.cy), c.scale32);
    xr[2 * k] = nx;
    xr[2 * k + 1] = ny;
    if (has_prev) {
      const float qx = __fdiv_rn(__fsub_rn(__fmul_rn(prv[wv][2 * k], fw), pr.cx), pr.scale32);
      const float qy = __fdiv_rn(__fsub_rn(__fmul_rn(prv[wv][2 * k + 1], fh), pr.cy), pr.scale32);
      const float dx = nx - qx, dy = ny - qy;
      vsum += ss_sqrt_rn_f32(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    }
    pdst[2 * k] = cur[wv][2 * k];  // this frame becomes the stream's previous kept frame
    pdst[2 * k + 1] = cur[wv][2 * k + 1];
  }
  vsum = wave_sum(vsum);
  if (lane == 0) {
    const float ux = __fmul_rn(cur[wv][2 * p.a_up], fw), uy = __fmul_rn(cur[wv][2 * p.a_up + 1], fh);
    const float lx = __fmul_rn(cur[wv][2 * p.a_lo], fw), ly = __fmul_rn(cur[wv][2 * p.a_lo + 1], fh);
    const float dx = __fsub_rn(ux, lx), dy = __fsub_rn(uy, ly);
    const float open_px = ss_sqrt_rn_f32(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    const double aspect = (double)open_px / __dadd_rn(c.mw, 1e-6);
    xr[2 * K] = has_prev ? vsum / (float)K : 0.f;
    xr[2 * K + 1] = open_px;
    xr[2 * K + 2] = (float)c.mw;
    xr[2 * K + 3] = (float)aspect;
    p.center[2 * i] = c.cx;
    p.center[2 * i + 1] = c.cy;
    p.fourth[i] = p.variant == 0 ? __dadd_rn(c.mw, 1e-6) : c.mw;
    p.kept[i] = 1;
    p.has_prev[sid] = 1;
  }
}

__global__ __launch_bounds__(256) void crop_idx_kernel(const float* __restrict__ center, const double* __restrict__ scale,
                                                       int n, int w, int h, int variant, int* __restrict__ box) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= n) return;
  const float cx32 = center[2 * q], cy32 = center[2 * q + 1];
  const double s = scale[q];
  const double hw = __dmul_rn(1.2, s), hh = __dmul_rn(1.0, s);
  int x1, x2, y1, y2, valid;
  if (variant == 0) {
    // float32 centre +- python float: NEP-50 weak promotion -> the python float is rounded to f32 first
    const float hwf = (float)hw, hhf = (float)hh;
    const float lox = __fsub_rn(cx32, hwf), hix = __fadd_rn(cx32, hwf);
    const float loy = __fsub_rn(cy32, hhf), hiy = __fadd_rn(cy32, hhf);
    x1 = lox > 0.f ? (int)lox : 0;
    x2 = hix < (float)w ? (int)hix : w;
    y1 = loy > 0.f ? (int)loy : 0;
    y2 = hiy < (float)h ? (int)hiy : h;
    valid = !(x2 <= x1 || y2 <= y1);
  } else {
    const double cx = (double)cx32, cy = (double)cy32;
    const double lox = __dsub_rn(cx, hw), hix = __dadd_rn(cx, hw);
    const double loy = __dsub_rn(cy, hh), hiy = __dadd_rn(cy, hh);
    x1 = lox > 0.0 ? (int)lox : 0;
    x2 = hix < (double)w ? (int)hix : w;
    y1 = loy > 0.0 ? (int)loy : 0;
    y2 = hiy < (double)h ? (int)hiy : h;
    valid = !(x2 <= x1 + 2 || y2 <= y1 + 2);
  }
  int* b = box + 5 * (long)q;
  b[0] = x1; b[1] = x2; b[2] = y1; b[3] = y2; b[4] = valid;
}

}  // namespace

extern "C" int ss_feature_fuse(const float* lm, const uint8_t* reset, int B, int T, int K, int w, int h, int a_left,
                               int a_right, int a_up, int a_lo, int variant, float* X, int ldx, float* center,
                               double* fourth, ss_stream_t stream) {
  SS_REQUIRE(lm && X && B > 0 && T > 0 && K > 0 && w > 0 && h > 0, SS_ERR_ARG);
  SS_REQUIRE(ldx >= 2 * K + 4 && (variant == 0 || variant == 1), SS_ERR_ARG);
  SS_REQUIRE(a_left >= 0 && a_left < K && a_right >= 0 && a_right < K && a_up >= 0 && a_up < K && a_lo >= 0 && a_lo < K,
             SS_ERR_ARG);
  SS_REQUIRE(K <= FF_MAXK, SS_ERR_UNSUPPORTED);
  FeatParams p;
  p.lm = lm; p.reset = reset; p.B = B; p.T = T; p.K = K; p.w = w; p.h = h;
  p.a_left = a_left; p.a_right = a_right; p.a_up = a_up; p.a_lo = a_lo; p.variant = variant;
  p.X = X; p.ldx = ldx; p.center = center; p.fourth = fourth;
  const long nframes = (long)B * T;
  hipLaunchKernelGGL(feature_fuse_kernel, dim3((unsigned)((nframes + FF_WAVES - 1) / FF_WAVES)), dim3(FF_WAVES * 64), 0,
                     static_cast<hipStream_t>(stream), p);
  return ss_launch_status();
}

extern "C" int ss_feature_fuse_stream(const float* lm, const int32_t* stream_ids, const uint8_t* recording, int n, int n_streams,
                                      int K, int w, int h, int a_left, int a_right, int a_up, int a_lo, int variant, double band_lo,
                                      double band_hi, float* prev_lm, uint8_t* has_prev, float* X, int ldx, float* center,
                                      double* fourth, uint8_t* kept, ss_stream_t stream) {
  SS_REQUIRE(lm && stream_ids && prev_lm && has_prev && X && center && fourth && kept, SS_ERR_ARG);
  SS_REQUIRE(n > 0 && n_streams > 0 && K > 0 && w > 0 && h > 0 && ldx >= 2 * K + 4 && (variant == 0 || variant == 1), SS_ERR_ARG);
  SS_REQUIRE(a_left >= 0 && a_left < K && a_right >= 0 && a_right < K && a_up >= 0 && a_up < K && a_lo >= 0 && a_lo < K,
             SS_ERR_ARG);
  SS_REQUIRE(K <= FF_MAXK, SS_ERR_UNSUPPORTED);
  FeatStreamParams p;
  p.lm = lm; p.ids = stream_ids; p.recording = recording; p.n = n; p.n_streams = n_streams; p.K = K; p.w = w; p.h = h;
  p.a_left = a_left; p.a_right = a_right; p.a_up = a_up; p.a_lo = a_lo; p.variant = variant;
  p.band_lo = band_lo; p.band_hi = band_hi; p.prev_lm = prev_lm; p.has_prev = has_prev;
  p.X = X; p.ldx = ldx; p.center = center; p.fourth = fourth; p.kept = kept;
  hipLaunchKernelGGL(feature_fuse_stream_kernel, dim3((unsigned)ceil_div(n, FF_WAVES)), dim3(FF_WAVES * 64), 0,
                     static_cast<hipStream_t>(stream), p);
  return ss_launch_status();
}

extern "C" int ss_roi_crop_idx(const float* center, const double* scale, int n, int w, int h, int variant,
                               int32_t* box, ss_stream_t stream) {
  SS_REQUIRE(center && scale && box && n > 0 && w > 0 && h > 0 && (variant == 0 || variant == 1), SS_ERR_ARG);
  hipLaunchKernelGGL(crop_idx_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), center,
                     scale, n, w, h, variant, box);
  return ss_launch_status();
}
