// ROI normalise + TinyROICNN for ANY frame size (H, W multiples of 4): the layer-by-layer form.
//
// The fused kernels (roi_cnn.hip / roi_cnn_bwd.hip) keep a whole frame and its feature maps in one CU's LDS and are
// instantiated for three sizes (roi_cnn_geom.h: 64x64, 48x96, 32x32); a frame of 96x96 would need 350 KB there.  The reference's
// model (/root/reference/train_model_official.py:209-229, normalisation :286-291) takes whatever ROI_H x ROI_W its constants say,
// so every other size runs here: each 3x3 convolution as im2col + the f32 MFMA GEMM of gemm.hip, with small HBM-bound kernels for
// the normalisation, ReLU + 2x2 max-pool (+ argmax), ReLU + global average (+ sign mask) and their gradients.  The sequencing and
// the buffers live in silent_speech_amd/cnn_generic.py (GenericCnn.forward / .backward); nothing here allocates or synchronises.  It is an
// order of magnitude slower than the fused kernels (every map crosses HBM several times) -- a correctness path for shapes outside
// the tuned set, not a tuned one.
//
// Layouts: maps between layers are planar (N, C, h, w) f32; GEMM outputs are pixel-major (N h w, C) f32; im2col rows are
// (N h w, ld >= 9 C) with k = c*9 + ky*3 + kx -- the order of nn.Conv2d's weight (Cout, C, 3, 3) flattened, so the weight tensor is
// the GEMM's [n][k] operand as it stands.
#include "ss_common.h"

namespace {

constexpr int GT = 256;

// exact integer statistics -> mean, unbiased std (clamp 1e-6), xn = (u/255 - mu)/sd with IEEE divides: roi_cnn.hip, stage 0
__global__ __launch_bounds__(GT) void roi_norm_kernel(const uint8_t* __restrict__ R, int HW, int standardize, float* __restrict__ xn,
                                                      float* __restrict__ stats) {
  __shared__ unsigned long long red[2][GT / 64];
  __shared__ float s_mu, s_sd;
  const long n = blockIdx.x;
  const uint8_t* src = R + n * HW;
  unsigned long long su = 0, sq = 0;
  for (int q = threadIdx.x; q < HW; q += GT) {
    const unsigned u = src[q];
    su += u;
    sq += u * u;
  }
  for (int off = 32; off; off >>= 1) {
    su += __shfl_xor(su, off, 64);
    sq += __shfl_xor(sq, off, 64);
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = su; red[1][threadIdx.x >> 6] = sq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long tsu = 0, tsq = 0;
    for (int k = 0; k < GT / 64; ++k) { tsu += red[0][k]; tsq += red[1][k]; }
    float mu = 0.f, sd = 1.f;
    if (standardize) {
      const double nn = (double)HW;
      const double mean_u = (double)tsu / nn;
      mu = (float)mean_u / 255.0f;
      const double var = ((double)tsq - (double)tsu * mean_u) / (nn - 1.0);
      sd = sqrtf((float)(var > 0.0 ? var : 0.0)) / 255.0f;
      sd = fmaxf(sd, 1e-6f);
    }
    s_mu = mu; s_sd = sd;
    if (stats) { stats[2 * n] = mu; stats[2 * n + 1] = sd; }
  }
  __syncthreads();
  const float mu = s_mu, sd = s_sd;
  for (int q = threadIdx.x; q < HW; q += GT) {
    const float r = (float)src[q] / 255.0f;
    xn[n * HW + q] = standardize ? (r - mu) / sd : r;
  }
}

__global__ __launch_bounds__(GT) void im2col3x3_kernel(const float* __restrict__ src, long total, int C, int H, int W,
                                                       float* __restrict__ col, int ld) {
  // one thread per (pixel row of col, channel): nine taps
  for (long q = (long)blockIdx.x * GT + threadIdx.x; q < total; q += (long)gridDim.x * GT) {
    const int c = (int)(q % C);
    const long pix = q / C;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const long n = pix / ((long)W * H);
    const float* plane = src + (n * C + c) * (long)H * W;
    float* dst = col + pix * ld + c * 9;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = y + ky - 1, xx = x + kx - 1;
        dst[ky * 3 + kx] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? plane[yy * W + xx] : 0.f;
      }
    if (c == 0)
      for (int k = 9 * C; k < ld; ++k) col[pix * ld + k] = 0.f;  // alignment padding of the row
  }
}

// y (N H W, C) pixel-major, pre-activation -> a (N, C, H/2, W/2) = max-pool(relu(y)), idx = winner inside the window (0..3, first in
// row-major order on ties: torch's max_pool2d)
__global__ __launch_bounds__(GT) void relu_pool2_kernel(const float* __restrict__ y, long total, int C, int H, int W,
                                                        float* __restrict__ a, uint8_t* __restrict__ idx) {
  const int H2 = H / 2, W2 = W / 2;
  for (long q = (long)blockIdx.x * GT + threadIdx.x; q < total; q += (long)gridDim.x * GT) {
    const int c = (int)(q % C);  // channel fastest: the four reads of a warp are contiguous runs of C floats
    const long w = q / C;
    const int px = (int)(w % W2), py = (int)((w / W2) % H2);
    const long n = w / ((long)W2 * H2);
    const float* base = y + ((n * H + 2 * py) * (long)W + 2 * px) * C + c;
    const float v00 = base[0], v01 = base[C], v10 = base[(long)W * C], v11 = base[(long)W * C + C];
    float best = v00;
    int bi = 0;
    if (v01 > best) { best = v01; bi = 1; }
    if (v10 > best) { best = v10; bi = 2; }
    if (v11 > best) { best = v11; bi = 3; }
    const long o = ((n * C + c) * H2 + py) * (long)W2 + px;
    a[o] = fmaxf(best, 0.f);
    idx[o] = (uint8_t)bi;
  }
}

// y (N P, C) pre-activation -> feat (N, C) = mean over the P pixels of relu(y), mask (N P, C) = y > 0
__global__ __launch_bounds__(GT) void relu_mean_kernel(const float* __restrict__ y, int P, int C, float* __restrict__ feat,
                                                       uint8_t* __restrict__ mask) {
  __shared__ float red[GT];
  const long n = blockIdx.x;
  for (int c = 0; c < C; ++c) {
    float s = 0.f;
    for (int p = threadIdx.x; p < P; p += GT) {
      const float v = y[(n * P + p) * C + c];
      s += fmaxf(v, 0.f);
      if (mask) mask[(n * P + p) * C + c] = v > 0.f;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = GT / 2; off; off >>= 1) {
      if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) feat[n * C + c] = red[0] / (float)P;
    __syncthreads();
  }
}

__global__ __launch_bounds__(GT) void mask_scale_kernel(const uint8_t* __restrict__ mask, const float* __restrict__ dfeat, long total,
                                                        int P, int C, float* __restrict__ dy) {
  for (long q = (long)blockIdx.x * GT + threadIdx.x; q < total; q += (long)gridDim.x * GT) {
    const int c = (int)(q % C);
    const long n = q / ((long)P * C);
    dy[q] = mask[q] ? dfeat[n * C + c] / (float)P : 0.f;
  }
}

// d (N, C, H, W) planar = the transposed convolution's gather: d[n][c][y][x] = sum over taps of dcol[(n, y+1-ky, x+1-kx)][c*9 + tap]
__global__ __launch_bounds__(GT) void col2im3x3_kernel(const float* __restrict__ dcol, int ld, long total, int C, int H, int W,
                                                       float* __restrict__ d) {
  // channel fastest: the C x 9 floats of a dcol row are read by C neighbouring lanes over the nine taps, i.e. whole cache lines
  // (with x fastest a lane's neighbours sat a row stride apart and every 4-byte read pulled its own line: 26 ms per step at
  // 96 x 96, B = 256); the planar stores are the smaller side
  for (long q = (long)blockIdx.x * GT + threadIdx.x; q < total; q += (long)gridDim.x * GT) {
    const int c = (int)(q % C);
    const long pix = q / C;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const long n = pix / ((long)W * H);
    float s = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = y + 1 - ky, xx = x + 1 - kx;  // the output pixel whose tap (ky, kx) read this input pixel
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) s += dcol[((n * H + yy) * (long)W + xx) * ld + c * 9 + ky * 3 + kx];
      }
    d[((n * C + c) * H + y) * (long)W + x] = s;
  }
}

// gradient through max-pool + ReLU: dy (N 2H2 2W2, C) pixel-major; the window's winner gets da if the pooled value is positive
__global__ __launch_bounds__(GT) void pool2_bwd_kernel(const float* __restrict__ da, const float* __restrict__ a,
                                                       const uint8_t* __restrict__ idx, long total, int C, int H2, int W2,
                                                       float* __restrict__ dy) {
  const int W = 2 * W2, H = 2 * H2;
  for (long q = (long)blockIdx.x * GT + threadIdx.x; q < total; q += (long)gridDim.x * GT) {
    const int c = (int)(q % C);
    const long w = q / C;
    const int px = (int)(w % W2), py = (int)((w / W2) % H2);
    const long n = w / ((long)W2 * H2);
    const long o = ((n * C + c) * H2 + py) * (long)W2 + px;
    const float g = a[o] > 0.f ? da[o] : 0.f;
    const int bi = idx[o];
    float* base = dy + ((n * H + 2 * py) * (long)W + 2 * px) * C + c;
    base[0] = bi == 0 ? g : 0.f;
    base[C] = bi == 1 ? g : 0.f;
    base[(long)W * C] = bi == 2 ? g : 0.f;
    base[(long)W * C + C] = bi == 3 ? g : 0.f;
  }
}

inline int grid_for(long total) {
  long b = (total + GT - 1) / GT;
  const long cap = 64L * ss_device_cus();
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int ss_roi_norm(const uint8_t* R, int N, int HW, int standardize, float* xn, float* stats, ss_stream_t stream) {
  SS_REQUIRE(R && xn && N > 0 && HW > 1, SS_ERR_ARG);
  hipLaunchKernelGGL(roi_norm_kernel, dim3(N), dim3(GT), 0, static_cast<hipStream_t>(stream), R, HW, standardize, xn, stats);
  return ss_launch_status();
}

extern "C" int ss_im2col3x3(const float* src, int N, int C, int H, int W, float* col, int ld_col, ss_stream_t stream) {
  SS_REQUIRE(src && col && N > 0 && C > 0 && H > 0 && W > 0 && ld_col >= 9 * C, SS_ERR_ARG);
  const long total = (long)N * H * W * C;
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for(total)), dim3(GT), 0, static_cast<hipStream_t>(stream), src, total, C, H, W, col,
                     ld_col);
  return ss_launch_status();
}

extern "C" int ss_relu_pool2(const float* y, int N, int C, int H, int W, float* a, uint8_t* idx, ss_stream_t stream) {
  SS_REQUIRE(y && a && idx && N > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, SS_ERR_ARG);
  const long total = (long)N * C * (H / 2) * (W / 2);
  hipLaunchKernelGGL(relu_pool2_kernel, dim3(grid_for(total)), dim3(GT), 0, static_cast<hipStream_t>(stream), y, total, C, H, W, a, idx);
  return ss_launch_status();
}

extern "C" int ss_relu_mean(const float* y, int N, int P, int C, float* feat, uint8_t* mask, ss_stream_t stream) {
  SS_REQUIRE(y && feat && N > 0 && P > 0 && C > 0, SS_ERR_ARG);
  hipLaunchKernelGGL(relu_mean_kernel, dim3(N), dim3(GT), 0, static_cast<hipStream_t>(stream), y, P, C, feat, mask);
  return ss_launch_status();
}

extern "C" int ss_mask_scale(const uint8_t* mask, const float* dfeat, int N, int P, int C, float* dy, ss_stream_t stream) {
  SS_REQUIRE(mask && dfeat && dy && N > 0 && P > 0 && C > 0, SS_ERR_ARG);
  const long total = (long)N * P * C;
  hipLaunchKernelGGL(mask_scale_kernel, dim3(grid_for(total)), dim3(GT), 0, static_cast<hipStream_t>(stream), mask, dfeat, total, P, C, dy);
  return ss_launch_status();
}

extern "C" int ss_col2im3x3(const float* dcol, int ld_col, int N, int C, int H, int W, float* d, ss_stream_t stream) {
  SS_REQUIRE(dcol && d && N > 0 && C > 0 && H > 0 && W > 0 && ld_col >= 9 * C, SS_ERR_ARG);
  const long total = (long)N * C * H * W;
  hipLaunchKernelGGL(col2im3x3_kernel, dim3(grid_for(total)), dim3(GT), 0, static_cast<hipStream_t>(stream), dcol, ld_col, total, C, H, W, d);
  return ss_launch_status();
}

extern "C" int ss_pool2_bwd(const float* da, const float* a, const uint8_t* idx, int N, int C, int H2, int W2, float* dy,
                            ss_stream_t stream) {
  SS_REQUIRE(da && a && idx && dy && N > 0 && C > 0 && H2 > 0 && W2 > 0, SS_ERR_ARG);
  const long total = (long)N * C * H2 * W2;
  hipLaunchKernelGGL(pool2_bwd_kernel, dim3(grid_for(total)), dim3(GT), 0, static_cast<hipStream_t>(stream), da, a, idx, total, C, H2, W2, dy);
  return ss_launch_status();
}
