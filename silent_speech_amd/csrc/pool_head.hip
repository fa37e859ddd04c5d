// AttnPool, LayerNorm, dropout, label-smoothed cross entropy: the small HBM-bound ops around the
// GRU (/root/reference/train_model_official.py:231-248, 271-277, 405).  Wavefront-shuffle
// reductions, coalesced row reads; none of these is matmul-shaped enough for MFMA (the two head
// Linear layers go through ss_gemm_f32).
#include "ss_common.h"

namespace {

// ---------------------------------------------------------------------------------- AttnPool
// one 256-thread workgroup per clip; dynamic LDS: T floats (scores -> weights)
__global__ __launch_bounds__(256) void attn_pool_fwd_kernel(const float* __restrict__ h, const int* __restrict__ lengths,
                                                            const float* __restrict__ w_score,
                                                            const float* __restrict__ b_score, int T, int D,
                                                            float* __restrict__ attn, float* __restrict__ pooled) {
  extern __shared__ __attribute__((aligned(16))) float sc[];
  __shared__ float red[8];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int len = lengths[b];
  const float* hb = h + (long)b * T * D;
  const float bias = b_score[0];
  for (int t = wid; t < T; t += 4) {
    float s = 0.f;
    if (t < len) {
      for (int d = lane; d < D; d += 64) s += hb[(long)t * D + d] * w_score[d];
      s = wave_sum(s) + bias;
    } else {
      s = -1e9f;  // masked_fill(~mask, -1e9), train_model_official.py:245
    }
    if (lane == 0) sc[t] = s;
  }
  __syncthreads();
  // softmax over t
  float m = -3.4e38f;
  for (int t = threadIdx.x; t < T; t += 256) m = fmaxf(m, sc[t]);
  m = wave_max(m);
  if (lane == 0) red[wid] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) {
    float e = expf(sc[t] - m);
    sc[t] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wid] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int t = threadIdx.x; t < T; t += 256) {
    float wt = sc[t] * inv;
    sc[t] = wt;
    attn[(long)b * T + t] = wt;
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 256) {
    float acc = 0.f;
    for (int t = 0; t < len; ++t) acc += sc[t] * hb[(long)t * D + d];
    pooled[(long)b * D + d] = acc;
  }
}

// dynamic LDS: 2*T floats (c_t = dp.h_t, then ds_t)
__global__ __launch_bounds__(256) void attn_pool_bwd_kernel(const float* __restrict__ h, const int* __restrict__ lengths,
                                                            const float* __restrict__ w_score,
                                                            const float* __restrict__ attn,
                                                            const float* __restrict__ d_pooled, int T, int D,
                                                            float* __restrict__ d_h, float* __restrict__ g_w,
                                                            float* __restrict__ g_b) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ float red[4];
  float* ds = sm;       // [T]
  float* wt = sm + T;   // [T]
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int len = lengths[b];
  const float* hb = h + (long)b * T * D;
  const float* dp = d_pooled + (long)b * D;
  for (int t = wid; t < T; t += 4) {
    float c = 0.f;
    if (t < len) {
      for (int d = lane; d < D; d += 64) c += hb[(long)t * D + d] * dp[d];
      c = wave_sum(c);
    }
    if (lane == 0) {
      ds[t] = c;
      wt[t] = attn[(long)b * T + t];
    }
  }
  __syncthreads();
  float part = 0.f;
  for (int t = threadIdx.x; t < len; t += 256) part += wt[t] * ds[t];
  part = wave_sum(part);
  if (lane == 0) red[wid] = part;
  __syncthreads();
  const float dot = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += 256) ds[t] = (t < len) ? wt[t] * (ds[t] - dot) : 0.f;
  __syncthreads();
  float gb = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) {
    const float dpd = dp[d], wd = w_score[d];
    float gw = 0.f;
    for (int t = 0; t < T; ++t) {
      float v = 0.f;
      if (t < len) {
        v = wt[t] * dpd + ds[t] * wd;
        gw += ds[t] * hb[(long)t * D + d];
      }
      d_h[((long)b * T + t) * D + d] = v;
    }
    atomicAdd(&g_w[d], gw);
  }
  if (threadIdx.x == 0) {
    for (int t = 0; t < len; ++t) gb += ds[t];
    atomicAdd(g_b, gb);
  }
}

// ---------------------------------------------------------------------------------- LayerNorm
// one wave per row
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int B, int D, float eps,
                                                            float* __restrict__ y, float* __restrict__ xhat,
                                                            float* __restrict__ rstd) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= B) return;
  const float* xr = x + (long)row * D;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s += xr[d];
  const float mean = wave_sum(s) / D;
  float v = 0.f;
  for (int d = lane; d < D; d += 64) {
    float c = xr[d] - mean;
    v += c * c;
  }
  const float rs = rsqrtf(wave_sum(v) / D + eps);
  for (int d = lane; d < D; d += 64) {
    float xh = (xr[d] - mean) * rs;
    if (xhat) xhat[(long)row * D + d] = xh;
    y[(long)row * D + d] = xh * gamma[d] + beta[d];
  }
  if (rstd && lane == 0) rstd[row] = rs;
}

__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ d_y, const float* __restrict__ xhat,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, int B, int D,
                                                            float* __restrict__ d_x, float* __restrict__ g_gamma,
                                                            float* __restrict__ g_beta) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= B) return;
  const float* dy = d_y + (long)row * D;
  const float* xh = xhat + (long)row * D;
  float s1 = 0.f, s2 = 0.f;
  for (int d = lane; d < D; d += 64) {
    float dxh = dy[d] * gamma[d];
    s1 += dxh;
    s2 += dxh * xh[d];
  }
  s1 = wave_sum(s1) / D;
  s2 = wave_sum(s2) / D;
  const float rs = rstd[row];
  for (int d = lane; d < D; d += 64) {
    float dxh = dy[d] * gamma[d];
    d_x[(long)row * D + d] = rs * (dxh - s1 - xh[d] * s2);
    atomicAdd(&g_gamma[d], dy[d] * xh[d]);
    atomicAdd(&g_beta[d], dy[d]);
  }
}

// ---------------------------------------------------------------------------------- dropout
// Philox4x32-10, one counter per 4 elements.

__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float p,
                                                      uint64_t seed, uint64_t offset,
                                                      const float* __restrict__ relu_of) {
  const float scale = 1.0f / (1.0f - p);
  const uint32_t thr = (uint32_t)((double)p * 4294967296.0);
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q * 4 < n; q += (long)gridDim.x * 256) {
    uint32_t rnd[4];
    uint64_t ctr = offset + (uint64_t)q;
    philox4((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), rnd);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      long idx = q * 4 + e;
      if (idx < n) {
        float v = x[idx];
        if (p > 0.f) v = (rnd[e] >= thr) ? v * scale : 0.f;
        if (relu_of && relu_of[idx] <= 0.f) v = 0.f;
        y[idx] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------- cross entropy
// one thread per clip (C is a handful of words)
__global__ __launch_bounds__(256) void ce_ls_kernel(const float* __restrict__ logits, const int64_t* __restrict__ y, int B,
                                                    int C, float eps, float denom, float* __restrict__ d_logits,
                                                    float* __restrict__ loss_sum, int* __restrict__ correct) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  float loss = 0.f;
  int ok = 0;
  if (b < B) {
    const float* lr = logits + (long)b * C;
    const int yy = (int)y[b];
    float m = lr[0];
    int am = 0;
    for (int c = 1; c < C; ++c)
      if (lr[c] > m) { m = lr[c]; am = c; }
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(lr[c] - m);
    const float lse = m + logf(se);
    float slp = 0.f;
    for (int c = 0; c < C; ++c) slp += lr[c] - lse;
    loss = (1.0f - eps) * (lse - lr[yy]) + eps * (-slp / C);
    if (d_logits) {
      for (int c = 0; c < C; ++c) {
        float pr = expf(lr[c] - lse);
        float tgt = (c == yy ? (1.0f - eps) : 0.f) + eps / C;
        d_logits[(long)b * C + c] = (pr - tgt) / denom;
      }
    }
    ok = (am == yy);
  }
  loss = wave_sum(loss / denom);
  if ((threadIdx.x & 63) == 0 && loss_sum) atomicAdd(loss_sum, loss);
  if (correct && ok) atomicAdd(correct, 1);
}

// softmax + top-k of every logit row (live_infer_official.py:223-226): one wave per clip, probabilities as
// exp(l - max) / sum like torch.softmax, k selection sweeps (largest first, lowest index among equals)
__global__ __launch_bounds__(256) void softmax_topk_kernel(const float* __restrict__ logits, int B, int C, int k,
                                                           float* __restrict__ probs, int32_t* __restrict__ idx) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;  // whole waves leave together: no wave-wide operation below is left short of lanes
  const float* lr = logits + (long)b * C;
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, lr[c]);
  m = wave_max(m);
  float se = 0.f;
  for (int c = lane; c < C; c += 64) se += expf(lr[c] - m);
  se = wave_sum(se);
  float last_v = INFINITY;
  int last_i = -1;
  for (int j = 0; j < k; ++j) {
    // best entry that comes after (last_v, last_i) in the order "value descending, index ascending"
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
      const float v = lr[c];
      const bool after = v < last_v || (v == last_v && c > last_i);
      if (after && (v > bv || (v == bv && c < bi))) { bv = v; bi = c; }
    }
    const float wv = wave_max(bv);
    int cand = (bv == wv) ? bi : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    last_v = wv;
    last_i = cand;
    if (lane == 0) {
      const bool have = cand != 0x7fffffff;
      probs[(long)b * k + j] = have ? expf(wv - m) / se : 0.f;
      idx[(long)b * k + j] = have ? cand : -1;
    }
  }
}

}  // namespace

extern "C" int ss_softmax_topk(const float* logits, int B, int C, int k, float* probs, int32_t* idx, ss_stream_t stream) {
  SS_REQUIRE(logits && probs && idx && B > 0 && C > 0 && k > 0, SS_ERR_ARG);
  SS_REQUIRE(k <= 64, SS_ERR_UNSUPPORTED);
  hipLaunchKernelGGL(softmax_topk_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), logits, B, C,
                     k, probs, idx);
  return ss_launch_status();
}

extern "C" int ss_attn_pool_fwd(const float* h, const int32_t* lengths, const float* w_score, const float* b_score,
                                int B, int T, int D, float* attn, float* pooled, ss_stream_t stream) {
  SS_REQUIRE(h && lengths && w_score && b_score && attn && pooled && B > 0 && T > 0 && D > 0, SS_ERR_ARG);
  SS_REQUIRE(T <= 8192, SS_ERR_UNSUPPORTED);
  hipLaunchKernelGGL(attn_pool_fwd_kernel, dim3(B), dim3(256), T * sizeof(float), static_cast<hipStream_t>(stream), h,
                     lengths, w_score, b_score, T, D, attn, pooled);
  return ss_launch_status();
}

extern "C" int ss_attn_pool_bwd(const float* h, const int32_t* lengths, const float* w_score, const float* attn,
                                const float* d_pooled, int B, int T, int D, float* d_h, float* g_w, float* g_b,
                                ss_stream_t stream) {
  SS_REQUIRE(h && lengths && w_score && attn && d_pooled && d_h && g_w && g_b && B > 0 && T > 0 && D > 0, SS_ERR_ARG);
  SS_REQUIRE(T <= 4096, SS_ERR_UNSUPPORTED);
  hipLaunchKernelGGL(attn_pool_bwd_kernel, dim3(B), dim3(256), 2 * T * sizeof(float), static_cast<hipStream_t>(stream),
                     h, lengths, w_score, attn, d_pooled, T, D, d_h, g_w, g_b);
  return ss_launch_status();
}

extern "C" int ss_layernorm_fwd(const float* x, const float* gamma, const float* beta, int B, int D, float eps,
                                float* y, float* xhat, float* rstd, ss_stream_t stream) {
  SS_REQUIRE(x && gamma && beta && y && B > 0 && D > 0, SS_ERR_ARG);
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     gamma, beta, B, D, eps, y, xhat, rstd);
  return ss_launch_status();
}

extern "C" int ss_layernorm_bwd(const float* d_y, const float* xhat, const float* rstd, const float* gamma, int B,
                                int D, float* d_x, float* g_gamma, float* g_beta, ss_stream_t stream) {
  SS_REQUIRE(d_y && xhat && rstd && gamma && d_x && g_gamma && g_beta && B > 0 && D > 0, SS_ERR_ARG);
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), d_y,
                     xhat, rstd, gamma, B, D, d_x, g_gamma, g_beta);
  return ss_launch_status();
}

extern "C" int ss_dropout(const float* x, float* y, long n, float p, uint64_t seed, uint64_t offset,
                          const float* relu_of, ss_stream_t stream) {
  SS_REQUIRE(x && y && n > 0 && p >= 0.f && p < 1.f, SS_ERR_ARG);
  long quads = (n + 3) / 4;
  int blocks = (int)((quads + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dropout_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, n, p, seed,
                     offset, relu_of);
  return ss_launch_status();
}

extern "C" int ss_ce_ls_fwd_bwd(const float* logits, const int64_t* y, int B, int C, float label_smoothing,
                                float denom, float* d_logits, float* loss_sum, int32_t* correct,
                                ss_stream_t stream) {
  SS_REQUIRE(logits && y && B > 0 && C > 0 && denom > 0.f, SS_ERR_ARG);
  hipLaunchKernelGGL(ce_ls_kernel, dim3(ceil_div(B, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), logits, y,
                     B, C, label_smoothing, denom, d_logits, loss_sum, correct);
  return ss_launch_status();
}
