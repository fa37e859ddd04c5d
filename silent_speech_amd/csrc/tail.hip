// Fused classifier tail: AttnPool -> LayerNorm -> Linear -> ReLU -> Dropout -> Linear (-> label-smoothed CE),
// one 256-thread workgroup per clip, and its backward down to the gradient of the GRU output.
//
// Replaces /root/reference/train_model_official.py:231-248 (AttnPool), :271-277 (head) and :405 (loss) as one
// launch each way: at B = 256 the seven separate forward ops and six backward ops are latency-bound launches
// (~15-30 us each, 0.23 ms per step) around a few hundred kFLOP per clip.  Everything per clip stays in LDS /
// registers; the two weight matrices stream from L2 (196 KB per workgroup).  The weight gradients of the two
// Linear layers stay batched GEMMs over the clips (ss_gemm_f32, K = B) on what this kernel stashes.
#include "ss_common.h"

namespace {

constexpr int TNT = 1024;  // 16 waves per clip: the stages are chains of L2 / HBM latencies, more waves = fewer links per chain
constexpr int NWT = TNT / 64;


// keep-scale of element idx of a dropout stream: the same stream ss_dropout draws (one Philox counter per 4 elements)
__device__ __forceinline__ float drop_scale(long idx, float p, uint64_t seed, uint64_t offset) {
  if (p <= 0.f) return 1.f;
  uint32_t rnd[4];
  const uint64_t ctr = offset + (uint64_t)(idx >> 2);
  philox4((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), rnd);
  const uint32_t thr = (uint32_t)((double)p * 4294967296.0);
  return rnd[idx & 3] >= thr ? 1.0f / (1.0f - p) : 0.f;
}

__device__ __forceinline__ float block_sum(float v, float* red) {  // NWT waves; all threads get the sum
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NWT; ++k) s += red[k];
  return s;
}

struct TailFwdParams {
  const float* h;        // (B,T,D)
  const int* lengths;
  const float *w_score, *b_score, *gamma, *beta, *w1, *b1, *w4, *b4;
  const int64_t* y;      // may be null: no loss
  int B, T, D, MID, C;
  float eps, drop_p, ls, denom;
  uint64_t seed, offset;
  float *attn, *xhat, *rstd, *ln, *mid, *mid_d, *logits, *d_logits, *loss_sum;
  int* correct;
};

// dynamic LDS: sc[T] | pooled[D] | lnv[D] | midv[MID] | lg[C]
__global__ __launch_bounds__(TNT) void tail_fwd_kernel(TailFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ float red[2 * NWT];
  const int T = p.T, D = p.D, MID = p.MID, C = p.C;
  float* sc = sm;
  float* pooled = sc + T;
  float* lnv = pooled + D;
  float* midv = lnv + D;
  float* lg = midv + MID;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int len = p.lengths[b];
  const float* hb = p.h + (long)b * T * D;

  // Linear(D -> MID) weights of this wave's rows, fetched NOW: the loop that consumed them straight from memory (one L2 round
  // trip per 64 columns, twelve in a row) was 20 of the kernel's 33 us; here the loads fly under the attention phases.
  // Wave wv owns rows 4 wv + u of each 64-row pass; PF_PASS x PF_IT covers MID <= 128, D <= 384, the rest takes the loop.
  constexpr int PF_PASS = 2, PF_IT = 6, PF_CH = 4;
  float wreg[PF_PASS][4][PF_IT];
#pragma unroll
  for (int ps = 0; ps < PF_PASS; ++ps)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int it = 0; it < PF_IT; ++it) {
        const int o = 4 * wv + u + 4 * NWT * ps, d = lane + 64 * it;
        wreg[ps][u][it] = (o < MID && d < D) ? p.w1[(long)o * D + d] : 0.f;
      }

  // ---- AttnPool: scores, masked softmax over t, weighted sum
  const float bsc = p.b_score[0];
  for (int t = wv; t < T; t += NWT) {
    float s = -1e9f;  // masked_fill(~mask, -1e9), train_model_official.py:245
    if (t < len) {
      s = 0.f;
      float hv[PF_IT], sw[PF_IT];  // all loads of a row first
#pragma unroll
      for (int it = 0; it < PF_IT; ++it) {
        const int d = lane + 64 * it;
        hv[it] = d < D ? hb[(long)t * D + d] : 0.f;
        sw[it] = d < D ? p.w_score[d] : 0.f;
      }
#pragma unroll
      for (int it = 0; it < PF_IT; ++it) s += hv[it] * sw[it];
      for (int d0 = lane + 64 * PF_IT; d0 < D; d0 += 64 * PF_CH) {  // (D > 384: config 5's 2H = 1024) PF_CH loads per round trip
        float h2[PF_CH], w2[PF_CH];
#pragma unroll
        for (int it = 0; it < PF_CH; ++it) {
          const int d = d0 + 64 * it;
          h2[it] = d < D ? hb[(long)t * D + d] : 0.f;
          w2[it] = d < D ? p.w_score[d] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < PF_CH; ++it) s += h2[it] * w2[it];
      }
      s = wave_sum(s) + bsc;
    }
    if (lane == 0) sc[t] = s;
  }
  __syncthreads();
  float m = -3.4e38f;
  for (int t = tid; t < T; t += TNT) m = fmaxf(m, sc[t]);
  m = wave_max(m);
  if (lane == 0) red[wv] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int k = 1; k < NWT; ++k) m = fmaxf(m, red[k]);
  float sum = 0.f;
  for (int t = tid; t < T; t += TNT) {
    const float e = expf(sc[t] - m);
    sc[t] = e;
    sum += e;
  }
  sum = block_sum(sum, red + NWT);
  const float inv = 1.0f / sum;
  for (int t = tid; t < T; t += TNT) {
    const float wt = sc[t] * inv;
    sc[t] = wt;
    if (p.attn) p.attn[(long)b * T + t] = wt;
  }
  __syncthreads();
  float s1 = 0.f;
  for (int d = tid; d < D; d += TNT) {
    float acc = 0.f;
    for (int t = 0; t < len; ++t) acc += sc[t] * hb[(long)t * D + d];
    pooled[d] = acc;
    s1 += acc;
  }
  // ---- LayerNorm
  const float mean = block_sum(s1, red) / D;
  float s2 = 0.f;
  for (int d = tid; d < D; d += TNT) {
    const float c = pooled[d] - mean;
    s2 += c * c;
  }
  const float rs = rsqrtf(block_sum(s2, red + NWT) / D + p.eps);
  for (int d = tid; d < D; d += TNT) {
    const float xh = (pooled[d] - mean) * rs;
    const float v = xh * p.gamma[d] + p.beta[d];
    lnv[d] = v;
    if (p.xhat) p.xhat[(long)b * D + d] = xh;
    if (p.ln) p.ln[(long)b * D + d] = v;
  }
  if (p.rstd && tid == 0) p.rstd[b] = rs;
  __syncthreads();
  // ---- Linear(D -> MID) + ReLU + Dropout: a wave per output row, lanes across the row (coalesced weight reads);
  // four rows at a time so the four shuffle-reduction chains interleave
  float xv[PF_IT];
#pragma unroll
  for (int it = 0; it < PF_IT; ++it) xv[it] = (lane + 64 * it < D) ? lnv[lane + 64 * it] : 0.f;
  int ps = 0;
  for (int o0 = 4 * wv; o0 < MID; o0 += 4 * NWT, ++ps) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (ps < PF_PASS) {
#pragma unroll
      for (int it = 0; it < PF_IT; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] += (ps == 0 ? wreg[0][u][it] : wreg[1][u][it]) * xv[it];
    }
    // the columns the prefetch did not cover (D > 384): PF_CH x 4 loads in flight per L2 round trip instead of 4 -- at D = 1024
    // the one-column-group-at-a-time loop was twenty dependent round trips per workgroup
    for (int d0 = lane + (ps < PF_PASS ? 64 * PF_IT : 0); d0 < D; d0 += 64 * PF_CH) {
      float w2[PF_CH][4];
#pragma unroll
      for (int it = 0; it < PF_CH; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int d = d0 + 64 * it;
          w2[it][u] = (d < D && o0 + u < MID) ? p.w1[(long)(o0 + u) * D + d] : 0.f;
        }
#pragma unroll
      for (int it = 0; it < PF_CH; ++it) {
        const int d = d0 + 64 * it;
        const float x = d < D ? lnv[d] : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] += w2[it][u] * x;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = wave_sum(acc[u]);
    if (lane < 4 && o0 + lane < MID) {
      const int o = o0 + lane;
      const float a = relu_f((lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3]) + p.b1[o]);
      const float ad = a * drop_scale((long)b * MID + o, p.drop_p, p.seed, p.offset);
      midv[o] = ad;
      if (p.mid) p.mid[(long)b * MID + o] = a;
      if (p.mid_d) p.mid_d[(long)b * MID + o] = ad;
    }
  }
  __syncthreads();
  // ---- Linear(MID -> C)
  for (int c = wv; c < C; c += NWT) {
    float acc = 0.f;
    const float* wr = p.w4 + (long)c * MID;
    for (int o = lane; o < MID; o += 64) acc += wr[o] * midv[o];
    acc = wave_sum(acc);
    if (lane == 0) {
      const float v = acc + p.b4[c];
      lg[c] = v;
      p.logits[(long)b * C + c] = v;
    }
  }
  if (!p.y) return;
  __syncthreads();
  // ---- CrossEntropyLoss(label_smoothing), mean over `denom` clips; d(loss)/d(logits)
  if (wv == 0) {
    const int yy = (int)p.y[b];
    float mx = -3.4e38f;
    int am = 0;
    for (int c = lane; c < C; c += 64)
      if (lg[c] > mx) { mx = lg[c]; am = c; }
    // wave arg-max, first index on ties (torch.argmax)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(mx, o, 64);
      const int oa = __shfl_xor(am, o, 64);
      if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
    }
    float se = 0.f, sl = 0.f;
    for (int c = lane; c < C; c += 64) se += expf(lg[c] - mx);
    se = wave_sum(se);
    const float lse = mx + logf(se);
    for (int c = lane; c < C; c += 64) sl += lg[c] - lse;
    sl = wave_sum(sl);
    for (int c = lane; c < C; c += 64) {
      const float pr = expf(lg[c] - lse);
      const float tgt = (c == yy ? (1.0f - p.ls) : 0.f) + p.ls / C;
      p.d_logits[(long)b * C + c] = (pr - tgt) / p.denom;
    }
    if (lane == 0) {
      const float loss = (1.0f - p.ls) * (lse - lg[yy]) + p.ls * (-sl / C);
      if (p.loss_sum) atomicAdd(p.loss_sum, loss / p.denom);
      if (p.correct && am == yy) atomicAdd(p.correct, 1);
    }
  }
}

struct TailBwdParams {
  const float* h;
  const int* lengths;
  const float *w_score, *gamma, *w1, *w4;
  const float *attn, *xhat, *rstd, *mid, *d_logits;
  int B, T, D, MID, C;
  float drop_p;
  uint64_t seed, offset;
  float *d_mid, *d_h, *g_gamma, *g_beta, *g_wscore, *g_bscore;
  float* col_part;  // NULL, or [B][3][D]: this clip's terms of g_gamma | g_beta | g_wscore, stored instead of added atomically
};

// dynamic LDS: dmid[MID] | dp[D] | ds[T] | wt[T] | dl[C]
__global__ __launch_bounds__(TNT) void tail_bwd_kernel(TailBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ float red[2 * NWT];
  const int T = p.T, D = p.D, MID = p.MID, C = p.C;
  float* dmid = sm;
  float* dp = dmid + MID;
  float* ds = dp + D;
  float* wt = ds + T;
  float* dl = wt + T;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int len = p.lengths[b];
  const float* hb = p.h + (long)b * T * D;

  for (int c = tid; c < C; c += TNT) dl[c] = p.d_logits[(long)b * C + c];
  for (int t = tid; t < T; t += TNT) wt[t] = p.attn[(long)b * T + t];
  __syncthreads();
  // ---- through Linear(MID -> C), Dropout, ReLU
  for (int o = tid; o < MID; o += TNT) {
    // (sixteen weight loads per L2 round trip here and below: four at a time, as the compiler unrolls these loops by itself,
    // made 25 + 32 dependent round trips per workgroup at C = 100, MID = 128 -- most of the kernel)
    float acc = 0.f;
#pragma unroll 16
    for (int c = 0; c < C; ++c) acc += dl[c] * p.w4[(long)c * MID + o];
    acc *= drop_scale((long)b * MID + o, p.drop_p, p.seed, p.offset);
    if (p.mid[(long)b * MID + o] <= 0.f) acc = 0.f;
    dmid[o] = acc;
    p.d_mid[(long)b * MID + o] = acc;
  }
  __syncthreads();
  // ---- through Linear(D -> MID) and LayerNorm
  float s1 = 0.f, s2 = 0.f;
  for (int d = tid; d < D; d += TNT) {
    float acc = 0.f;
#pragma unroll 16
    for (int o = 0; o < MID; ++o) acc += dmid[o] * p.w1[(long)o * D + d];
    const float xh = p.xhat[(long)b * D + d];
    // 256 clips adding to the same D addresses serialise at the memory side (~30 ns per add: 8 us of this kernel); with a
    // scratch row per clip the sums are taken by a column-sum pass off the critical path
    if (p.col_part) {
      p.col_part[((long)b * 3 + 0) * D + d] = acc * xh;
      p.col_part[((long)b * 3 + 1) * D + d] = acc;
    } else {
      atomicAdd(&p.g_gamma[d], acc * xh);
      atomicAdd(&p.g_beta[d], acc);
    }
    const float dxh = acc * p.gamma[d];
    dp[d] = dxh;
    s1 += dxh;
    s2 += dxh * xh;
  }
  s1 = block_sum(s1, red) / D;
  s2 = block_sum(s2, red + NWT) / D;
  const float rs = p.rstd[b];
  for (int d = tid; d < D; d += TNT) dp[d] = rs * (dp[d] - s1 - p.xhat[(long)b * D + d] * s2);  // d pooled
  __syncthreads();
  // ---- AttnPool backward
  for (int t = wv; t < T; t += NWT) {
    float c = 0.f;
    if (t < len) {
      for (int d = lane; d < D; d += 64) c += hb[(long)t * D + d] * dp[d];
      c = wave_sum(c);
    }
    if (lane == 0) ds[t] = c;
  }
  __syncthreads();
  float part = 0.f;
  for (int t = tid; t < len; t += TNT) part += wt[t] * ds[t];
  const float dot = block_sum(part, red);
  for (int t = tid; t < T; t += TNT) ds[t] = (t < len) ? wt[t] * (ds[t] - dot) : 0.f;
  __syncthreads();
  for (int d = tid; d < D; d += TNT) {
    const float dpd = dp[d], wd = p.w_score[d];
    float gw = 0.f;  // (the loads of h apart from the stores of d h: one load per round trip when they shared a loop)
#pragma unroll 8
    for (int t = 0; t < len; ++t) gw += ds[t] * hb[(long)t * D + d];
    for (int t = 0; t < T; ++t) p.d_h[((long)b * T + t) * D + d] = (t < len) ? wt[t] * dpd + ds[t] * wd : 0.f;
    if (p.col_part) p.col_part[((long)b * 3 + 2) * D + d] = gw;
    else atomicAdd(&p.g_wscore[d], gw);
  }
  if (tid == 0) {
    float gb = 0.f;
    for (int t = 0; t < len; ++t) gb += ds[t];
    atomicAdd(p.g_bscore, gb);
  }
}

}  // namespace

extern "C" int ss_tail_fwd(const float* h, const int32_t* lengths, const float* w_score, const float* b_score,
                           const float* gamma, const float* beta, const float* w1, const float* b1, const float* w4,
                           const float* b4, const int64_t* y, int B, int T, int D, int MID, int C, float ln_eps,
                           float drop_p, uint64_t seed, uint64_t offset, float label_smoothing, float denom,
                           float* attn, float* xhat, float* rstd, float* ln, float* mid, float* mid_d, float* logits,
                           float* d_logits, float* loss_sum, int32_t* correct, ss_stream_t stream) {
  SS_REQUIRE(h && lengths && w_score && b_score && gamma && beta && w1 && b1 && w4 && b4 && logits, SS_ERR_ARG);
  SS_REQUIRE(B > 0 && T > 0 && D > 0 && MID > 0 && C > 0 && drop_p >= 0.f && drop_p < 1.f, SS_ERR_ARG);
  SS_REQUIRE(!y || (d_logits && denom > 0.f), SS_ERR_ARG);
  const size_t lds = (size_t)(T + 2 * D + MID + C) * sizeof(float);
  SS_REQUIRE(lds <= 60 * 1024, SS_ERR_UNSUPPORTED);
  TailFwdParams p;
  p.h = h; p.lengths = lengths; p.w_score = w_score; p.b_score = b_score; p.gamma = gamma; p.beta = beta;
  p.w1 = w1; p.b1 = b1; p.w4 = w4; p.b4 = b4; p.y = y;
  p.B = B; p.T = T; p.D = D; p.MID = MID; p.C = C;
  p.eps = ln_eps; p.drop_p = drop_p; p.ls = label_smoothing; p.denom = denom; p.seed = seed; p.offset = offset;
  p.attn = attn; p.xhat = xhat; p.rstd = rstd; p.ln = ln; p.mid = mid; p.mid_d = mid_d; p.logits = logits;
  p.d_logits = d_logits; p.loss_sum = loss_sum; p.correct = correct;
  hipLaunchKernelGGL(tail_fwd_kernel, dim3(B), dim3(TNT), lds, static_cast<hipStream_t>(stream), p);
  return ss_launch_status();
}

extern "C" int ss_tail_bwd(const float* h, const int32_t* lengths, const float* w_score, const float* gamma,
                           const float* w1, const float* w4, const float* attn, const float* xhat, const float* rstd,
                           const float* mid, const float* d_logits, int B, int T, int D, int MID, int C, float drop_p,
                           uint64_t seed, uint64_t offset, float* d_mid, float* d_h, float* g_gamma, float* g_beta,
                           float* g_wscore, float* g_bscore, float* col_part, ss_stream_t stream) {
  SS_REQUIRE(h && lengths && w_score && gamma && w1 && w4 && attn && xhat && rstd && mid && d_logits, SS_ERR_ARG);
  SS_REQUIRE(d_mid && d_h && g_gamma && g_beta && g_wscore && g_bscore, SS_ERR_ARG);
  SS_REQUIRE(B > 0 && T > 0 && D > 0 && MID > 0 && C > 0 && drop_p >= 0.f && drop_p < 1.f, SS_ERR_ARG);
  const size_t lds = (size_t)(MID + D + 2 * T + C) * sizeof(float);
  SS_REQUIRE(lds <= 60 * 1024, SS_ERR_UNSUPPORTED);
  TailBwdParams p;
  p.h = h; p.lengths = lengths; p.w_score = w_score; p.gamma = gamma; p.w1 = w1; p.w4 = w4;
  p.attn = attn; p.xhat = xhat; p.rstd = rstd; p.mid = mid; p.d_logits = d_logits;
  p.B = B; p.T = T; p.D = D; p.MID = MID; p.C = C; p.drop_p = drop_p; p.seed = seed; p.offset = offset;
  p.d_mid = d_mid; p.d_h = d_h; p.g_gamma = g_gamma; p.g_beta = g_beta; p.g_wscore = g_wscore; p.g_bscore = g_bscore;
  p.col_part = col_part;
  hipLaunchKernelGGL(tail_bwd_kernel, dim3(B), dim3(TNT), lds, static_cast<hipStream_t>(stream), p);
  return ss_launch_status();
}
