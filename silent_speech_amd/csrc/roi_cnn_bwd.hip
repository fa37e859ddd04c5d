// Backward of the fused ROI normalise + TinyROICNN block w.r.t. its eight parameter tensors
// (the uint8 image has no gradient).  Autograd counterpart of
// /root/reference/train_model_official.py:212-229 as driven by loss.backward() (:437).
//
// One persistent 512-thread workgroup per CU walks frames.  Per frame it reads the forward's stash
// (pooled maps a1, a2, pool argmaxes, conv3 sign mask, averaged features: 67 KB for 64x64) plus the
// 4 KB uint8 frame, and runs five contractions out of LDS:
//
//   S1  dW3[n][c][tap] += sum_p dy3[n][p] * a2[c][p+tap]     MFMA  M=n(24->32) N=c(16)/tap K=pixels
//   S2  da2[c][p] = sum_{n,tap} dy3[n][p-tap] * W3[n][c][tap] MFMA  M=pixels N=c(16) K=(n,tap)=216
//   S3  dW2[n][c][tap] += sum_x dy2[n][x] * a1[c][x+tap]      MFMA  M=n(16) N=(tap,c)=72->80 K=pixels
//   S4  da1[c][x] = sum_{n,tap} dy2[n][x-tap] * W2[n][c][tap] MFMA  M=pixels N=c(8->16) K=(n,tap)=144
//   S5  dW1[c][tap] += sum_q da1[c][q] * x[2q+argmax+tap]     VALU gather (pool sparsity: 1 of 4 live)
//
// dy2 (the gradient before the second max-pool) is never materialised: the A operand of S3/S4 is
// formed on the fly from the pooled gradient and the 2-bit argmax.  Weight-gradient partial sums
// stay in registers for the whole frame walk (the K = pixel dimension is split over the 8 waves)
// and are reduced through LDS, then one float atomic per element per workgroup, at the end.
#include "ss_common.h"
#include "roi_cnn_geom.h"

namespace {

constexpr int NT = 512;
constexpr int NWV = NT / 64;
constexpr int MAXCH = 3;

struct BwdLayout {
  int o_a1h, o_U, usize, o_C, o_W, o_misc, total;  // float offsets
  int xss;                                          // row stride of the un-haloed normalised image
};

static inline BwdLayout make_bwd_layout(const CnnGeom& g) {
  BwdLayout L;
  const int HW2 = g.H2 * g.W2;
  L.xss = g.W + 1;
  L.o_a1h = 0;
  L.o_U = 8 * g.P1;
  const int u1 = 16 * g.P2 + 24 * g.P2;             // a2h | dy3h
  const int u2 = 8 * HW2 + 2 * HW2 + g.H * L.xss;   // da1 | i1 (bytes) | x
  L.usize = ((u1 > u2 ? u1 : u2) + 3) & ~3;
  L.o_C = L.o_U + L.usize;
  L.o_W = L.o_C + 16 * g.P + 4 * g.P;               // da2m | i2 (bytes)
  L.o_misc = L.o_W + 3456 + 1152;
  L.total = L.o_misc + 512;
  return L;
}

struct CnnBwdParams {
  const uint8_t* R;
  int N, standardize;
  const float *w2, *w3, *wfc;
  int E;
  const float* st_a1;
  const uint8_t* st_i1;
  const float* st_a2;
  const uint8_t* st_i2;
  const uint8_t* st_m3;
  const float* st_feat;
  const float* d_out;
  int ld_dout;
  float *g_w1, *g_b1, *g_w2, *g_b2, *g_w3, *g_b3, *g_wfc, *g_bfc;
  CnnGeom g;
  BwdLayout L;
};

__global__ __launch_bounds__(NT) void roi_cnn_bwd_kernel(CnnBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const CnnGeom& G = p.g;
  const int P = G.P, HW2 = G.H2 * G.W2, HW = G.H * G.W;
  const int W2 = G.W2, W4 = G.W4, S1 = G.S1, S2 = G.S2, P1 = G.P1, P2 = G.P2, XSS = p.L.xss;
  float* a1h = lds + p.L.o_a1h;
  float* U = lds + p.L.o_U;
  float* a2h = U;
  float* dy3h = U + 16 * P2;
  float* da1 = U;
  uint8_t* i1b = reinterpret_cast<uint8_t*>(U + 8 * HW2);
  float* xs = U + 8 * HW2 + 2 * HW2;
  float* da2m = lds + p.L.o_C;
  uint8_t* i2b = reinterpret_cast<uint8_t*>(da2m + 16 * P);
  float* w3s = lds + p.L.o_W;
  float* w2s = w3s + 3456;
  float* misc = lds + p.L.o_misc;
  float* s_dout = misc;          // [64]
  float* s_feat = misc + 64;     // [32]
  float* s_dfeat = misc + 96;    // [32]  d feat[c] / P
  float* s_stat = misc + 128;    // mu, sd
  unsigned* s_red = reinterpret_cast<unsigned*>(misc + 136);  // [2*NWV]
  float* s_gb3 = misc + 160;     // [32]  accumulated over the frame walk

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wvu = __builtin_amdgcn_readfirstlane(wv);
  const int i = lane & 15, g = lane >> 4;
  const int E = p.E;

  for (int q = tid; q < p.L.total; q += NT) lds[q] = 0.f;
  __syncthreads();
  for (int q = tid; q < 3456; q += NT) w3s[q] = p.w3[q];
  for (int q = tid; q < 1152; q += NT) w2s[q] = p.w2[q];

  // persistent per-thread accumulators
  f32x4 acc3[2][9], acc2[5];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 9; ++b) acc3[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 5; ++a) acc2[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  float acc1[10];
#pragma unroll
  for (int a = 0; a < 10; ++a) acc1[a] = 0.f;
  float accfc[3] = {0.f, 0.f, 0.f}, accbfc = 0.f, accb2 = 0.f;

  // S3 B-operand offsets: column idx = 16*nt + i  ->  tap = idx/8, c = idx%8
  int boff[5];
#pragma unroll
  for (int nt = 0; nt < 5; ++nt) {
    int idx = 16 * nt + i;
    int tap = idx >> 3, c = idx & 7;
    boff[nt] = (idx < 72) ? c * P1 + (tap / 3) * S1 + (tap % 3) : -1;
  }

  uint4 px[MAXCH], ix1[2];
  auto load_frame = [&](int n) {
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
      int q = tid + k * NT;
      if (q * 16 < HW) px[k] = reinterpret_cast<const uint4*>(p.R + (long)n * HW)[q];
    }
  };
  if ((int)blockIdx.x < p.N) load_frame(blockIdx.x);
  __syncthreads();

  for (int n = blockIdx.x; n < p.N; n += gridDim.x) {
    // ---------------- L0: clear the union region, stage the small vectors, pixel statistics
    for (int q = tid; q < p.L.usize / 4; q += NT) reinterpret_cast<f32x4*>(U)[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < E) s_dout[tid] = p.d_out[(long)n * p.ld_dout + tid];
    if (tid < 24) s_feat[tid] = p.st_feat[(long)n * 24 + tid];
    {
      unsigned su = 0, sq = 0;
#pragma unroll
      for (int k = 0; k < MAXCH; ++k) {
        int q = tid + k * NT;
        if (q * 16 < HW) {
          const unsigned wds[4] = {px[k].x, px[k].y, px[k].z, px[k].w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              unsigned u = (wds[e] >> (8 * b)) & 255u;
              su += u;
              sq += u * u;
            }
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        su += __shfl_xor(su, o, 64);
        sq += __shfl_xor(sq, o, 64);
      }
      if (lane == 0) { s_red[2 * wv] = su; s_red[2 * wv + 1] = sq; }
    }
    // the 8*HW2-byte argmax map of pool 1: up to two 16-byte pieces per thread (HW2 <= 2048)
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if ((tid + k * NT) * 16 < 8 * HW2)
        ix1[k] = reinterpret_cast<const uint4*>(p.st_i1 + (long)n * 8 * HW2)[tid + k * NT];
    __syncthreads();  // A

    if (tid < 24) {
      float s = 0.f;
      for (int e = 0; e < E; ++e) s += s_dout[e] * p.wfc[e * 24 + tid];
      s_dfeat[tid] = s / (float)P;
    }
    if (tid == 32) {
      unsigned long long tsu = 0, tsq = 0;
      for (int k = 0; k < NWV; ++k) { tsu += s_red[2 * k]; tsq += s_red[2 * k + 1]; }
      float mu = 0.f, sd = 1.f;
      if (p.standardize) {
        const double nn = (double)HW;
        mu = (float)((double)tsu / nn) / 255.0f;
        double var = ((double)tsq - (double)tsu * (double)tsu / nn) / (nn - 1.0);
        sd = (float)(sqrt(var > 0.0 ? var : 0.0) / 255.0);
        sd = fmaxf(sd, 1e-6f);
      }
      s_stat[0] = mu;
      s_stat[1] = sd;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int idx = tid + k * NT;
      if (idx < E * 24) accfc[k] += s_dout[idx / 24] * s_feat[idx % 24];
    }
    if (tid < E) accbfc += s_dout[tid];
    for (int q = tid; q < 16 * P; q += NT) {
      int c = q / P, r = q % P;
      a2h[c * P2 + (r / W4 + 1) * S2 + (r % W4) + 1] = p.st_a2[(long)n * 16 * P + q];
    }
    for (int q = tid; q < 4 * P; q += NT)
      reinterpret_cast<unsigned*>(i2b)[q] = reinterpret_cast<const unsigned*>(p.st_i2 + (long)n * 16 * P)[q];
    for (int q = tid; q < 8 * HW2; q += NT) {
      int c = q / HW2, r = q % HW2;
      a1h[c * P1 + (r / W2 + 1) * S1 + (r % W2) + 1] = p.st_a1[(long)n * 8 * HW2 + q];
    }
    __syncthreads();  // B

    // dy3 = mask3 * dfeat / P into its haloed planes; db3
    for (int q = tid; q < 6 * P; q += NT) {  // 24*P mask bytes as 6*P words
      const unsigned word = reinterpret_cast<const unsigned*>(p.st_m3 + (long)n * 24 * P)[q];
      const int c = (4 * q) / P, r = (4 * q) % P;
      const float dv = s_dfeat[c];
      float* dst = dy3h + c * P2 + (r / W4 + 1) * S2 + (r % W4) + 1;  // 4 | W4: the 4 pixels share a row
      int cnt = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const bool on = (word >> (8 * b)) & 1u;
        dst[b] = on ? dv : 0.f;
        cnt += on;
      }
      if (cnt) atomicAdd(&s_gb3[c], dv * (float)cnt);
    }
    __syncthreads();  // C

    // ---------------- S1: dW3
    {
      const int kpw = P / NWV;
      const int pbase = wvu * kpw;
      for (int kk = 0; kk < kpw / 4; ++kk) {
        const int p0 = pbase + 4 * kk;
        const int y = p0 / W4, x = p0 % W4 + g;
        const int hal = (y + 1) * S2 + x + 1;
        const float a0 = dy3h[i * P2 + hal];
        const float a1v = (i < 8) ? dy3h[(16 + i) * P2 + hal] : 0.f;
        const float* bp = a2h + i * P2 + y * S2 + x;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const float b = bp[(tap / 3) * S2 + (tap % 3)];
          acc3[0][tap] = mfma16(a0, b, acc3[0][tap]);
          acc3[1][tap] = mfma16(a1v, b, acc3[1][tap]);
        }
      }
    }
    // ---------------- S2: da2 (masked by a2 > 0) -> da2m ; db2
    {
      const int tiles = P / 16;
      for (int tile = wv; tile < tiles; tile += NWV) {
        const int pp = 16 * tile + i;
        const int y = pp / W4, x = pp % W4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* ap = dy3h + g * P2 + (y + 2) * S2 + (x + 2);
#pragma unroll
        for (int kk = 0; kk < 54; ++kk) {
          const int tap = kk / 6, nb = 4 * (kk % 6);
          const float a = ap[nb * P2 - (tap / 3) * S2 - (tap % 3)];
          const float b = w3s[(nb + g) * 144 + i * 9 + tap];
          acc = mfma16(a, b, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pq = 16 * tile + 4 * g + r;
          const float av = a2h[i * P2 + (pq / W4 + 1) * S2 + (pq % W4) + 1];
          const float v = av > 0.f ? acc[r] : 0.f;
          da2m[i * P + pq] = v;
          accb2 += v;
        }
      }
    }
    __syncthreads();  // D: dy3h / a2h are dead, da2m is complete

    // normalised image (no halo) and the pool-1 argmax bytes into the union region
    {
      const float mu = s_stat[0], sd = s_stat[1];
#pragma unroll
      for (int k = 0; k < MAXCH; ++k) {
        int q = tid + k * NT;
        if (q * 16 < HW) {
          const int lin = q * 16;
          float* dst = xs + (lin / G.W) * XSS + (lin % G.W);
          const unsigned wds[4] = {px[k].x, px[k].y, px[k].z, px[k].w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              float r = (float)((wds[e] >> (8 * b)) & 255u) / 255.0f;
              dst[4 * e + b] = p.standardize ? (r - mu) / sd : r;
            }
        }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if ((tid + k * NT) * 16 < 8 * HW2) reinterpret_cast<uint4*>(i1b)[tid + k * NT] = ix1[k];
    }
    if (n + (int)gridDim.x < p.N) load_frame(n + gridDim.x);

    // ---------------- S3: dW2
    {
      const int kpw = HW2 / NWV;
      const int pbase = wvu * kpw;
      for (int kk = 0; kk < kpw / 4; ++kk) {
        const int p0 = pbase + 4 * kk;
        const int y = p0 / W2, x = p0 % W2 + g;
        const int q = (y >> 1) * W4 + (x >> 1), o = (y & 1) * 2 + (x & 1);
        const float a = (i2b[i * P + q] == o) ? da2m[i * P + q] : 0.f;
        const float* bp = a1h + y * S1 + x;
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) {
          const float b = (boff[nt] >= 0) ? bp[boff[nt] < 0 ? 0 : boff[nt]] : 0.f;
          acc2[nt] = mfma16(a, b, acc2[nt]);
        }
      }
    }
    // ---------------- S4: da1 (masked by a1 > 0)
    {
      const int tiles = HW2 / 16;
      for (int tile = wv; tile < tiles; tile += NWV) {
        const int pp = 16 * tile + i;
        const int y = pp / W2, x = pp % W2;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int sy = y + 1 - tap / 3, sx = x + 1 - tap % 3;
          const bool inb = sy >= 0 && sy < G.H2 && sx >= 0 && sx < W2;
          const int q = inb ? (sy >> 1) * W4 + (sx >> 1) : 0;
          const int o = (sy & 1) * 2 + (sx & 1);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int nn = 4 * j + g;
            const float a = (inb && i2b[nn * P + q] == o) ? da2m[nn * P + q] : 0.f;
            const float b = (i < 8) ? w2s[nn * 72 + i * 9 + tap] : 0.f;
            acc = mfma16(a, b, acc);
          }
        }
        if (i < 8) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int pq = 16 * tile + 4 * g + r;
            const float av = a1h[i * P1 + (pq / W2 + 1) * S1 + (pq % W2) + 1];
            da1[i * HW2 + pq] = av > 0.f ? acc[r] : 0.f;
          }
        }
      }
    }
    __syncthreads();  // E

    // ---------------- S5: dW1, db1 (wave = channel)
    {
      const int c = wv;
      for (int q = lane; q < HW2; q += 64) {
        const float d = da1[c * HW2 + q];
        const int o = i1b[c * HW2 + q];
        const int y0 = 2 * (q / W2) + (o >> 1) - 1, x0 = 2 * (q % W2) + (o & 1) - 1;
        acc1[9] += d;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int yy = y0 + ky, xx = x0 + kx;
            const bool inb = yy >= 0 && yy < G.H && xx >= 0 && xx < G.W;
            const float xv = inb ? xs[(inb ? yy : 0) * XSS + (inb ? xx : 0)] : 0.f;
            acc1[ky * 3 + kx] += d * xv;
          }
      }
    }
    __syncthreads();  // F
  }

  // ---------------- flush: reduce the per-wave partials through LDS, then one atomic per element
  float* r_w3 = lds;            // [3456]
  float* r_w2 = r_w3 + 3456;    // [1152]
  float* r_w1 = r_w2 + 1152;    // [72]
  float* r_b1 = r_w1 + 72;      // [8]
  float* r_b2 = r_b1 + 8;       // [16]
  float* r_fc = r_b2 + 16;      // [E*24]
  const int rtot = 3456 + 1152 + 72 + 8 + 16 + E * 24;
  // (r_* overlay a1h / U, both dead; s_gb3 in misc is untouched)
  for (int q = tid; q < rtot; q += NT) lds[q] = 0.f;
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = 16 * mt + 4 * g + r;
        if (nn < 24) atomicAdd(&r_w3[nn * 144 + i * 9 + tap], acc3[mt][tap][r]);
      }
#pragma unroll
  for (int nt = 0; nt < 5; ++nt) {
    const int idx = 16 * nt + i;
    if (idx < 72) {
      const int tap = idx >> 3, c = idx & 7;
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&r_w2[(4 * g + r) * 72 + c * 9 + tap], acc2[nt][r]);
    }
  }
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    const float s = wave_sum(acc1[k]);
    if (lane == 0) {
      if (k < 9) r_w1[wv * 9 + k] = s;
      else r_b1[wv] = s;
    }
  }
  {
    float s = accb2;
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (g == 0) atomicAdd(&r_b2[i], s);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int idx = tid + k * NT;
    if (idx < E * 24) r_fc[idx] = accfc[k];
  }
  __syncthreads();
  for (int q = tid; q < 3456; q += NT) atomicAdd(&p.g_w3[q], r_w3[q]);
  for (int q = tid; q < 1152; q += NT) atomicAdd(&p.g_w2[q], r_w2[q]);
  if (tid < 72) atomicAdd(&p.g_w1[tid], r_w1[tid]);
  if (tid < 8) atomicAdd(&p.g_b1[tid], r_b1[tid]);
  if (tid < 16) atomicAdd(&p.g_b2[tid], r_b2[tid]);
  if (tid < 24) atomicAdd(&p.g_b3[tid], s_gb3[tid]);
  for (int q = tid; q < E * 24; q += NT) atomicAdd(&p.g_wfc[q], r_fc[q]);
  if (tid < E) atomicAdd(&p.g_bfc[tid], accbfc);
}

}  // namespace

extern "C" int ss_roi_cnn_bwd(const uint8_t* R, int N, int H, int W, int standardize, const float* w1,
                              const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                              const float* wfc, const float* bfc, int E, const float* st_a1, const uint8_t* st_i1,
                              const float* st_a2, const uint8_t* st_i2, const uint8_t* st_m3, const float* st_feat,
                              const float* d_out, int ld_dout, float* g_w1, float* g_b1, float* g_w2, float* g_b2,
                              float* g_w3, float* g_b3, float* g_wfc, float* g_bfc, ss_stream_t stream) {
  (void)w1; (void)b1; (void)b2; (void)b3; (void)bfc;  // the stashed activations already contain their effect
  SS_REQUIRE(R && w2 && w3 && wfc && st_a1 && st_i1 && st_a2 && st_i2 && st_m3 && st_feat && d_out, SS_ERR_ARG);
  SS_REQUIRE(g_w1 && g_b1 && g_w2 && g_b2 && g_w3 && g_b3 && g_wfc && g_bfc, SS_ERR_ARG);
  SS_REQUIRE(N > 0 && E > 0 && ld_dout >= E, SS_ERR_ARG);
  SS_REQUIRE(E <= 64 && H % 4 == 0 && W % 32 == 0 && H >= 4 && H * W <= MAXCH * 16 * NT, SS_ERR_UNSUPPORTED);
  CnnBwdParams p;
  p.R = R; p.N = N; p.standardize = standardize; p.w2 = w2; p.w3 = w3; p.wfc = wfc; p.E = E;
  p.st_a1 = st_a1; p.st_i1 = st_i1; p.st_a2 = st_a2; p.st_i2 = st_i2; p.st_m3 = st_m3; p.st_feat = st_feat;
  p.d_out = d_out; p.ld_dout = ld_dout;
  p.g_w1 = g_w1; p.g_b1 = g_b1; p.g_w2 = g_w2; p.g_b2 = g_b2; p.g_w3 = g_w3; p.g_b3 = g_b3; p.g_wfc = g_wfc; p.g_bfc = g_bfc;
  p.g = make_geom(H, W);
  p.L = make_bwd_layout(p.g);
  // the K = pixel splits need whole 4-pixel k-steps per wave, the argmax map one 16-byte piece per thread
  SS_REQUIRE(p.g.P % (4 * NWV) == 0 && (p.g.H2 * p.g.W2) % (4 * NWV) == 0 && 8 * p.g.H2 * p.g.W2 <= 2 * 16 * NT,
             SS_ERR_UNSUPPORTED);
  const size_t lds_bytes = (size_t)p.L.total * sizeof(float);
  SS_REQUIRE(lds_bytes <= 160 * 1024, SS_ERR_UNSUPPORTED);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(roi_cnn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return SS_ERR_LAUNCH;
    attr_set = true;
  }
  int grid = N < 256 ? N : 256;
  hipLaunchKernelGGL(roi_cnn_bwd_kernel, dim3(grid), dim3(NT), lds_bytes, static_cast<hipStream_t>(stream), p);
  return ss_launch_status();
}
