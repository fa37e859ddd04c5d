// Masked bidirectional GRU recurrence with bf16 MFMA for wide hidden states (BASELINE config 5: H = 512), forward and BPTT.
//
// Replaces pack_padded_sequence -> nn.GRU -> pad_packed_sequence (/root/reference/train_model_official.py:261-267,
// 301-305) for hidden sizes whose W_hh (3H x H = 1.5 MB in bf16 at H = 512) no longer fits the registers of a handful
// of CUs -- the f32 kernels of gru.hip / gru_split.h keep W_hh resident for H <= 192.  Here ONE launch is ONE time step of
// BOTH directions: 2 * (H/16) * ceil(B/64) workgroups (256 at B = 256, H = 512: every CU), each producing 16 hidden units
// of 64 clips.  W_hh streams from L2 as bf16 (3 MB for both directions, resident in every XCD's 4 MB L2 across the
// steps), the previous state comes as the bf16 copy the previous launch left, the four waves of a workgroup split the
// contraction (K = H forward, K = 3H backward) and reduce through LDS, and the gate arithmetic stays f32 on the
// accumulators.  A step is a kernel boundary (~1.5 us) + one L2 round trip + 48 (forward) / 48 (backward) MFMAs per wave:
// the same price as a step of the multi-CU persistent form, with no inter-workgroup protocol to keep alive.
//
// Packed-sequence semantics by masking, exactly as gru.hip: forward direction t = s, reverse direction t = T-1-s at step
// s; a step with t >= len[b] leaves the state (and emits zeros); the reverse direction therefore starts at len-1.
#include "bf16_common.h"
namespace {
STAMP_TABLE(ss_debug_stamps_gru_bf16)
}
#include "gru_bf16_pers.h"

namespace {

// clips per workgroup = CT MFMA column tiles: 64 clips -> 2 * (H/16) * ceil(B/64) workgroups = 256 at B = 256, H = 512, one per
// CU.  (CT = 2, two workgroups per CU with half the accumulators each, was measured: 5 % slower per step -- W_hh is read twice
// as often and a step is one memory round trip either way.)
constexpr int CT = 4, CG = 16 * CT;

// ---- once per optimiser step and layer: W_hh (f32, [3H][H]) of both directions -> bf16 copy and bf16 transpose [H][3H]
// (blockIdx.z = 0, 1), and W_ih (f32, [3H][K]) of both directions -> bf16 [3H][Kp], Kp = K rounded up to 8, zero-padded
// (blockIdx.z = 2, 3): the MFMA operands of the recurrence and of the layer's three GEMM families
__global__ __launch_bounds__(256) void whh_prep_kernel(const float* __restrict__ w_f, const float* __restrict__ w_r, int H,
                                                       bf16_t* __restrict__ wb, bf16_t* __restrict__ wtb,
                                                       const float* __restrict__ wi_f, const float* __restrict__ wi_r, int K, int Kp,
                                                       bf16_t* __restrict__ wib) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;  // rows of W (3H), columns
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  if (blockIdx.z >= 2) {
    if (!wib || c0 >= Kp) return;
    const int dir = blockIdx.z - 2;
    const float* w = dir ? wi_r : wi_f;
    for (int i = ty; i < 32; i += 8)
      if (c0 + tx < Kp) wib[(long)dir * 3 * H * Kp + (long)(r0 + i) * Kp + c0 + tx] = to_bf16(c0 + tx < K ? w[(long)(r0 + i) * K + c0 + tx] : 0.f);
    return;
  }
  if (c0 >= H) return;
  const int dir = blockIdx.z;
  const float* w = dir ? w_r : w_f;
  for (int i = ty; i < 32; i += 8) {
    const float v = w[(long)(r0 + i) * H + c0 + tx];
    tile[i][tx] = v;
    wb[(long)dir * 3 * H * H + (long)(r0 + i) * H + c0 + tx] = to_bf16(v);
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    wtb[(long)dir * 3 * H * H + (long)(c0 + i) * 3 * H + r0 + tx] = to_bf16(tile[tx][i]);
}

struct StepFwdParams {
  const float* gi;        // (2, N, 3H) f32: W_ih x + b_ih
  const bf16_t* whh;      // (2, 3H, H) bf16
  const float *bhh_f, *bhh_r;
  const int* lengths;
  int B, T, H, s;
  float* out;             // (N, 2H) f32
  float* save;            // (2, N, 4, H) f32 or null
  const bf16_t* hb_in;    // (2, B, H) bf16 state after step s-1 (unused at s = 0)
  bf16_t* hb_out;         // (2, B, H)
};

template <int H>
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(StepFwdParams p) {
  __shared__ __attribute__((aligned(16))) float red[4 * CT * 3 * 64 * 4];  // [wave][clip tile][gate][lane][4]
  const int B = p.B, T = p.T;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, li = lane & 15;
  const int dir = blockIdx.y;
  constexpr int nut = H / 16;
  const int ut = blockIdx.x % nut, cg = blockIdx.x / nut;
  const int t = dir ? (T - 1 - p.s) : p.s;
  const long N = (long)B * T;
  constexpr int KS = H / 128;  // 32-deep k steps of this wave's quarter of the contraction

  // ---- everything the gate phase needs is requested first: one memory round trip under the MFMA loop instead of a second
  // one behind it (a step is latency, not bandwidth)
  const int bq = CG * cg + 16 * wv + li;        // the clip this lane finishes (clip tile wv; waves >= CT only multiply)
  const int u = 16 * ut + 4 * g;                // its 4 hidden units
  const bool live = wv < CT && bq < B;
  const long row = (long)(live ? bq : 0) * T + t;
  const int len = p.lengths[live ? bq : 0];
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 gir = z4, giz = z4, gin = z4, br = z4, bz = z4, bn = z4, hp = z4;
  if (live) {  // not ``valid``: the loads must not wait for the clip length to arrive (a row past the clip's end is just unused)
    const float* gi = p.gi + ((long)dir * N + row) * 3 * H;
    const float* bhh = dir ? p.bhh_r : p.bhh_f;
    gir = *reinterpret_cast<const f32x4*>(gi + u); giz = *reinterpret_cast<const f32x4*>(gi + H + u);
    gin = *reinterpret_cast<const f32x4*>(gi + 2 * H + u);
    br = *reinterpret_cast<const f32x4*>(bhh + u); bz = *reinterpret_cast<const f32x4*>(bhh + H + u);
    bn = *reinterpret_cast<const f32x4*>(bhh + 2 * H + u);
    const int tp = dir ? t + 1 : t - 1;
    if (p.s > 0 && tp >= 0 && tp < T) hp = *reinterpret_cast<const f32x4*>(p.out + ((long)bq * T + tp) * 2 * H + dir * H + u);
  }

  f32x4 acc[CT][3];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[c][q] = z4;
  if (p.s > 0) {
    const bf16_t* W = p.whh + (long)dir * 3 * H * H;
    const bf16_t* hb = p.hb_in + (long)dir * B * H;
    s16x8 fa[KS][3], fb[KS][CT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {  // all operand loads of the wave in flight together
      const int k0 = wv * (H / 4) + 32 * ks;
#pragma unroll
      for (int q = 0; q < 3; ++q) fa[ks][q] = *reinterpret_cast<const s16x8*>(W + (long)(q * H + 16 * ut + li) * H + k0 + 8 * g);
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int b = CG * cg + 16 * c + li;
        fb[ks][c] = b < B ? *reinterpret_cast<const s16x8*>(hb + (long)b * H + k0 + 8 * g) : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int q = 0; q < 3; ++q) acc[c][q] = mfma_bf16(fa[ks][q], fb[ks][c], acc[c][q]);
  }
  const bool valid = live && t < len;
  // K slices -> LDS; wave w then owns clip tile w
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<f32x4*>(red + (((wv * CT + c) * 3 + q) * 64 + lane) * 4) = acc[c][q];
  __syncthreads();
  if (!live) return;
  f32x4 gh[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    f32x4 s = z4;
#pragma unroll
    for (int w = 0; w < 4; ++w) s += *reinterpret_cast<const f32x4*>(red + (((w * CT + wv) * 3 + q) * 64 + lane) * 4);
    gh[q] = s;
  }
  f32x4 hn = z4;
  if (valid) {
    f32x4 rr, zz, nn, hh;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float r = sigmoid_f(gir[e] + gh[0][e] + br[e]);
      const float z = sigmoid_f(giz[e] + gh[1][e] + bz[e]);
      const float hpre = gh[2][e] + bn[e];
      const float n = tanh_f(gin[e] + r * hpre);
      rr[e] = r; zz[e] = z; nn[e] = n; hh[e] = hpre;
      hn[e] = (1.0f - z) * n + z * hp[e];
    }
    if (p.save) {
      float* sv = p.save + ((long)dir * N + row) * 4 * H;
      *reinterpret_cast<f32x4*>(sv + u) = rr;
      *reinterpret_cast<f32x4*>(sv + H + u) = zz;
      *reinterpret_cast<f32x4*>(sv + 2 * H + u) = nn;
      *reinterpret_cast<f32x4*>(sv + 3 * H + u) = hh;
    }
  }
  *reinterpret_cast<f32x4*>(p.out + row * 2 * H + dir * H + u) = hn;
  *reinterpret_cast<uint2*>(p.hb_out + ((long)dir * B + bq) * H + u) = pack_bf16x4(hn[0], hn[1], hn[2], hn[3]);
}

struct StepBwdParams {
  const float* d_out;     // (N, 2H) f32 gradient w.r.t. this layer's (dropped-out) output
  const float* out;       // (N, 2H) f32 this layer's output (h_prev)
  const float* save;      // (2, N, 4, H)
  const bf16_t* whht;     // (2, H, 3H) bf16 transposed W_hh
  const int* lengths;
  int B, T, H, s;
  float* dG;              // (2, N, 4, H) f32: d gi_r, d gi_z, d gi_n, d(W_hn h + b_hn)
  const bf16_t* dgh_in;   // (2, B, 3H) bf16: d gh_r | d gh_z | d gh_n of step s-1
  bf16_t* dgh_out;
  float* dhz;             // (2, B, H) f32: the part of d h carried on without passing W_hh (dh * z, or all of it past a clip's end)
  float drop_p;
  uint64_t seed, offset;
};

template <int H>
__global__ __launch_bounds__(256) void gru_step_bwd_kernel(StepBwdParams p) {
  __shared__ __attribute__((aligned(16))) float red[4 * CT * 64 * 4];  // [wave][clip tile][lane][4]
  const int B = p.B, T = p.T;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, li = lane & 15;
  const int dir = blockIdx.y;
  constexpr int nut = H / 16;
  const int ut = blockIdx.x % nut, cg = blockIdx.x / nut;
  // BPTT walks the forward order backwards: forward direction t = T-1-s, reverse direction t = s
  const int t = dir ? p.s : (T - 1 - p.s);
  const long N = (long)B * T;
  constexpr int KS = 3 * H / 128;  // 32-deep k steps of this wave's quarter of the contraction (K = 3H)
  constexpr int KB = 6;            // k steps whose operands are in flight together

  // ---- the element-wise phase's operands first (see the forward kernel)
  const int bq = CG * cg + 16 * wv + li;
  const int u = 16 * ut + 4 * g;
  const bool live = wv < CT && bq < B;
  const long row = (long)(live ? bq : 0) * T + t;
  const int len = p.lengths[live ? bq : 0];
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 dh = z4, r = z4, z = z4, n = z4, hpre = z4, hp = z4, dhz_in = z4;
  float* dhz = p.dhz + ((long)dir * B + (live ? bq : 0)) * H + u;
  if (live && p.s > 0) dhz_in = *reinterpret_cast<const f32x4*>(dhz);
  if (live) {  // unconditional of the clip length (see the forward kernel); rows past a clip's end hold zeros / are ignored
    dh = *reinterpret_cast<const f32x4*>(p.d_out + row * 2 * H + dir * H + u);
    const float* sv = p.save + ((long)dir * N + row) * 4 * H;
    r = *reinterpret_cast<const f32x4*>(sv + u); z = *reinterpret_cast<const f32x4*>(sv + H + u);
    n = *reinterpret_cast<const f32x4*>(sv + 2 * H + u); hpre = *reinterpret_cast<const f32x4*>(sv + 3 * H + u);
    const int tp = dir ? t + 1 : t - 1;  // the step before this one in forward order
    if (tp >= 0 && tp < T) hp = *reinterpret_cast<const f32x4*>(p.out + ((long)bq * T + tp) * 2 * H + dir * H + u);
  }

  f32x4 acc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) acc[c] = z4;
  if (p.s > 0) {
    const bf16_t* Wt = p.whht + (long)dir * 3 * H * H;
    const bf16_t* dg = p.dgh_in + (long)dir * B * 3 * H;
#pragma unroll
    for (int kb = 0; kb < KS; kb += KB) {
      s16x8 fa[KB], fb[KB][CT];
#pragma unroll
      for (int ks = 0; ks < KB; ++ks) {
        if (kb + ks >= KS) break;  // compile-time: KS need not be a multiple of KB
        const int k0 = wv * (3 * H / 4) + 32 * (kb + ks);
        fa[ks] = *reinterpret_cast<const s16x8*>(Wt + (long)(16 * ut + li) * 3 * H + k0 + 8 * g);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const int b = CG * cg + 16 * c + li;
          fb[ks][c] = b < B ? *reinterpret_cast<const s16x8*>(dg + (long)b * 3 * H + k0 + 8 * g) : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
      }
#pragma unroll
      for (int ks = 0; ks < KB; ++ks) {
        if (kb + ks >= KS) break;
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = mfma_bf16(fa[ks], fb[ks][c], acc[c]);
      }
    }
  }
  const bool valid = live && t < len;
#pragma unroll
  for (int c = 0; c < CT; ++c) *reinterpret_cast<f32x4*>(red + ((wv * CT + c) * 64 + lane) * 4) = acc[c];
  __syncthreads();
  if (!live) return;
  f32x4 carry = dhz_in;
#pragma unroll
  for (int w = 0; w < 4; ++w) carry += *reinterpret_cast<const f32x4*>(red + ((w * CT + wv) * 64 + lane) * 4);
  f32x4 drp = z4, dzp = z4, dnp = z4, dhn = z4, keep = carry;
  if (valid) {
    if (p.drop_p > 0.f) dh *= drop_scale4((row * 2 * H + dir * H + u) >> 2, p.drop_p, p.seed, p.offset);
    dh += carry;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float dn = dh[e] * (1.0f - z[e]);
      const float dz = dh[e] * (hp[e] - n[e]);
      keep[e] = dh[e] * z[e];
      dnp[e] = dn * (1.0f - n[e] * n[e]);
      dhn[e] = dnp[e] * r[e];
      drp[e] = dnp[e] * hpre[e] * r[e] * (1.0f - r[e]);
      dzp[e] = dz * z[e] * (1.0f - z[e]);
    }
  }
  float* dG = p.dG + ((long)dir * N + row) * 4 * H;
  *reinterpret_cast<f32x4*>(dG + u) = drp;
  *reinterpret_cast<f32x4*>(dG + H + u) = dzp;
  *reinterpret_cast<f32x4*>(dG + 2 * H + u) = dnp;
  *reinterpret_cast<f32x4*>(dG + 3 * H + u) = dhn;
  bf16_t* dgo = p.dgh_out + ((long)dir * B + bq) * 3 * H;
  *reinterpret_cast<uint2*>(dgo + u) = pack_bf16x4(drp[0], drp[1], drp[2], drp[3]);
  *reinterpret_cast<uint2*>(dgo + H + u) = pack_bf16x4(dzp[0], dzp[1], dzp[2], dzp[3]);
  *reinterpret_cast<uint2*>(dgo + 2 * H + u) = pack_bf16x4(dhn[0], dhn[1], dhn[2], dhn[3]);
  *reinterpret_cast<f32x4*>(dhz) = keep;
}

// H-templated launchers: the kernels want compile-time trip counts (every operand load of a wave issued up front)
template <int H>
void launch_fwd_steps(StepFwdParams p, bf16_t* hb, int T, hipStream_t stream) {
  const long slot = 2L * p.B * H;
  dim3 grid((H / 16) * ceil_div(p.B, CG), 2);
  for (int s = 0; s < T; ++s) {
    p.s = s;
    p.hb_in = hb + ((s + 1) & 1) * slot;
    p.hb_out = hb + (s & 1) * slot;
    hipLaunchKernelGGL(gru_step_fwd_kernel<H>, grid, dim3(256), 0, stream, p);
  }
}

template <int H>
void launch_bwd_steps(StepBwdParams p, bf16_t* base, int T, hipStream_t stream) {
  const long slot = 2L * p.B * 3 * H;
  dim3 grid((H / 16) * ceil_div(p.B, CG), 2);
  for (int s = 0; s < T; ++s) {
    p.s = s;
    p.dgh_in = base + ((s + 1) & 1) * slot;
    p.dgh_out = base + (s & 1) * slot;
    hipLaunchKernelGGL(gru_step_bwd_kernel<H>, grid, dim3(256), 0, stream, p);
  }
}

// f32 rows -> bf16 rows (optionally through a dropout mask of ss_dropout's Philox stream over the SOURCE index space), 4 elements
// per thread; destination columns [cols, ld_y) are zero-filled (the GEMM reads whole 8-element chunks)
__global__ __launch_bounds__(256) void cvt_bf16_rows_kernel(const float* __restrict__ x, int ld_x, bf16_t* __restrict__ y, int ld_y,
                                                            long rows, int cols, float drop_p, uint64_t seed, uint64_t offset) {
  const int c4 = ld_y >> 2;
  const long total = rows * c4;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
    const long r = q / c4;
    const int c = 4 * (int)(q - r * c4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c < cols) {
      v = *reinterpret_cast<const f32x4*>(x + r * ld_x + c);
      if (drop_p > 0.f) v *= drop_scale4((r * ld_x + c) >> 2, drop_p, seed, offset);
    }
    *reinterpret_cast<uint2*>(y + r * ld_y + c) = pack_bf16x4(v[0], v[1], v[2], v[3]);
  }
}

int device_cus() { return ss_device_cus(); }

void cvt_rows(const float* x, int ld_x, bf16_t* y, int ld_y, long rows, int cols, float drop_p, uint64_t seed, uint64_t offset,
              hipStream_t st) {
  const long total = rows * (ld_y / 4);
  long blocks = (total + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  hipLaunchKernelGGL(cvt_bf16_rows_kernel, dim3((int)blocks), dim3(256), 0, st, x, ld_x, y, ld_y, rows, cols, drop_p, seed, offset);
}

template <int H>
void launch_pers_fwd(PersFwdParams p, void* sync_ws, int chunk, hipStream_t st) {
  constexpr int P = H / PUNITS;
  unsigned* sy = static_cast<unsigned*>(sync_ws);
  const int gmax = 2 * ceil_div(chunk, PSLICE);
  u64* xid = reinterpret_cast<u64*>(sy + SYNC_HDR_WORDS);
  u64* hx = xid + pers_xid_granules(gmax, P);
  for (int c0 = 0; c0 < p.B; c0 += chunk) {
    p.c0 = c0;
    p.nc = p.B - c0 < chunk ? p.B - c0 : chunk;
    hipLaunchKernelGGL(gru_pers_fwd_kernel<H>, dim3(2 * ceil_div(p.nc, PSLICE) * P), dim3(256), 0, st, p, sy, xid, hx);
  }
}

template <int H>
void launch_pers_bwd(PersBwdParams p, void* sync_ws, int chunk, hipStream_t st) {
  constexpr int P = H / PUNITS;
  unsigned* sy = static_cast<unsigned*>(sync_ws);
  const int gmax = 2 * ceil_div(chunk, PSLICE);
  u64* xid = reinterpret_cast<u64*>(sy + SYNC_HDR_WORDS);
  u64* xg = xid + pers_xid_granules(gmax, P) + pers_fwd_granules(gmax, H);
  for (int c0 = 0; c0 < p.B; c0 += chunk) {
    p.c0 = c0;
    p.nc = p.B - c0 < chunk ? p.B - c0 : chunk;
    hipLaunchKernelGGL(gru_pers_bwd_kernel<H>, dim3(2 * ceil_div(p.nc, PSLICE) * P), dim3(256), 0, st, p, sy, xid, xg);
  }
}

}  // namespace

extern "C" int ss_gru_bf16_prep(const float* w_hh_f, const float* w_hh_r, int H, uint16_t* whh_bf16, uint16_t* whh_t_bf16,
                                const float* w_ih_f, const float* w_ih_r, int K, uint16_t* wih_bf16, ss_stream_t stream) {
  SS_REQUIRE(w_hh_f && w_hh_r && whh_bf16 && whh_t_bf16 && H > 0 && H % 32 == 0, SS_ERR_ARG);
  SS_REQUIRE(!wih_bf16 || (w_ih_f && w_ih_r && K > 0), SS_ERR_ARG);
  const int Kp = wih_bf16 ? (K + 7) / 8 * 8 : 0;
  const int gx = ceil_div(H > Kp ? H : Kp, 32);
  hipLaunchKernelGGL(whh_prep_kernel, dim3(gx, 3 * H / 32, wih_bf16 ? 4 : 2), dim3(256), 0, static_cast<hipStream_t>(stream), w_hh_f,
                     w_hh_r, H, whh_bf16, whh_t_bf16, w_ih_f, w_ih_r, K, Kp, wih_bf16);
  return ss_launch_status();
}

extern "C" int ss_gru_bf16_ws_bytes(int B, int H, long* bytes) {
  SS_REQUIRE(bytes && B > 0 && H > 0, SS_ERR_ARG);
  // two state slots (2,B,H) bf16, two gate-gradient slots (2,B,3H) bf16, one (2,B,H) f32 carry
  *bytes = 2L * 2 * B * H * 2 + 2L * 2 * B * 3 * H * 2 + 2L * B * H * 4;
  return SS_OK;
}

extern "C" int ss_gru_bf16_sync_bytes(int B, int T, int H, long* bytes) {
  SS_REQUIRE(bytes && B > 0 && T > 0 && H > 0, SS_ERR_ARG);
  *bytes = 0;
  if (!pers_supported(H) || T > 1022) return SS_OK;  // step tags are 10 bits
  const int chunk = pers_chunk_clips(B, H, device_cus());
  if (chunk <= 0) return SS_OK;
  const int gmax = 2 * ceil_div(chunk, PSLICE);
  *bytes = SYNC_HDR_WORDS * 4L + (pers_xid_granules(gmax, H / PUNITS) + pers_fwd_granules(gmax, H) + pers_bwd_granules(gmax, H)) * 8;
  return SS_OK;
}

extern "C" int ss_cvt_bf16_rows(const float* x, int ld_x, uint16_t* y, int ld_y, long rows, int cols, float drop_p, uint64_t seed,
                                uint64_t offset, ss_stream_t stream) {
  SS_REQUIRE(x && y && rows > 0 && cols > 0 && ld_x >= cols && ld_y >= cols, SS_ERR_ARG);
  SS_REQUIRE(cols % 4 == 0 && ld_x % 4 == 0 && ld_y % 4 == 0 && drop_p >= 0.f && drop_p < 1.f, SS_ERR_ARG);
  SS_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0, SS_ERR_ARG);
  cvt_rows(x, ld_x, y, ld_y, rows, cols, drop_p, seed, offset, static_cast<hipStream_t>(stream));
  return ss_launch_status();
}

// One layer, both directions.  With a sync workspace (ss_gru_bf16_sync_bytes > 0, zeroed once by the caller) ONE persistent
// launch per clip chunk; otherwise T step launches.
extern "C" int ss_gru_bf16_fwd(const float* gi, const uint16_t* whh_bf16, const float* b_hh_f, const float* b_hh_r,
                               const int32_t* lengths, int B, int T, int H, float* out, float* save, uint16_t* out_bf16,
                               uint16_t* out_drop_bf16, float drop_p, uint64_t seed, uint64_t offset, void* ws, void* sync_ws,
                               long sync_bytes, ss_stream_t stream) {
  SS_REQUIRE(gi && whh_bf16 && b_hh_f && b_hh_r && lengths && out && ws, SS_ERR_ARG);
  SS_REQUIRE(B > 0 && T > 0 && drop_p >= 0.f && drop_p < 1.f, SS_ERR_ARG);
  SS_REQUIRE(H >= 128 && H % 128 == 0, SS_ERR_UNSUPPORTED);  // (step kernels: four waves x whole 32-deep k steps; built: 128..512, 1024)
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int chunk = (sync_ws && pers_supported(H) && T <= 1022) ? pers_chunk_clips(B, H, device_cus()) : 0;
  if (chunk > 0) {
    long need = 0;
    if (ss_gru_bf16_sync_bytes(B, T, H, &need) != SS_OK) return SS_ERR_ARG;
    SS_REQUIRE(sync_bytes >= need, SS_ERR_ARG);  // the section offsets below are derived from B: an area sized for another batch may be short
    PersFwdParams q;
    q.gi = gi; q.whh = whh_bf16; q.bhh_f = b_hh_f; q.bhh_r = b_hh_r; q.lengths = lengths; q.B = B; q.T = T;
    q.out = out; q.save = save; q.out_bf = out_bf16; q.out_drop_bf = out_drop_bf16; q.drop_p = drop_p; q.seed = seed; q.offset = offset;
    switch (H) {
      case 128: launch_pers_fwd<128>(q, sync_ws, chunk, st); break;
      case 256: launch_pers_fwd<256>(q, sync_ws, chunk, st); break;
      case 384: launch_pers_fwd<384>(q, sync_ws, chunk, st); break;
      case 512: launch_pers_fwd<512>(q, sync_ws, chunk, st); break;
      default: return SS_ERR_UNSUPPORTED;
    }
    return ss_launch_status();
  }
  StepFwdParams p;
  p.gi = gi; p.whh = whh_bf16; p.bhh_f = b_hh_f; p.bhh_r = b_hh_r; p.lengths = lengths;
  p.B = B; p.T = T; p.H = H; p.out = out; p.save = save;
  bf16_t* hb = static_cast<bf16_t*>(ws);
  switch (H) {
    case 128: launch_fwd_steps<128>(p, hb, T, st); break;
    case 256: launch_fwd_steps<256>(p, hb, T, st); break;
    case 384: launch_fwd_steps<384>(p, hb, T, st); break;
    case 512: launch_fwd_steps<512>(p, hb, T, st); break;
    case 1024: launch_fwd_steps<1024>(p, hb, T, st); break;
    default: return SS_ERR_UNSUPPORTED;
  }
  const long N = (long)B * T;
  if (out_bf16) cvt_rows(out, 2 * H, out_bf16, 2 * H, N, 2 * H, 0.f, 0, 0, st);
  if (out_drop_bf16) cvt_rows(out, 2 * H, out_drop_bf16, 2 * H, N, 2 * H, drop_p, seed, offset, st);
  return ss_launch_status();
}

extern "C" int ss_gru_bf16_bwd(const float* d_out, const float* out, const float* save, const uint16_t* whh_t_bf16,
                               const int32_t* lengths, int B, int T, int H, float* d_g, uint16_t* d_g_bf16, float drop_p,
                               uint64_t seed, uint64_t offset, float* g_bih_f, float* g_bhh_f, float* g_bih_r, float* g_bhh_r,
                               void* ws, void* sync_ws, long sync_bytes, ss_stream_t stream) {
  SS_REQUIRE(d_out && out && save && whh_t_bf16 && lengths && (d_g || d_g_bf16) && ws, SS_ERR_ARG);
  SS_REQUIRE(B > 0 && T > 0 && drop_p >= 0.f && drop_p < 1.f, SS_ERR_ARG);
  SS_REQUIRE(H >= 128 && H % 128 == 0, SS_ERR_UNSUPPORTED);
  const bool want_bias = g_bih_f || g_bhh_f || g_bih_r || g_bhh_r;
  SS_REQUIRE(!want_bias || (g_bih_f && g_bhh_f && g_bih_r && g_bhh_r), SS_ERR_ARG);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int chunk = (sync_ws && pers_supported(H) && T <= 1022) ? pers_chunk_clips(B, H, device_cus()) : 0;
  if (chunk > 0) {
    long need = 0;
    if (ss_gru_bf16_sync_bytes(B, T, H, &need) != SS_OK) return SS_ERR_ARG;
    SS_REQUIRE(sync_bytes >= need, SS_ERR_ARG);  // the section offsets below are derived from B: an area sized for another batch may be short
    PersBwdParams q;
    q.d_out = d_out; q.out = out; q.save = save; q.whht = whh_t_bf16; q.lengths = lengths; q.B = B; q.T = T;
    q.dG = d_g; q.dG_bf = d_g_bf16; q.drop_p = drop_p; q.seed = seed; q.offset = offset;
    q.g_bih[0] = g_bih_f; q.g_bih[1] = g_bih_r; q.g_bhh[0] = g_bhh_f; q.g_bhh[1] = g_bhh_r;
    switch (H) {
      case 128: launch_pers_bwd<128>(q, sync_ws, chunk, st); break;
      case 256: launch_pers_bwd<256>(q, sync_ws, chunk, st); break;
      case 384: launch_pers_bwd<384>(q, sync_ws, chunk, st); break;
      case 512: launch_pers_bwd<512>(q, sync_ws, chunk, st); break;
      default: return SS_ERR_UNSUPPORTED;
    }
    return ss_launch_status();
  }
  SS_REQUIRE(d_g, SS_ERR_ARG);  // the step kernels hand the gate gradients on through d_g
  StepBwdParams p;
  p.d_out = d_out; p.out = out; p.save = save; p.whht = whh_t_bf16; p.lengths = lengths;
  p.B = B; p.T = T; p.H = H; p.dG = d_g; p.drop_p = drop_p; p.seed = seed; p.offset = offset;
  bf16_t* base = static_cast<bf16_t*>(ws) + 2L * 2 * B * H;  // behind the two forward state slots
  const long slot = 2L * B * 3 * H;
  p.dhz = reinterpret_cast<float*>(base + 2 * slot);
  switch (H) {
    case 128: launch_bwd_steps<128>(p, base, T, st); break;
    case 256: launch_bwd_steps<256>(p, base, T, st); break;
    case 384: launch_bwd_steps<384>(p, base, T, st); break;
    case 512: launch_bwd_steps<512>(p, base, T, st); break;
    case 1024: launch_bwd_steps<1024>(p, base, T, st); break;
    default: return SS_ERR_UNSUPPORTED;
  }
  const long N = (long)B * T;
  if (d_g_bf16) cvt_rows(d_g, 4 * H, d_g_bf16, 4 * H, 2 * N, 4 * H, 0.f, 0, 0, st);
  if (want_bias) {
    const int rc = ss_gru_bias_grad(d_g, (int)N, H, g_bih_f, g_bhh_f, g_bih_r, g_bhh_r, stream);
    if (rc != SS_OK) return rc;
  }
  return ss_launch_status();
}
