// Masked bidirectional GRU recurrence, forward and BPTT, with W_hh resident in registers.
//
// Replaces the packed nn.GRU call of /root/reference/train_model_official.py:301-305 (one layer,
// both directions per launch; the input projection W_ih x + b_ih for all timesteps is a dense
// GEMM done beforehand, ss_gemm_f32).  pack_padded_sequence semantics are reproduced by masking:
// the forward direction freezes h and emits zeros for t >= len, the reverse direction walks
// t = T-1 .. 0 and simply skips t >= len, so it starts from h = 0 at each clip's last valid frame.
//
// Work split: one workgroup per (16-clip slice, direction).  H/16 waves; wave w owns hidden units
// [16w, 16w+16) of all three gates.  v_mfma_f32_16x16x4_f32 with A = W_hh rows (one VGPR per 16x4
// fragment, 3*H/4 fragments per lane = the whole 3H x H matrix spread over the workgroup's
// register file, loaded once), B = h^T (k x clip) read from a 2-slot LDS ring, D rows = hidden
// unit, D cols = clip.  The three gate accumulators of one (unit, clip) land in the same lane, so
// sigmoid/tanh/blend run in registers and each step costs one workgroup barrier.
#include "ss_common.h"

namespace {

constexpr int SLICE = 16;  // clips per workgroup = MFMA N

template <int H>
struct GruCfg {
  static constexpr int NW = H / 16;        // waves
  static constexpr int KS = H / 4;         // k-steps of the h GEMM
  static constexpr int KS_B = 3 * H / 4;   // k-steps of the transposed (BPTT) GEMM
  // 12 waves (H=192) leave 168 VGPRs per lane: not enough for 144 weight fragments plus the working
  // set, so the tail of each fragment list lives in LDS (lane-linear, conflict-free ds_read_b32).
  static constexpr int KREG_F = (H > 128) ? 34 : KS;          // per gate, forward
  static constexpr int KLDS_F = KS - KREG_F;
  static constexpr int KREG_B = (H > 128) ? 108 : KS_B;       // backward
  static constexpr int KLDS_B = KS_B - KREG_B;
};

struct GruFwdParams {
  const float* gi;      // [2][B*T][3H]   W_ih x + b_ih, gate order r|z|n
  const float* w_hh[2]; // [3H][H]
  const float* b_hh[2]; // [3H]
  const int* lengths;   // [B]
  float* out;           // [B][T][2H]   forward dir in cols [0,H), reverse in [H,2H)
  float* save;          // [2][B*T][4][H]  r, z, n, q=(W_hn h + b_hn); may be null (inference)
  int B, T;
  // nn.GRU's inter-layer dropout as a by-product (multi-CU form only): out_drop = out * Philox keep-scale of ss_dropout's stream
  // (seed, offset, element index = position in `out`), or null
  float* out_drop;
  float drop_p;
  uint64_t drop_seed, drop_off;
};

template <int H>
__global__ __launch_bounds__(H * 4) void gru_fwd_kernel(GruFwdParams p) {
  using C = GruCfg<H>;
  __shared__ __attribute__((aligned(16))) float lds[2 * H * SLICE + 3 * H + 3 * C::KLDS_F * C::NW * 64];
  float* hbuf = lds;                   // [2][H][16]
  float* bias = lds + 2 * H * SLICE;   // [3H]
  float* wlds = bias + 3 * H;          // [3*KLDS_F][NW][64]

  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * SLICE;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int j0 = 16 * w + 4 * g;  // first of this lane's 4 hidden units
  const int clip = b0 + i;
  const bool clip_ok = clip < p.B;
  const int len = clip_ok ? p.lengths[clip] : 0;
  const int T = p.T;

  // W_hh fragments.  MFMA slot (kk, g) carries k = 16*(kk/4) + 4g + kk%4 (any bijection works as long as the B
  // operand uses the same one): a lane's four consecutive slots are then four CONSECUTIVE floats of its weight
  // row, so the 442 KB matrix streams in as 16-byte loads in full 64-byte sectors (the natural k = 4kk+g mapping
  // needs 4x the load instructions at 25 % sector efficiency, 40 us per launch).
  float wf[3][C::KREG_F];
  {
    const float* W = p.w_hh[dir];
#pragma unroll
    for (int G = 0; G < 3; ++G)
#pragma unroll
      for (int kq = 0; kq < C::KS / 4; ++kq) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(W + (long)(G * H + 16 * w + i) * H + 16 * kq + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kk = 4 * kq + e;
          if (kk < C::KREG_F) wf[G][kk] = v[e];
          else wlds[((G * C::KLDS_F + kk - C::KREG_F) * C::NW + w) * 64 + lane] = v[e];
        }
      }
  }
  for (int q = threadIdx.x; q < 3 * H; q += blockDim.x) bias[q] = p.b_hh[dir][q];
  for (int q = threadIdx.x; q < 2 * H * SLICE; q += blockDim.x) hbuf[q] = 0.f;
  __syncthreads();

  f32x4 hp = {0.f, 0.f, 0.f, 0.f};
  const long dir_off = (long)dir * p.B * T;
  int cur = 0;
  for (int s = 0; s < T; ++s) {
    const int t = dir ? (T - 1 - s) : s;
    const long frame = (long)clip * T + t;
    // input-projection gates for this (clip, t): issued early, consumed after the MFMA chain
    f32x4 gr = {0.f, 0.f, 0.f, 0.f}, gz = gr, gn = gr;
    const bool valid = t < len;
    if (valid) {
      const float* gp = p.gi + (dir_off + frame) * (3 * H) + j0;
      gr = *reinterpret_cast<const f32x4*>(gp);
      gz = *reinterpret_cast<const f32x4*>(gp + H);
      gn = *reinterpret_cast<const f32x4*>(gp + 2 * H);
    }
    f32x4 ar = *reinterpret_cast<const f32x4*>(&bias[j0]);
    f32x4 az = *reinterpret_cast<const f32x4*>(&bias[H + j0]);
    f32x4 an = *reinterpret_cast<const f32x4*>(&bias[2 * H + j0]);
    const float* hb = hbuf + cur * H * SLICE;
    // 4 k-steps per batch: their h values (and, past KREG_F, their LDS-resident weight fragments) are read
    // together, then the 12 MFMAs run while the other two waves of the SIMD cover the LDS latency
#pragma unroll
    for (int kb = 0; kb < C::KS; kb += 4) {
      float b[4], w0[4], w1[4], w2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kk = kb + u;
        b[u] = hb[(16 * (kk >> 2) + 4 * g + (kk & 3)) * SLICE + i];
        if (kk < C::KREG_F) {
          w0[u] = wf[0][kk]; w1[u] = wf[1][kk]; w2[u] = wf[2][kk];
        } else {
          const float* wl = wlds + ((kk - C::KREG_F) * C::NW + w) * 64 + lane;
          w0[u] = wl[0];
          w1[u] = wl[C::KLDS_F * C::NW * 64];
          w2[u] = wl[2 * C::KLDS_F * C::NW * 64];
        }
      }
      SS_SCHED_FENCE();
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ar = mfma16(w0[u], b[u], ar);
        az = mfma16(w1[u], b[u], az);
        an = mfma16(w2[u], b[u], an);
      }
      SS_SCHED_FENCE();
    }
    f32x4 r, z, n, hn;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      r[e] = sigmoid_f(gr[e] + ar[e]);
      z[e] = sigmoid_f(gz[e] + az[e]);
      n[e] = tanh_f(gn[e] + r[e] * an[e]);
      hn[e] = (1.0f - z[e]) * n[e] + z[e] * hp[e];
    }
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
      hp = hn;
      o = hn;
    }
    if (clip_ok) {
      *reinterpret_cast<f32x4*>(p.out + frame * (2 * H) + dir * H + j0) = o;
      if (p.save && valid) {
        float* sp = p.save + (dir_off + frame) * (4 * H) + j0;
        *reinterpret_cast<f32x4*>(sp) = r;
        *reinterpret_cast<f32x4*>(sp + H) = z;
        *reinterpret_cast<f32x4*>(sp + 2 * H) = n;
        *reinterpret_cast<f32x4*>(sp + 3 * H) = an;
      }
    }
    float* hnx = hbuf + (cur ^ 1) * H * SLICE;
#pragma unroll
    for (int e = 0; e < 4; ++e) hnx[(j0 + e) * SLICE + i] = hp[e];
    __syncthreads();
    cur ^= 1;
  }
}

struct GruBwdParams {
  const float* d_out;   // [B][T][2H]  gradient w.r.t. this layer's output
  const float* out;     // [B][T][2H]  forward output (h_{t-1} comes from here)
  const float* save;    // [2][B*T][4][H]
  const float* w_hh[2]; // [3H][H]
  const int* lengths;
  float* d_g;           // [2][B*T][4][H]: d a_r, d a_z, d a_n (= d gi), and d a_n * r (hh side of n)
  int B, T;
  // d_out is the gradient w.r.t. the DROPPED-OUT output of this layer when drop_p > 0: the mask of the forward
  // (ss_dropout stream (seed, offset) over the (B*T, 2H) tensor) is re-drawn while d_out is read
  float drop_p;
  uint64_t drop_seed, drop_off;
  // bias gradients (each [3H], accumulated; all NULL = not wanted): d b_ih += sum over (clip, t) of (d a_r, d a_z, d a_n),
  // d b_hh += the same with d a_n * r in the n block.  They ride on the BPTT: a pass over d_g afterwards costs as much HBM
  // traffic as the weight-gradient GEMM that follows it.
  float* g_bih[2];
  float* g_bhh[2];
};

// Per-lane running sums of the four gate gradients of hidden units j0..j0+3 (lane row = 16 clips); flush() adds the rows up on
// the DPP crossbar and lets one lane per row add them to the gradient vectors.  Every lane of the wave must call flush().
struct GruBiasAcc {
  f32x4 r = {0.f, 0.f, 0.f, 0.f}, z = r, n = r, q = r;
  __device__ __forceinline__ void add(const f32x4& dar, const f32x4& daz, const f32x4& dan, const f32x4& dqn) {
    r += dar; z += daz; n += dan; q += dqn;
  }
  __device__ __forceinline__ void flush(const GruBwdParams& p, int dir, int H, int j0, int i) {
    if (!p.g_bih[dir]) return;
    float* gi = p.g_bih[dir];
    float* gh = p.g_bhh[dir];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float sr = row_sum(r[e]), sz = row_sum(z[e]), sn = row_sum(n[e]), sq = row_sum(q[e]);
      if (i == 0) {
        atomicAdd(&gi[j0 + e], sr);
        atomicAdd(&gh[j0 + e], sr);
        atomicAdd(&gi[H + j0 + e], sz);
        atomicAdd(&gh[H + j0 + e], sz);
        atomicAdd(&gi[2 * H + j0 + e], sn);
        atomicAdd(&gh[2 * H + j0 + e], sq);
      }
    }
  }
};

template <int H>
__global__ __launch_bounds__(H * 4) void gru_bwd_kernel(GruBwdParams p) {
  using C = GruCfg<H>;
  __shared__ __attribute__((aligned(16))) float lds[3 * H * SLICE + C::KLDS_B * C::NW * 64];
  float* db = lds;                      // [3H][16]
  float* wlds = lds + 3 * H * SLICE;    // [KLDS_B][NW][64]

  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * SLICE;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int j0 = 16 * w + 4 * g;
  const int clip = b0 + i;
  const bool clip_ok = clip < p.B;
  const int len = clip_ok ? p.lengths[clip] : 0;
  const int T = p.T;

  // transposed fragments: dh_prev[k] = sum_row W[row][k] * dgh[row];  A[i][slot] = W[4ks+g][16w+i]
  float wf[C::KREG_B];
  {
    const float* W = p.w_hh[dir];
#pragma unroll
    for (int ks = 0; ks < C::KS_B; ++ks) {
      float v = W[(long)(4 * ks + g) * H + 16 * w + i];
      if (ks < C::KREG_B) wf[ks] = v;
      else wlds[((ks - C::KREG_B) * C::NW + w) * 64 + lane] = v;
    }
  }

  f32x4 dh = {0.f, 0.f, 0.f, 0.f};  // gradient flowing into h_t from later steps of the recurrence
  GruBiasAcc bacc;
  const long dir_off = (long)dir * p.B * T;
  for (int s = 0; s < T; ++s) {
    // reverse of the forward iteration order
    const int t = dir ? s : (T - 1 - s);
    const int tp = dir ? t + 1 : t - 1;  // where h_prev was emitted
    const long frame = (long)clip * T + t;
    const bool valid = t < len;
    f32x4 dar = {0.f, 0.f, 0.f, 0.f}, daz = dar, dan = dar, dqn = dar, dcarry = dh;
    if (valid) {
      f32x4 go = *reinterpret_cast<const f32x4*>(p.d_out + frame * (2 * H) + dir * H + j0);
      if (p.drop_p > 0.f) go *= drop_scale4((frame * (2 * H) + dir * H + j0) >> 2, p.drop_p, p.drop_seed, p.drop_off);
      const float* sp = p.save + (dir_off + frame) * (4 * H) + j0;
      f32x4 r = *reinterpret_cast<const f32x4*>(sp);
      f32x4 z = *reinterpret_cast<const f32x4*>(sp + H);
      f32x4 n = *reinterpret_cast<const f32x4*>(sp + 2 * H);
      f32x4 q = *reinterpret_cast<const f32x4*>(sp + 3 * H);
      f32x4 hprev = {0.f, 0.f, 0.f, 0.f};
      if (tp >= 0 && tp < len) hprev = *reinterpret_cast<const f32x4*>(p.out + ((long)clip * T + tp) * (2 * H) + dir * H + j0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float d = go[e] + dh[e];
        float dn = d * (1.0f - z[e]);
        float dz = d * (hprev[e] - n[e]);
        dan[e] = dn * (1.0f - n[e] * n[e]);
        dar[e] = dan[e] * q[e] * r[e] * (1.0f - r[e]);
        daz[e] = dz * z[e] * (1.0f - z[e]);
        dqn[e] = dan[e] * r[e];
        dcarry[e] = d * z[e];
      }
    }
    bacc.add(dar, daz, dan, dqn);
    if (clip_ok) {
      float* gp = p.d_g + (dir_off + frame) * (4 * H) + j0;
      *reinterpret_cast<f32x4*>(gp) = dar;
      *reinterpret_cast<f32x4*>(gp + H) = daz;
      *reinterpret_cast<f32x4*>(gp + 2 * H) = dan;
      *reinterpret_cast<f32x4*>(gp + 3 * H) = dqn;
    }
    __syncthreads();  // previous step's MFMA reads of db are done (also orders the wlds fill)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      db[(j0 + e) * SLICE + i] = dar[e];
      db[(H + j0 + e) * SLICE + i] = daz[e];
      db[(2 * H + j0 + e) * SLICE + i] = dqn[e];
    }
    __syncthreads();
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
#pragma unroll
    for (int kb = 0; kb < C::KS_B; kb += 6) {  // 6 k-steps per batch: reads first, then the MFMAs
      float a[6], b[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int ks = kb + u;
        a[u] = (ks < C::KREG_B) ? wf[ks < C::KREG_B ? ks : 0] : wlds[((ks - C::KREG_B) * C::NW + w) * 64 + lane];
        b[u] = db[(4 * ks + g) * SLICE + i];
      }
      SS_SCHED_FENCE();
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        if (u % 3 == 0) a0 = mfma16(a[u], b[u], a0);
        else if (u % 3 == 1) a1 = mfma16(a[u], b[u], a1);
        else a2 = mfma16(a[u], b[u], a2);
      }
      SS_SCHED_FENCE();
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) dh[e] = dcarry[e] + (a0[e] + a1[e]) + a2[e];
  }
  bacc.flush(p, dir, H, j0, i);
}

}  // namespace

STAMP_TABLE(ss_debug_stamps_gru)

extern int ss_cnn_max_wgs;  // roi_cnn.hip: the CU cap of the persistent CNN kernels (0 = none)
#include "gru_split.h"

extern "C" int ss_gru_sync_bytes(int B, int T, int H, long* bytes) {
  SS_REQUIRE(bytes && B > 0 && T > 0 && H > 0, SS_ERR_ARG);
  const int P = gru_split_small_parts(B, gru_split_parts(B, T, H));
  *bytes = P ? SYNC_HDR_WORDS * 4L + (gru_xid_granules(B) + gru_fwd_granules(B, H) + gru_bwd_granules(B, H, P)) * 8 : 0;
  return SS_OK;
}

// out_drop != NULL: the dropped-out copy of `out` the next layer's input projection multiplies (nn.GRU's inter-layer dropout, the
// stream of ss_dropout at (seed, offset)) is written by the recurrence kernel itself, behind each step's publish -- one launch and
// one read of `out` less per training step.  Needs the multi-CU form (sync_ws given and the batch small enough for it:
// ss_gru_sync_bytes() != 0); SS_ERR_UNSUPPORTED otherwise -- the caller then runs ss_dropout.
extern "C" int ss_gru_fwd_drop(const float* gi, const float* w_hh_f, const float* w_hh_r, const float* b_hh_f,
                               const float* b_hh_r, const int32_t* lengths, int B, int T, int H, float* out, float* save,
                               float* out_drop, float drop_p, uint64_t drop_seed, uint64_t drop_offset, void* sync_ws,
                               ss_stream_t stream) {
  SS_REQUIRE(gi && w_hh_f && w_hh_r && b_hh_f && b_hh_r && lengths && out, SS_ERR_ARG);
  SS_REQUIRE(B > 0 && T > 0 && drop_p >= 0.f && drop_p < 1.f, SS_ERR_ARG);
  GruFwdParams p;
  p.gi = gi; p.w_hh[0] = w_hh_f; p.w_hh[1] = w_hh_r; p.b_hh[0] = b_hh_f; p.b_hh[1] = b_hh_r;
  p.lengths = lengths; p.out = out; p.save = save; p.B = B; p.T = T;
  p.out_drop = out_drop; p.drop_p = drop_p; p.drop_seed = drop_seed; p.drop_off = drop_offset;
  dim3 grid(ceil_div(B, SLICE), 2);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int P = sync_ws ? gru_split_parts(B, T, H) : 0;
  SS_REQUIRE(P || !out_drop, SS_ERR_UNSUPPORTED);
  if (P) {
    // (while a CU cap is set on the persistent CNN kernels -- Trainer(micro_batches > 1) -- the slices keep six parts: the CUs the cap
    // reserves hold 96 recurrence workgroups, what two 8-slice micro-batches of six parts need)
    const int PF = ss_cnn_max_wgs ? P : gru_split_small_parts(B, P);
    dim3 sgrid(gru_split_grid_pairs(B, PF) * PF);
    unsigned* sy = static_cast<unsigned*>(sync_ws);
    u64* xid = reinterpret_cast<u64*>(sy + SYNC_HDR_WORDS);
    u64* hx = xid + gru_xid_granules(B);
    if (H == 192 && PF == 12) hipLaunchKernelGGL((gru_split_fwd_kernel<192, 12>), sgrid, dim3(256), 0, s, p, sy, xid, hx);
    else if (H == 192) hipLaunchKernelGGL((gru_split_fwd_kernel<192, 6>), sgrid, dim3(256), 0, s, p, sy, xid, hx);
    else hipLaunchKernelGGL((gru_split_fwd_kernel<64, 2>), sgrid, dim3(256), 0, s, p, sy, xid, hx);
    return ss_launch_status();
  }
  if (H == 192) hipLaunchKernelGGL(gru_fwd_kernel<192>, grid, dim3(768), 0, s, p);
  else if (H == 64) hipLaunchKernelGGL(gru_fwd_kernel<64>, grid, dim3(256), 0, s, p);
  else return SS_ERR_UNSUPPORTED;
  return ss_launch_status();
}

extern "C" int ss_gru_fwd(const float* gi, const float* w_hh_f, const float* w_hh_r, const float* b_hh_f,
                          const float* b_hh_r, const int32_t* lengths, int B, int T, int H, float* out, float* save,
                          void* sync_ws, ss_stream_t stream) {
  return ss_gru_fwd_drop(gi, w_hh_f, w_hh_r, b_hh_f, b_hh_r, lengths, B, T, H, out, save, nullptr, 0.f, 0, 0, sync_ws, stream);
}

extern "C" int ss_gru_bwd(const float* d_out, const float* out, const float* save, const float* w_hh_f,
                          const float* w_hh_r, const int32_t* lengths, int B, int T, int H, float* d_g,
                          float drop_p, uint64_t drop_seed, uint64_t drop_offset, float* g_bih_f, float* g_bhh_f,
                          float* g_bih_r, float* g_bhh_r, void* sync_ws, ss_stream_t stream) {
  SS_REQUIRE(d_out && out && save && w_hh_f && w_hh_r && lengths && d_g, SS_ERR_ARG);
  SS_REQUIRE(B > 0 && T > 0 && drop_p >= 0.f && drop_p < 1.f, SS_ERR_ARG);
  const bool any_b = g_bih_f || g_bhh_f || g_bih_r || g_bhh_r, all_b = g_bih_f && g_bhh_f && g_bih_r && g_bhh_r;
  SS_REQUIRE(!any_b || all_b, SS_ERR_ARG);
  GruBwdParams p;
  p.g_bih[0] = g_bih_f; p.g_bih[1] = g_bih_r; p.g_bhh[0] = g_bhh_f; p.g_bhh[1] = g_bhh_r;
  p.d_out = d_out; p.out = out; p.save = save; p.w_hh[0] = w_hh_f; p.w_hh[1] = w_hh_r;
  p.lengths = lengths; p.d_g = d_g; p.B = B; p.T = T;
  p.drop_p = drop_p; p.drop_seed = drop_seed; p.drop_off = drop_offset;
  dim3 grid(ceil_div(B, SLICE), 2);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int P = sync_ws ? gru_split_parts(B, T, H) : 0;
  if (P) {
    const int PB = ss_cnn_max_wgs ? P : gru_split_small_parts(B, P);
    dim3 sgrid(gru_split_grid_pairs(B, PB) * PB);
    unsigned* sy = static_cast<unsigned*>(sync_ws);
    u64* xid = reinterpret_cast<u64*>(sy + SYNC_HDR_WORDS);
    u64* xg = xid + gru_xid_granules(B) + gru_fwd_granules(B, H);
    if (H == 192 && PB == 12) hipLaunchKernelGGL((gru_split_bwd_kernel<192, 12>), sgrid, dim3(256), 0, s, p, sy, xid, xg);
    else if (H == 192) hipLaunchKernelGGL((gru_split_bwd_kernel<192, 6>), sgrid, dim3(256), 0, s, p, sy, xid, xg);
    else hipLaunchKernelGGL((gru_split_bwd_kernel<64, 2>), sgrid, dim3(256), 0, s, p, sy, xid, xg);
    return ss_launch_status();
  }
  if (H == 192) hipLaunchKernelGGL(gru_bwd_kernel<192>, grid, dim3(768), 0, s, p);
  else if (H == 64) hipLaunchKernelGGL(gru_bwd_kernel<64>, grid, dim3(256), 0, s, p);
  else return SS_ERR_UNSUPPORTED;
  return ss_launch_status();
}
