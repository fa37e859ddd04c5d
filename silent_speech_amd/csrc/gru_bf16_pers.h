// Persistent form of the bf16 BiGRU recurrence (BASELINE config 5, H = 512): one launch per layer instead of one per time step.
//
// gru_bf16.hip's step kernels pay a kernel boundary + a cold start (W_hh from L2, the state from HBM) per time step: 9 / 14 us per
// step at H = 512 although a step's matrix work is 0.3 us of the chip.  Here a (16-clip slice, direction) is spread over
// P = H/64 workgroups of 4 waves that stay resident for all T steps:
//   forward   part p keeps the W_hh rows of its 64 hidden units (3 gates x 64 rows x H bf16 = 192 registers per lane) as MFMA A
//             fragments, multiplies them by the FULL previous state of the slice (all-gather of 16 x H bf16 per step) and
//             publishes the 16 x 64 slice it produced; wave w owns unit tile w, its three gate accumulators of one (unit, clip)
//             land in one lane, so the gate arithmetic runs in registers and a step has ONE workgroup barrier;
//   backward  W_hh^T is split over its contraction index: the part multiplies by the pre-activation gradients of its OWN units
//             (no gather), gets a partial d h_prev for all H units and sends every part the 16 x 64 piece it owns
//             (reduce-scatter; the partials travel as bf16, see below); gate gradients, the bias-gradient sums and the bf16 copies the weight-gradient
//             GEMMs read are by-products of the step.
// Exchange: the tagged 8-byte granules of granule_xchg.h, {2 x bf16, tag} in both directions, same-XCD fast path
// when the group count is a multiple of 8, bounded sweeps, NaN poison of a workgroup's own outputs when a partner is lost.
// The partial d h_prev sums of the backward pass are rounded to bf16 for the trip (f32 {value, tag} granules double the bytes
// a workgroup stores and sweeps per step -- 64 KB each -- and the step took 7.9 us against 3.5 us forward): P partials of
// 192 products each, rounded to 2^-9, add an error of the size the bf16 rounding of the 1 536 gate-gradient operands of the
// same sum already has; the carry d h * z that does not pass W_hh stays f32 in registers.
// The launcher only uses this form when the whole grid is co-resident (2 * ceil(B/16) * P <= the CU count; larger batches
// run as several launches over clip chunks) and H is 128 ... 512 in steps of 64; anything else stays on the step kernels.
#pragma once
#include "granule_xchg.h"

namespace {

constexpr int PSLICE = 16;   // clips per group = MFMA N
constexpr int PUNITS = 64;   // hidden units per part = 4 waves x 16

// generic sweep: thread re-reads N 16-byte granule pairs (stride `stride` pairs) until every tag matches; payloads raw
template <int N>
__device__ __forceinline__ bool sweep_pairs(rsrc_t rs, int pair0, int stride, unsigned tag, unsigned (&v)[2 * N], unsigned* errors,
                                            int lane) {
  for (int spins = 0;;) {
    bool ok = true;
    asm volatile("" ::: "memory");  // every pass really loads again
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, (pair0 + stride * k) * 16, 0, AUX_SC1);
      v[2 * k] = x[0];
      v[2 * k + 1] = x[2];
      ok &= x[1] == tag && x[3] == tag;
    }
    if (__all(ok)) return true;
    if (++spins > (1 << 20)) {
      if (lane == 0) atomicAdd(errors, 1u);
      return false;
    }
  }
}

// A sweep in two halves: issue the first pass, do something else while its loads fly, then look at the tags (and fall back to the
// spinning form when a partner has not published yet)
template <int N>
__device__ __forceinline__ void sweep_issue(rsrc_t rs, int pair0, int stride, u32x4 (&raw)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) raw[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (pair0 + stride * k) * 16, 0, AUX_SC1);
}
template <int N>
__device__ __forceinline__ bool sweep_check(const u32x4 (&raw)[N], unsigned tag, unsigned (&v)[2 * N]) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    v[2 * k] = raw[k][0];
    v[2 * k + 1] = raw[k][2];
    ok &= raw[k][1] == tag && raw[k][3] == tag;
  }
  return __all(ok);
}

__device__ __forceinline__ void store_pair_raw(rsrc_t rs, int pair, unsigned tag, unsigned p0, unsigned p1, bool same_xcd) {
  const u32x4 d = {p0, tag, p1, tag};
  if (same_xcd) __builtin_amdgcn_raw_buffer_store_b128(d, rs, pair * 16, 0, 0);
  else __builtin_amdgcn_raw_buffer_store_b128(d, rs, pair * 16, 0, AUX_SC1);
}

struct PersFwdParams {
  const float* gi;        // (2, N, 3H) f32
  const bf16_t* whh;      // (2, 3H, H) bf16
  const float *bhh_f, *bhh_r;
  const int* lengths;
  int B, T, c0, nc;       // all clips / frames per clip / first clip and clip count of this launch
  float* out;             // (N, 2H) f32
  float* save;            // (2, N, 4, H) f32 or null
  bf16_t* out_bf;         // (N, 2H) bf16 copy of out, or null        (h_prev operand of the d W_hh GEMM)
  bf16_t* out_drop_bf;    // (N, 2H) bf16 of dropout(out), or null    (input of the next layer's GEMMs)
  float drop_p;
  uint64_t seed, offset;
};

// blockIdx -> (group = slice * 2 + direction, part); part-major when the group count is a multiple of 8 (partners share an XCD)
__device__ __forceinline__ void pers_ids(int P, int& groups, int& group, int& part) {
  groups = gridDim.x / P;
  const bool part_major = (groups & 7) == 0;
  part = part_major ? blockIdx.x / groups : blockIdx.x % P;
  group = part_major ? blockIdx.x % groups : blockIdx.x / P;
}

template <int H>
__global__ __launch_bounds__(256) void gru_pers_fwd_kernel(PersFwdParams p, unsigned* sync, u64* xid, u64* hx) {
  constexpr int P = H / PUNITS, KS = H / 32;
  constexpr int LDH = H + 16;                   // panel row stride (bf16) == 16 (mod 32): conflict-free under ds_read_b128's lane groups
  constexpr int NGP = H / 64;                   // granule pairs per thread in a sweep of the 16 x H panel (4 H / 256)
  // Two panels, by step parity.  A step has ONE workgroup barrier (between the panel writes and the fragment reads), and nothing
  // else orders a wave's panel writes of step s + 1 behind its siblings' fragment reads of step s: a wave whose sweep only covers
  // units that OTHER parts publish (H = 512: wave 1 of part 0 sweeps units 256..511) can pass the sweep of step s + 1 while a
  // sibling still reads step s's panel.  With two panels step s + 1 writes the other one, and step s + 2 -- which reuses this
  // one -- lies behind the barrier of step s + 1, which every sibling reaches only after its reads of step s.
  __shared__ __attribute__((aligned(16))) bf16_t hpan2[2][PSLICE * LDH];
  __shared__ unsigned s_gen;
  int groups, group, part;
  pers_ids(P, groups, group, part);
  const int dir = group & 1, b0 = p.c0 + (group >> 1) * PSLICE;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, g = lane >> 4;
  const int u0 = part * PUNITS + 16 * w + 4 * g;  // first of this lane's 4 hidden units (D rows 4g .. 4g+3 of unit tile w)
  const int clip = b0 + li;
  const bool clip_ok = clip < p.c0 + p.nc;
  const int len = clip_ok ? p.lengths[clip] : 0;
  const int T = p.T;
  if (tid == 0) s_gen = __hip_atomic_load(&sync[0], __ATOMIC_RELAXED, SS_AGENT);
  if (plays_dead(sync)) {  // wave-uniform; tests only
    __syncthreads();
    finish_launch(sync, s_gen);
    return;
  }
  __builtin_amdgcn_s_setprio(3);

  // A fragments: rows = this wave's 16 units of gate G, k = 32 ks + 8 g + j
  s16x8 fa[3][KS];
  {
    const bf16_t* W = p.whh + (long)dir * 3 * H * H;
#pragma unroll
    for (int G = 0; G < 3; ++G)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        fa[G][ks] = *reinterpret_cast<const s16x8*>(W + (long)(G * H + part * PUNITS + 16 * w + li) * H + 32 * ks + 8 * g);
  }
  const float* bhh = dir ? p.bhh_r : p.bhh_f;
  const f32x4 br = *reinterpret_cast<const f32x4*>(bhh + u0), bz = *reinterpret_cast<const f32x4*>(bhh + H + u0),
              bn = *reinterpret_cast<const f32x4*>(bhh + 2 * H + u0);
  __syncthreads();
  const unsigned gen = s_gen;
  const unsigned base = (gen & 0x3FFFFFu) << 10;
  bool dead = false;
  const bool same_xcd = partners_share_xcd(xid, group, part, P, base, &sync[2], lane, &dead);
  if (tid == 0 && same_xcd) atomicAdd(&sync[3], 1u);

  const long N = (long)p.B * T;
  const rsrc_t hrs = granule_rsrc(hx, 2L * groups * PSLICE * (H / 2));
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 hp = z4;
  // The stores of a step (out, its bf16 copies, the dropout draw, the saved gates) are not on the recurrence's critical path but
  // their issue was: 1 800 of a step's 9 000 cycles (stage timers).  They are kept in registers and issued in the NEXT step,
  // between the request of the sweep's first pass and the look at its tags.
  f32x4 st_o = z4, st_r = z4, st_z = z4, st_n = z4, st_q = z4;
  long st_row = 0;
  bool st_valid = false, st_have = false;
  auto flush_stores = [&]() {
    if (!st_have || !clip_ok) return;
    const f32x4 o = st_o;
    const uint2 ob = pack_bf16x4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<f32x4*>(p.out + st_row * (2 * H) + dir * H + u0) = o;
    if (p.out_bf) *reinterpret_cast<uint2*>(p.out_bf + st_row * (2 * H) + dir * H + u0) = ob;
    if (p.out_drop_bf) {
      f32x4 od = o;
      if (p.drop_p > 0.f) od *= drop_scale4((st_row * (2 * H) + dir * H + u0) >> 2, p.drop_p, p.seed, p.offset);
      *reinterpret_cast<uint2*>(p.out_drop_bf + st_row * (2 * H) + dir * H + u0) = pack_bf16x4(od[0], od[1], od[2], od[3]);
    }
    if (p.save && st_valid) {
      float* sv = p.save + ((long)dir * N + st_row) * (4 * H) + u0;
      *reinterpret_cast<f32x4*>(sv) = st_r;
      *reinterpret_cast<f32x4*>(sv + H) = st_z;
      *reinterpret_cast<f32x4*>(sv + 2 * H) = st_n;
      *reinterpret_cast<f32x4*>(sv + 3 * H) = st_q;
    }
  };
  f32x4 ngr, ngz, ngn;  // input-projection gates of the NEXT step
  auto load_gi = [&](int s) {
    const int t = dir ? (T - 1 - s) : s;
    const long rowl = (long)(clip_ok ? clip : p.c0) * T + (t < len ? t : 0);
    const float* gp = p.gi + ((long)dir * N + rowl) * (3 * H) + u0;
    ngr = *reinterpret_cast<const f32x4*>(gp);
    ngz = *reinterpret_cast<const f32x4*>(gp + H);
    ngn = *reinterpret_cast<const f32x4*>(gp + 2 * H);
  };
  load_gi(0);
  STAMP_ENTRY;
  STAMP_DECL;
  for (int s = 0; s < T; ++s) {
    STAMP(15);
    const int t = dir ? (T - 1 - s) : s;
    const long row = (long)clip * T + t;
    const bool valid = t < len;
    // In flight during the sweep -- and UNCONDITIONAL (a lane past its clip's end reads a row of a valid clip and never uses it):
    // behind `if (valid)` the registers were cleared first, and a write to the destination of an older load makes hipcc wait for
    // that load by COUNT -- it cannot count the younger stores issued under divergent branches, so it waited for (nearly) all of
    // them, write-through granule stores included.  In the BPTT kernel below that wait was 3.3 k of 11.6 k cycles per step.
    // ... and requested a step AHEAD, behind the previous step's sweep: loads return in order, so requested at the top of their own
    // step (the first version) the sweep's L2 hits queued behind these three HBM misses.
    const f32x4 gr = ngr, gz = ngz, gn = ngn;
    if (s == 0 && T > 1) load_gi(1);
    f32x4 ar = br, az = bz, an = bn;
    if (s > 0) {
      // the full previous state of the slice: every thread sweeps NGP granule pairs (4 units of one clip each) into the panel
      unsigned hv[2 * NGP];
      const int pr = (((s - 1) & 1) * groups + group) * (PSLICE * H / 4) + tid;
      u32x4 raw[NGP];
      if (!dead) sweep_issue<NGP>(hrs, pr, 256, raw);
      flush_stores();  // the previous step's, under the sweep's first pass
      if (!dead && !sweep_check<NGP>(raw, base + (unsigned)s, hv)) dead = !sweep_pairs<NGP>(hrs, pr, 256, base + (unsigned)s, hv, &sync[2], lane);
      STAMP(0);
      if (s + 1 < T) load_gi(s + 1);  // behind the sweep: a whole step to arrive
      bf16_t* hpan = hpan2[s & 1];
#pragma unroll
      for (int k = 0; k < NGP; ++k) {
        const int q = tid + 256 * k;  // clip = q / (H/4), units 4 (q % (H/4)) ..
        *reinterpret_cast<uint2*>(&hpan[(q / (H / 4)) * LDH + 4 * (q % (H / 4))]) = uint2{hv[2 * k], hv[2 * k + 1]};
      }
      __syncthreads();
      STAMP(1);
      s16x8 fb[KS];  // every B fragment of the step requested before the first MFMA (one LDS latency instead of four)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) fb[ks] = *reinterpret_cast<const s16x8*>(&hpan[li * LDH + 32 * ks + 8 * g]);
      SS_SCHED_FENCE();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        ar = mfma_bf16(fa[0][ks], fb[ks], ar);
        az = mfma_bf16(fa[1][ks], fb[ks], az);
        an = mfma_bf16(fa[2][ks], fb[ks], an);
      }
      SS_SCHED_FENCE();
    }
    STAMP(2);
    f32x4 r, z, n, o = z4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      r[e] = sigmoid_f(gr[e] + ar[e]);
      z[e] = sigmoid_f(gz[e] + az[e]);
      n[e] = tanh_f(gn[e] + r[e] * an[e]);
    }
    if (valid) {
#pragma unroll
      for (int e = 0; e < 4; ++e) hp[e] = (1.0f - z[e]) * n[e] + z[e] * hp[e];
      o = hp;
    }
    if (dead) o = f32x4{NAN_F, NAN_F, NAN_F, NAN_F};
    if (s + 1 < T) {  // publish (padding clips too: the partners sweep whole panels)
      const uint2 ob = pack_bf16x4(o[0], o[1], o[2], o[3]);
      const int pw = ((s & 1) * groups + group) * (PSLICE * H / 4) + li * (H / 4) + (u0 >> 2);
      store_pair_raw(hrs, pw, base + (unsigned)s + 1u, ob.x, ob.y, same_xcd);
    }
    STAMP(3);
    st_o = o; st_r = r; st_z = z; st_n = n; st_q = an; st_row = row; st_valid = valid; st_have = true;
    STAMP(4);
  }
  flush_stores();
  STAMP_FLUSH();
  finish_launch(sync, gen);
}

struct PersBwdParams {
  const float* d_out;     // (N, 2H) f32 gradient w.r.t. this layer's (dropped-out) output
  const float* out;       // (N, 2H) f32 this layer's output (h_prev)
  const float* save;      // (2, N, 4, H)
  const bf16_t* whht;     // (2, H, 3H) bf16 transposed W_hh
  const int* lengths;
  int B, T, c0, nc;
  float* dG;              // (2, N, 4, H) f32: d gi_r, d gi_z, d gi_n, d(W_hn h + b_hn), or null (126 MB per launch at config 5 that
                          // nobody reads when the GEMMs take the bf16 copy and the bias sums are made here)
  bf16_t* dG_bf;          // the same as bf16, or null (operand of the d layer_in / weight-gradient GEMMs)
  float drop_p;
  uint64_t seed, offset;
  float* g_bih[2];        // bias gradients (+=), or null: d b_ih = sum of (dr, dz, dn), d b_hh = sum of (dr, dz, dhn)
  float* g_bhh[2];
};

template <int H>
__global__ __launch_bounds__(256) void gru_pers_bwd_kernel(PersBwdParams p, unsigned* sync, u64* xid, u64* xg) {
  constexpr int P = H / PUNITS;
  constexpr int KR = 3 * PUNITS, KSB = KR / 32;  // contraction over this part's r | z | n rows: 192 = 6 k-steps
  constexpr int MT = H / 64;                     // output unit tiles per wave (H/16 tiles over 4 waves)
  constexpr int LDP = KR + 8;                    // d_pre panel row stride (bf16)
  constexpr int LDS_ = PUNITS + 4;               // summed d h_prev rows (f32)
  __shared__ __attribute__((aligned(16))) bf16_t dpan[PSLICE * LDP];
  __shared__ __attribute__((aligned(16))) float dsum[PSLICE * LDS_];
  __shared__ unsigned s_gen;
  int groups, group, part;
  pers_ids(P, groups, group, part);
  const int dir = group & 1, b0 = p.c0 + (group >> 1) * PSLICE;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, g = lane >> 4;
  const int ul = 16 * w + 4 * g;                 // local index of this lane's 4 units inside the part
  const int u0 = part * PUNITS + ul;
  const int clip = b0 + li;
  const bool clip_ok = clip < p.c0 + p.nc;
  const int len = clip_ok ? p.lengths[clip] : 0;
  const int T = p.T;
  if (tid == 0) s_gen = __hip_atomic_load(&sync[0], __ATOMIC_RELAXED, SS_AGENT);
  if (plays_dead(sync)) {  // wave-uniform; tests only
    __syncthreads();
    finish_launch(sync, s_gen);
    return;
  }
  __builtin_amdgcn_s_setprio(3);

  // A fragments: rows = output units 16 (MT w + mt) + li, k = local row 32 ks + 8 g + j = (gate, unit of this part)
  s16x8 fa[MT][KSB];
  {
    const bf16_t* Wt = p.whht + (long)dir * 3 * H * H;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int ks = 0; ks < KSB; ++ks) {
        const int kk = 32 * ks + 8 * g;  // an 8-chunk never straddles a gate (64 % 8 == 0)
        fa[mt][ks] = *reinterpret_cast<const s16x8*>(Wt + (long)(16 * (MT * w + mt) + li) * (3 * H) + (kk / PUNITS) * H +
                                                     part * PUNITS + kk % PUNITS);
      }
  }
  __syncthreads();
  const unsigned gen = s_gen;
  const unsigned base = (gen & 0x3FFFFFu) << 10;
  bool dead = false;
  const bool same_xcd = partners_share_xcd(xid, group, part, P, base, &sync[2], lane, &dead);
  if (tid == 0 && same_xcd) atomicAdd(&sync[3], 1u);

  const long N = (long)p.B * T;
  const rsrc_t xrs = granule_rsrc(xg, 2L * groups * P * P * PSLICE * (PUNITS / 2));
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 dh = z4;
  f32x4 sb_r = z4, sb_z = z4, sb_n = z4, sb_q = z4;  // bias-gradient sums of this lane's (clip, 4 units)
  f32x4 go, gsc, sr, sz, sn, sq, hprev;
  // The loads of a step's inputs are UNCONDITIONAL (a lane past its clip's end, or without a previous state, reads a row of a valid
  // clip; the step tests `valid` / `hp_ok` where it uses the values): see the forward kernel -- behind `if (t < len)` the registers
  // were cleared first and hipcc waited for the step's granule stores before it touched them (3.3 k of 11.6 k cycles per step).
  auto load_inputs = [&](int s) {
    const int t = dir ? s : (T - 1 - s);
    const int tp = dir ? t + 1 : t - 1;
    const bool in = t < len;
    const long cl = clip_ok ? clip : p.c0;
    const long row = cl * T + (in ? t : 0), rowp = cl * T + ((in && tp >= 0 && tp < len) ? tp : 0);
    go = *reinterpret_cast<const f32x4*>(p.d_out + row * (2 * H) + dir * H + u0);
    // the dropout scale stays beside the raw load: multiplying here made the wave sit through the load's whole latency in
    // every step of a layer with dropout (2 700 of 12 200 cycles, stage timers)
    gsc = f32x4{1.f, 1.f, 1.f, 1.f};
    if (p.drop_p > 0.f) gsc = drop_scale4((row * (2 * H) + dir * H + u0) >> 2, p.drop_p, p.seed, p.offset);  // (wave-uniform branch)
    const float* sv = p.save + ((long)dir * N + row) * (4 * H) + u0;
    sr = *reinterpret_cast<const f32x4*>(sv);
    sz = *reinterpret_cast<const f32x4*>(sv + H);
    sn = *reinterpret_cast<const f32x4*>(sv + 2 * H);
    sq = *reinterpret_cast<const f32x4*>(sv + 3 * H);
    hprev = *reinterpret_cast<const f32x4*>(p.out + rowp * (2 * H) + dir * H + u0);
  };
  load_inputs(0);
  STAMP_ENTRY;
  STAMP_DECL;
  for (int s = 0; s < T; ++s) {
    STAMP(15);
    const int t = dir ? s : (T - 1 - s);
    const long row = (long)clip * T + t;
    const bool valid = t < len;
    const bool last = s + 1 == T;  // nothing consumes the last d h_prev
    const int tpc = dir ? t + 1 : t - 1;
    const bool hp_ok = tpc >= 0 && tpc < len;
    f32x4 dcarry = dh;
    f32x4 dar = z4, daz = z4, dan = z4, dqn = z4;
    if (valid) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = go[e] * gsc[e] + dh[e];
        const float dn = d * (1.0f - sz[e]);
        const float dz = d * ((hp_ok ? hprev[e] : 0.f) - sn[e]);
        dan[e] = dn * (1.0f - sn[e] * sn[e]);
        dar[e] = dan[e] * sq[e] * sr[e] * (1.0f - sr[e]);
        daz[e] = dz * sz[e] * (1.0f - sz[e]);
        dqn[e] = dan[e] * sr[e];
        dcarry[e] = d * sz[e];
      }
    }
    if (dead) dar = daz = dan = dqn = f32x4{NAN_F, NAN_F, NAN_F, NAN_F};
    // The next step's inputs are requested HERE, into the registers the gate gradients above have just finished with: loads return
    // in order, so requested behind the publish (as the first version did) the sweep's L2 hits queued behind six HBM misses and
    // every step paid max(miss, exchange) -- "next step's inputs" 3.3 k of 11.6 k cycles in the stage timers.
    if (!last) load_inputs(s + 1);
    const uint2 br_ = pack_bf16x4(dar[0], dar[1], dar[2], dar[3]), bz_ = pack_bf16x4(daz[0], daz[1], daz[2], daz[3]),
                bq_ = pack_bf16x4(dqn[0], dqn[1], dqn[2], dqn[3]);
    if (!last) {
      bf16_t* dp = &dpan[li * LDP + ul];
      *reinterpret_cast<uint2*>(dp) = br_;
      *reinterpret_cast<uint2*>(dp + PUNITS) = bz_;
      *reinterpret_cast<uint2*>(dp + 2 * PUNITS) = bq_;
      STAMP(5);
      __syncthreads();
      STAMP(6);
      f32x4 acc[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = z4;
      s16x8 fb[KSB];
#pragma unroll
      for (int ks = 0; ks < KSB; ++ks) fb[ks] = *reinterpret_cast<const s16x8*>(&dpan[li * LDP + 32 * ks + 8 * g]);
#pragma unroll
      for (int ks = 0; ks < KSB; ++ks)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma_bf16(fa[mt][ks], fb[ks], acc[mt]);
      // D rows 4g .. 4g+3 = units U .. U+3 of tile MT w + mt, column li = clip: the partial goes to the part that owns U,
      // region [parity][group][dst][src][clip][unit], one granule pair = 4 units of one clip
      const int xw = ((s & 1) * groups + group) * (P * P * PSLICE * PUNITS / 4);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int U = 16 * (MT * w + mt) + 4 * g;
        const int dst = xw + (((U / PUNITS) * P + part) * PSLICE + li) * (PUNITS / 4) + (U % PUNITS) / 4;
        const uint2 pb = pack_bf16x4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
        store_pair_raw(xrs, dst, base + (unsigned)s + 1u, pb.x, pb.y, same_xcd);
      }
      STAMP(7);
    }
    sb_r += dar; sb_z += daz; sb_n += dan; sb_q += dqn;
    if (clip_ok) {
      if (p.dG) {
        float* gp = p.dG + ((long)dir * N + row) * (4 * H) + u0;
        *reinterpret_cast<f32x4*>(gp) = dar;
        *reinterpret_cast<f32x4*>(gp + H) = daz;
        *reinterpret_cast<f32x4*>(gp + 2 * H) = dan;
        *reinterpret_cast<f32x4*>(gp + 3 * H) = dqn;
      }
      if (p.dG_bf) {
        bf16_t* gb = p.dG_bf + ((long)dir * N + row) * (4 * H) + u0;
        *reinterpret_cast<uint2*>(gb) = br_;
        *reinterpret_cast<uint2*>(gb + H) = bz_;
        *reinterpret_cast<uint2*>(gb + 2 * H) = pack_bf16x4(dan[0], dan[1], dan[2], dan[3]);
        *reinterpret_cast<uint2*>(gb + 3 * H) = bq_;
      }
    }
    if (last) break;
    STAMP(8);
    STAMP(9);
    // sum the P partials of this part's units: thread -> granule pair tid of the [clip][64] tile (clip tid / 16, units
    // 4 (tid % 16) ..), one per source part
    unsigned xv[2 * P];
    const int xr = ((((s & 1) * groups + group) * P + part) * P) * (PSLICE * PUNITS / 4) + tid;
    if (!dead) dead = !sweep_pairs<P>(xrs, xr, PSLICE * PUNITS / 4, base + (unsigned)s + 1u, xv, &sync[2], lane);
    STAMP(10);
    {
      f32x4 sm = z4;
#pragma unroll
      for (int q = 0; q < P; ++q) {
        sm[0] += __uint_as_float(xv[2 * q] << 16);
        sm[1] += __uint_as_float(xv[2 * q] & 0xffff0000u);
        sm[2] += __uint_as_float(xv[2 * q + 1] << 16);
        sm[3] += __uint_as_float(xv[2 * q + 1] & 0xffff0000u);
      }
      *reinterpret_cast<f32x4*>(&dsum[(tid >> 4) * LDS_ + 4 * (tid & 15)]) = sm;
    }
    __syncthreads();
    dh = dcarry + *reinterpret_cast<const f32x4*>(&dsum[li * LDS_ + ul]);
    STAMP(11);
  }
  STAMP_FLUSH();
  if (p.g_bih[0]) {  // sum over the 16 clips of the slice (the lanes of a row), one atomic per unit and workgroup
    float* gbi = p.g_bih[dir];
    float* gbh = p.g_bhh[dir];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = row_sum(sb_r[e]), b = row_sum(sb_z[e]), c = row_sum(sb_n[e]), d = row_sum(sb_q[e]);
      if (li == 0) {
        atomicAdd(gbi + u0 + e, a); atomicAdd(gbi + H + u0 + e, b); atomicAdd(gbi + 2 * H + u0 + e, c);
        atomicAdd(gbh + u0 + e, a); atomicAdd(gbh + H + u0 + e, b); atomicAdd(gbh + 2 * H + u0 + e, d);
      }
    }
  }
  finish_launch(sync, gen);
}

// sync workspace sections (granules of 8 bytes) for up to `groups` groups
inline long pers_xid_granules(int groups, int P) { return ((long)groups * P + 7) / 8 * 8; }
inline long pers_fwd_granules(int groups, int H) { return 2L * groups * PSLICE * (H / 2); }
inline long pers_bwd_granules(int groups, int H) { return 2L * groups * (H / PUNITS) * (H / PUNITS) * PSLICE * (PUNITS / 2); }

inline bool pers_supported(int H) { return H >= 128 && H <= 512 && H % 64 == 0; }

// clips per launch: as many 16-clip slices as fit the chip with both directions and all parts co-resident; several launches
// over equal clip chunks otherwise (clips are independent), a multiple of 4 slices each where that fits (group count % 8 == 0:
// the partners of a group then share an XCD)
inline int pers_chunk_clips(int B, int H, int cus) {
  const int P = H / PUNITS;
  const int slices = cus / (2 * P);
  if (slices < 1) return 0;
  const int need = ceil_div(B, PSLICE);
  if (slices >= need) return B;
  const int launches = ceil_div(need, slices);
  int per = ceil_div(need, launches);
  if ((per + 3) / 4 * 4 <= slices) per = (per + 3) / 4 * 4;
  return per * PSLICE;
}

}  // namespace
