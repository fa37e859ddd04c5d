// Latency form of the masked BiGRU recurrence: one (16-clip slice, direction) spread over P compute units.
//
// The single-workgroup kernels in gru.hip keep the whole 3H x H matrix W_hh on one CU, so a time step can never be
// shorter than 3H*H*16*2 flop / (one CU's MFMA rate) = 5.8 us at H = 192 -- and the recurrence is a chain of T such
// steps with nothing else to do on the other 224 CUs.  Here part p of P owns hidden units [p*H/P, (p+1)*H/P):
//   forward   it holds those rows of W_hh (all three gates), multiplies them by the FULL previous state and
//             publishes its slice of the new state (an all-gather of 16 x H floats per step);
//   backward  the transposed product dh_prev = W_hh^T d_pre is split over its CONTRACTION index instead: the part
//             multiplies by the 3*H/P pre-activation gradients of its own units (no gather needed), gets a partial
//             dh_prev for ALL H units and publishes it; every part sums the P slices of its own units (a
//             reduce-scatter, 16 x H floats into each workgroup per step where an all-gather of d_pre would pull
//             16 x 3H through every one).
//
// Exchange: 8-byte {value, tag} granules written with ONE agent-scope (sc1, write-through) store each and swept
// with agent-scope loads until every tag matches -- the data is its own flag, so a hop costs one store->load
// round trip and needs no fence, no drain and no barrier on the producer side (MI355X_MICROARCH.md, hand-off price
// list: "handoff-1to1" vs "handoff-flag"; measured here: 5.3 us/step with flags).  tag =
// (generation << 10) + step + 1; the generation lives in the sync header and is bumped by the last workgroup of
// every launch, so granules left by an earlier launch -- or an earlier replay of a captured graph -- never match and
// nothing has to be cleared between launches.  Buffers alternate with the step parity: a slot is rewritten two
// steps later, which its reader has provably left (it had to publish the step in between first).
//
// Same-XCD fast path: a write-through (sc1) granule leaves the writer's L2 and a partner reads it back from memory.
// When all P partners of a pair run on ONE XCD -- workgroups are dealt round-robin over the 8 XCDs, so the grid is
// ordered part-major whenever the pair count is a multiple of 8 -- their shared L2 is the coherence point: plain
// stores stay in it and the partners' sc1 loads (L1-bypassing, L2-served) find them there, a shorter round trip
// (forward 3.5 -> 3.0 us per step, backward 4.7 -> 3.4).  Placement is never assumed: every workgroup reads its
// HW_REG_XCC_ID, the partners exchange the ids once per launch through write-through granules, and only a pair whose
// ids all agree switches to plain stores; the rest keep the write-through form.  Dirty granule lines never outlive a
// launch (the end-of-kernel release writes the L2 back), and a stale copy can only ever show an old tag.
//
// Progress: the launcher only picks this form when the whole
// grid is co-resident (<= 240 workgroups of 4 waves), and every wait is bounded -- a lost partner poisons the output
// with NaN and counts an error in the sync header instead of hanging the GPU.
//
// sync_ws layout: [0] generation, [1] finished workgroups of the running launch, [2] sweep time-outs ever seen,
// [3] workgroup-launches that took the same-XCD fast path, [5] fault injection (tests only, see plays_dead),
// [.. 64) pad | XCD ids [pairs][P] u64 | forward granules [2][pairs][16][H] u64 | backward granules
// [2][pairs][P dest][P src][16][H/P] u64.
// The owner zeroes it once; after that the kernels keep it consistent.
#pragma once
#include "granule_xchg.h"

namespace {

// co-residency bound of the multi-CU recurrence: one 4-wave workgroup per CU, 16 CUs of slack (240 on MI355X's 256 CUs)
inline int split_max_wgs() { return ss_device_cus() - 16; }
constexpr int SPLIT_MAX_T = 1022;

template <int H, int P>
struct SplitCfg {
  static constexpr int UP = H / P;          // hidden units per part
  static constexpr int UT = UP / 16;        // 16-unit MFMA tiles per part
  static constexpr int KSPLIT = 4 / UT;     // forward: waves sharing one tile split the contraction
  static_assert((UP == 32 || UP == 16) && UT * KSPLIT == 4, "a part is one or two unit tiles");
  static constexpr int QF = H / 16 / KSPLIT;       // 16-wide k groups per wave, forward (K = H)
  static_assert(QF * 16 * KSPLIT == H, "k split");
  static constexpr int RED = (KSPLIT - 1) * UT * 3 * 64 * 4;
  // (Measured, not kept: every wave of a tile finishing 4 / KSPLIT of a lane's four units -- gates, publish, stash -- instead of the
  // owner wave all four, the k slices handed over as [wave][gate][unit][lane] and summed in the same order (bit-identical, tests
  // green): forward launch 0.363 -> 0.393 ms at the shipped batch of 16, 0.173 -> 0.185 at config 2.  Twelve 4-byte LDS stores and
  // loads and 4-byte global traffic per lane cost more than three quarters of the transcendental chain save.)
  static constexpr int NGP = SLICE * H / 512;      // granule pairs per thread in a sweep of a 16 x H panel
};

// grid: pairs * P workgroups, blockIdx.x = pair * P + part, pair = slice * 2 + direction
template <int H, int P>
__global__ __launch_bounds__(256) void gru_split_fwd_kernel(GruFwdParams p, unsigned* sync, u64* xid, u64* hx) {
  using C = SplitCfg<H, P>;
  constexpr int LDH = H + 4;  // panel row stride: 16-byte aligned rows, clips 4 banks apart
  __shared__ __attribute__((aligned(16))) float red[C::RED];
  __shared__ __attribute__((aligned(16))) float hpan[SLICE * LDH];  // previous state of the slice, [clip][unit]
  __shared__ unsigned s_gen;
  // part-major when the pair count is a multiple of 8: workgroups b and b + 8k share an XCD, so do the partners then
  // (the launch pads the pair count to a multiple of 8 where the padded grid still fits -- gru_split_grid_pairs: workgroups of the
  // pad pairs leave at once -- so small batches, the reference's own 16 clips among them, get the same-XCD exchange too)
  const int pairs = 2 * ((p.B + SLICE - 1) / SLICE), grid_pairs = gridDim.x / P;
  const bool part_major = (grid_pairs & 7) == 0;
  const int part = part_major ? blockIdx.x / grid_pairs : blockIdx.x % P;
  const int pair = part_major ? blockIdx.x % grid_pairs : blockIdx.x / P;
  const int dir = pair & 1, b0 = (pair >> 1) * SLICE;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int ut = w % C::UT, kh = w / C::UT;
  const bool owner = kh == 0;
  const int i = lane & 15, g = lane >> 4;
  const int j0 = part * C::UP + 16 * ut + 4 * g;  // first of this lane's 4 hidden units (D rows 4g..4g+3)
  const int clip = b0 + i;
  const bool clip_ok = clip < p.B;
  const int len = clip_ok ? p.lengths[clip] : 0;
  const int T = p.T;
  if (threadIdx.x == 0) s_gen = __hip_atomic_load(&sync[0], __ATOMIC_RELAXED, SS_AGENT);
  if (plays_dead(sync) || pair >= pairs) {  // wave-uniform; tests only, resp. a workgroup of the grid's pad
    __syncthreads();
    finish_launch(sync, s_gen);
    return;
  }
  // every cycle of this chain is on the step's critical path: issue ahead of the weight-gradient GEMM waves that
  // share the SIMDs
  __builtin_amdgcn_s_setprio(3);

  // A fragments: rows = this tile's units of gate G, slots (q, e) carry k = 16*(kh*QF + q) + 4g + e
  float wf[3][4 * C::QF];
  {
    const float* W = p.w_hh[dir];
#pragma unroll
    for (int G = 0; G < 3; ++G)
#pragma unroll
      for (int q = 0; q < C::QF; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(W + (long)(G * H + part * C::UP + 16 * ut + i) * H +
                                                        16 * (kh * C::QF + q) + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) wf[G][4 * q + e] = v[e];
      }
  }
  f32x4 br = {0.f, 0.f, 0.f, 0.f}, bz = br, bn = br;
  if (owner) {
    br = *reinterpret_cast<const f32x4*>(p.b_hh[dir] + j0);
    bz = *reinterpret_cast<const f32x4*>(p.b_hh[dir] + H + j0);
    bn = *reinterpret_cast<const f32x4*>(p.b_hh[dir] + 2 * H + j0);
  }
  __syncthreads();
  const unsigned gen = s_gen;
  const unsigned base = (gen & 0x3FFFFFu) << 10;
  bool dead = false;  // a partner never arrived: everything this workgroup emits from then on is NaN
  const bool same_xcd = partners_share_xcd(xid, pair, part, P, base, &sync[2], lane, &dead);
  if (threadIdx.x == 0 && same_xcd) atomicAdd(&sync[3], 1u);  // observability: workgroup-launches on the fast path

  f32x4 hp = {0.f, 0.f, 0.f, 0.f};
  const long dir_off = (long)dir * p.B * T;
  const rsrc_t hrs = granule_rsrc(hx, 2L * pairs * SLICE * H);
  // (unconditional for the owner waves: a lane past its clip's end reads its clip's first row and never uses the values -- behind
  // `valid` the registers were cleared first, and hipcc waits for a step's younger stores before it overwrites the destination
  // of an older load)
  // input-projection gates of a step: independent of the recurrence, requested a step AHEAD behind the previous step's sweep into
  // the other of two register sets (the step loop runs two steps per turn).  The first version requested them at the top of their
  // own step, "in flight during the sweep": loads return in order, so the sweep's L2 hits queued behind these three HBM misses
  // (found in the bf16 kernels' stage timers); one set copied at the top of the step made the copy wait for them there.
  struct Gi { f32x4 r, z, n; };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  Gi gA{zero4, zero4, zero4}, gB{zero4, zero4, zero4};
  auto load_gi = [&](int s, Gi& d) {
    if (!owner) return;  // wave-uniform
    const int t = dir ? (T - 1 - s) : s;
    const long fl = (long)(clip_ok ? clip : b0) * T + (t < len ? t : 0);
    const float* gp = p.gi + (dir_off + fl) * (3 * H) + j0;
    d.r = *reinterpret_cast<const f32x4*>(gp);
    d.z = *reinterpret_cast<const f32x4*>(gp + H);
    d.n = *reinterpret_cast<const f32x4*>(gp + 2 * H);
  };
  load_gi(0, gA);
  STAMP_ENTRY;
  STAMP_DECL;
  auto step = [&](int s, const Gi& gc, Gi& gnx) {
    STAMP(15);
    const int t = dir ? (T - 1 - s) : s;
    const long frame = (long)clip * T + t;
    const bool valid = t < len;
    if (s == 0 && T > 1) load_gi(1, gnx);
    f32x4 ar = br, az = bz, an = bn;
    if (s > 0) {
      // the full previous state of the slice, swept ONCE per workgroup into an LDS panel (what a step publishes is
      // what it emits: a masked step emits what it keeps -- zeros in the reverse direction -- and the forward
      // direction never uses a frozen state again)
      float hv[2 * C::NGP];
      const int hr = (((s - 1) & 1) * pairs + pair) * (SLICE * H / 2) + threadIdx.x;
      if (!dead) dead = !sweep_granules<C::NGP>(hrs, hr, base + (unsigned)s, hv, &sync[2], lane);
      if (s + 1 < T) load_gi(s + 1, gnx);  // behind the sweep: a whole step to arrive
      STAMP(0);
#pragma unroll
      for (int k = 0; k < C::NGP; ++k) {
        const int idx = 2 * (threadIdx.x + 256 * k);
        *reinterpret_cast<float2*>(&hpan[(idx / H) * LDH + idx % H]) = float2{hv[2 * k], hv[2 * k + 1]};
      }
      __syncthreads();
      STAMP(1);
      f32x4 hq[C::QF];
#pragma unroll
      for (int q = 0; q < C::QF; ++q) hq[q] = *reinterpret_cast<const f32x4*>(&hpan[i * LDH + 16 * (kh * C::QF + q) + 4 * g]);
#pragma unroll
      for (int q = 0; q < C::QF; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ar = mfma16(wf[0][4 * q + e], hq[q][e], ar);
          az = mfma16(wf[1][4 * q + e], hq[q][e], az);
          an = mfma16(wf[2][4 * q + e], hq[q][e], an);
        }
    }
    STAMP(2);
    if (C::KSPLIT > 1) {  // sum the k slices of a tile in its owner wave
      if (!owner) {
        f32x4* r4 = reinterpret_cast<f32x4*>(red) + (((kh - 1) * C::UT + ut) * 3) * 64 + lane;
        r4[0] = ar; r4[64] = az; r4[128] = an;
      }
      __syncthreads();
      if (owner) {
#pragma unroll
        for (int k2 = 1; k2 < C::KSPLIT; ++k2) {
          const f32x4* r4 = reinterpret_cast<const f32x4*>(red) + (((k2 - 1) * C::UT + ut) * 3) * 64 + lane;
          ar += r4[0]; az += r4[64]; an += r4[128];
        }
      }
    }
    STAMP(3);
    if (owner) {
      f32x4 r, z, n, hn;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        r[e] = sigmoid_f(gc.r[e] + ar[e]);
        z[e] = sigmoid_f(gc.z[e] + az[e]);
        n[e] = tanh_f(gc.n[e] + r[e] * an[e]);
        hn[e] = (1.0f - z[e]) * n[e] + z[e] * hp[e];
      }
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (valid) {
        hp = hn;
        o = hn;
      }
      if (dead) o = f32x4{NAN_F, NAN_F, NAN_F, NAN_F};  // padding rows too: the tail masks by length, the poison must not hide
      if (s + 1 < T) {  // publish (padding clips too: the partners sweep whole panels)
        const int hw = ((((s & 1) * pairs + pair) * SLICE + i) * H + j0) / 2;
        store_granule_pair(hrs, hw, base + (unsigned)s + 1u, o[0], o[1], same_xcd);
        store_granule_pair(hrs, hw + 1, base + (unsigned)s + 1u, o[2], o[3], same_xcd);
      }
      STAMP(4);
      if (clip_ok) {
        *reinterpret_cast<f32x4*>(p.out + frame * (2 * H) + dir * H + j0) = o;
        if (p.out_drop) {  // (wave-uniform) behind the publish: the Philox rounds run under the partners' exchange
          const long e0 = frame * (2 * H) + dir * H + j0;
          f32x4 od = o;
          if (p.drop_p > 0.f) od *= drop_scale4(e0 >> 2, p.drop_p, p.drop_seed, p.drop_off);
          *reinterpret_cast<f32x4*>(p.out_drop + e0) = od;
        }
        if (p.save && valid) {
          float* sp = p.save + (dir_off + frame) * (4 * H) + j0;
          *reinterpret_cast<f32x4*>(sp) = r;
          *reinterpret_cast<f32x4*>(sp + H) = z;
          *reinterpret_cast<f32x4*>(sp + 2 * H) = n;
          *reinterpret_cast<f32x4*>(sp + 3 * H) = an;
        }
      }
    }
  };
  for (int s = 0; s < T; s += 2) {
    step(s, gA, gB);
    if (s + 1 < T) step(s + 1, gB, gA);
  }
  STAMP_FLUSH();
  finish_launch(sync, gen);
}

// xg: [2 step parity][pairs][P dest parts][P source parts][16 clips][UP]  partial dh_prev granules
template <int H, int P>
__global__ __launch_bounds__(256) void gru_split_bwd_kernel(GruBwdParams p, unsigned* sync, u64* xid, u64* xg) {
  using C = SplitCfg<H, P>;
  constexpr int KR = 3 * C::UP;        // contraction length of this part: its r | z | n rows
  constexpr int QK = KR / 16;          // 16-wide k groups
  constexpr int NT = H / 16 / 4;       // output unit tiles per wave
  constexpr int LDP = KR + 4;
  // granule pairs per thread: P sources x 16 * UP / 2 position pairs over 256 threads.  UP = 32: thread = position pair, one per
  // source; UP = 16 (twelve parts): threads 0..127 take the even sources, 128..255 the odd ones of position pair tid % 128
  constexpr int NGP = P * C::UP / 32;
  static_assert(QK * 16 == KR && NT * 64 == H && (SLICE * C::UP == 512 || SLICE * C::UP == 256), "tile split");
  __shared__ __attribute__((aligned(16))) float dpan[SLICE * LDP];     // this part's d_pre, [clip][local row]
  __shared__ __attribute__((aligned(16))) float dsum[SLICE * 32];      // summed dh_prev of this part's units (UP = 16: two halves)
  __shared__ unsigned s_gen;
  // part-major when the pair count is a multiple of 8: workgroups b and b + 8k share an XCD, so do the partners then
  // (the launch pads the pair count to a multiple of 8 where the padded grid still fits -- gru_split_grid_pairs: workgroups of the
  // pad pairs leave at once -- so small batches, the reference's own 16 clips among them, get the same-XCD exchange too)
  const int pairs = 2 * ((p.B + SLICE - 1) / SLICE), grid_pairs = gridDim.x / P;
  const bool part_major = (grid_pairs & 7) == 0;
  const int part = part_major ? blockIdx.x / grid_pairs : blockIdx.x % P;
  const int pair = part_major ? blockIdx.x % grid_pairs : blockIdx.x / P;
  const int dir = pair & 1, b0 = (pair >> 1) * SLICE;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int ut = w % C::UT;
  const bool owner = w < C::UT;        // owner waves carry dh of one unit tile of this part
  const int i = lane & 15, g = lane >> 4;
  const int j0 = part * C::UP + 16 * ut + 4 * g;
  const int clip = b0 + i;
  const bool clip_ok = clip < p.B;
  const int len = clip_ok ? p.lengths[clip] : 0;
  const int T = p.T;
  if (threadIdx.x == 0) s_gen = __hip_atomic_load(&sync[0], __ATOMIC_RELAXED, SS_AGENT);
  if (plays_dead(sync) || pair >= pairs) {  // wave-uniform; tests only, resp. a workgroup of the grid's pad
    __syncthreads();
    finish_launch(sync, s_gen);
    return;
  }
  __builtin_amdgcn_s_setprio(3);  // as in the forward kernel

  // A fragments: A[i = output unit 16*(NT*w + nt) + i][slot (q, e)] = W[row(16q + 4g + e)][unit], local row
  // kk = G * UP + u  <->  W row G*H + part*UP + u
  float wf[NT][4 * QK];
  {
    const float* W = p.w_hh[dir];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < QK; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kk = 16 * q + 4 * g + e;
          wf[nt][4 * q + e] = W[(long)((kk / C::UP) * H + part * C::UP + kk % C::UP) * H + 16 * (NT * w + nt) + i];
        }
  }
  __syncthreads();
  const unsigned gen = s_gen;
  const unsigned base = (gen & 0x3FFFFFu) << 10;
  bool dead = false;  // a partner never arrived: every gate gradient this workgroup emits from then on is NaN
  const bool same_xcd = partners_share_xcd(xid, pair, part, P, base, &sync[2], lane, &dead);
  if (threadIdx.x == 0 && same_xcd) atomicAdd(&sync[3], 1u);  // observability: workgroup-launches on the fast path

  f32x4 dh = {0.f, 0.f, 0.f, 0.f};
  GruBiasAcc bacc;
  const long dir_off = (long)dir * p.B * T;
  const rsrc_t xrs = granule_rsrc(xg, 2L * pairs * P * SLICE * H);
  // inputs of one step (owner waves): loaded a step ahead, while the exchange of the current step is in flight
  f32x4 go, gsc, sr, sz, sn, sq, hprev;
  auto load_inputs = [&](int s) {
    const int t = dir ? s : (T - 1 - s);
    const int tp = dir ? t + 1 : t - 1;
    // UNCONDITIONAL loads (an owner lane past its clip's end, or without a previous state, reads a row of its clip's first step
    // and the step tests `valid` / `hp_ok` where it uses the values): behind `if (t < len)` the registers were cleared first, and a
    // write to the destination of an older load makes hipcc wait for that load by COUNT -- it cannot count the younger stores issued
    // under divergent branches, so it waited for the step's granule stores as well (found in the bf16 kernel's stage timers:
    // 3.3 k of 11.6 k cycles per step)
    gsc = f32x4{1.f, 1.f, 1.f, 1.f};
    if (owner) {  // (the non-owner lanes of a wave hold no units: wave-uniform by construction of the lane map)
      const bool in = t < len;
      const long cl = clip_ok ? clip : b0;
      const long frame = cl * T + (in ? t : 0), framep = cl * T + ((in && tp >= 0 && tp < len) ? tp : 0);
      go = *reinterpret_cast<const f32x4*>(p.d_out + frame * (2 * H) + dir * H + j0);
      // the dropout scale stays beside the raw load and is applied where the gradient is used, a step later: multiplying here
      // made the wave wait for the load it had just issued -- a memory latency in front of every sweep of a layer with dropout
      // (found with the stage timers of the bf16 kernel, which had inherited the line)
      if (p.drop_p > 0.f) gsc = drop_scale4((frame * (2 * H) + dir * H + j0) >> 2, p.drop_p, p.drop_seed, p.drop_off);
      const float* sp = p.save + (dir_off + frame) * (4 * H) + j0;
      sr = *reinterpret_cast<const f32x4*>(sp);
      sz = *reinterpret_cast<const f32x4*>(sp + H);
      sn = *reinterpret_cast<const f32x4*>(sp + 2 * H);
      sq = *reinterpret_cast<const f32x4*>(sp + 3 * H);
      hprev = *reinterpret_cast<const f32x4*>(p.out + framep * (2 * H) + dir * H + j0);
    }
  };
  load_inputs(0);
  STAMP_ENTRY;
  STAMP_DECL;
  for (int s = 0; s < T; ++s) {
    STAMP(15);
    const int t = dir ? s : (T - 1 - s);
    const long frame = (long)clip * T + t;
    const bool valid = t < len;
    const bool last = s + 1 == T;  // nothing consumes the last dh_prev
    const int tpc = dir ? t + 1 : t - 1;
    const bool hp_ok = tpc >= 0 && tpc < len;
    f32x4 dcarry = dh;
    f32x4 dar = {0.f, 0.f, 0.f, 0.f}, daz = dar, dan = dar, dqn = dar;
    if (owner) {
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
      if (!last) {
        float* dp = &dpan[i * LDP + 16 * ut + 4 * g];
        *reinterpret_cast<f32x4*>(dp) = dar;
        *reinterpret_cast<f32x4*>(dp + C::UP) = daz;
        *reinterpret_cast<f32x4*>(dp + 2 * C::UP) = dqn;
      }
    }
    // The next step's inputs are requested here, into the registers the gate gradients above have just finished with: loads return
    // in order, so requested behind the publish (the first version) the sweep's L2 hits queued behind six HBM misses
    // (found in the bf16 kernel's stage timers: 3.3 k of 11.6 k cycles per step)
    if (!last) load_inputs(s + 1);
    if (!last) {
      __syncthreads();
      STAMP(4);
      f32x4 acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 dq[QK];
#pragma unroll
      for (int q = 0; q < QK; ++q) dq[q] = *reinterpret_cast<const f32x4*>(&dpan[i * LDP + 16 * q + 4 * g]);
#pragma unroll
      for (int q = 0; q < QK; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(wf[nt][4 * q + e], dq[q][e], acc[nt]);
      // D rows 4g..4g+3 = units U of tile NT*w + nt, column i = clip: the partial goes to the part that owns U
      const int xw = ((s & 1) * pairs + pair) * P * P * SLICE * C::UP;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int U = 16 * (NT * w + nt) + 4 * g;
        const int dst = (xw + (((U / C::UP) * P + part) * SLICE + i) * C::UP + U % C::UP) / 2;
        store_granule_pair(xrs, dst, base + (unsigned)s + 1u, acc[nt][0], acc[nt][1], same_xcd);
        store_granule_pair(xrs, dst + 1, base + (unsigned)s + 1u, acc[nt][2], acc[nt][3], same_xcd);
      }
      STAMP(2);
    }
    if (owner) bacc.add(dar, daz, dan, dqn);
    if (owner && clip_ok) {
      float* gp = p.d_g + (dir_off + frame) * (4 * H) + j0;
      *reinterpret_cast<f32x4*>(gp) = dar;
      *reinterpret_cast<f32x4*>(gp + H) = daz;
      *reinterpret_cast<f32x4*>(gp + 2 * H) = dan;
      *reinterpret_cast<f32x4*>(gp + 3 * H) = dqn;
    }
    if (last) break;
    // sum the P partials of this part's units: thread -> positions 2 tid, 2 tid + 1 of the [clip][UP] tile
    float xv[2 * NGP];
    const int xr = ((((s & 1) * pairs + pair) * P + part) * P * SLICE * C::UP) / 2 + threadIdx.x;
    if (!dead) dead = !sweep_granules<NGP>(xrs, xr, base + (unsigned)s + 1u, xv, &sync[2], lane);
    STAMP(0);
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int q = 0; q < NGP; ++q) {
      s0 += xv[2 * q];
      s1 += xv[2 * q + 1];
    }
    *reinterpret_cast<float2*>(&dsum[2 * threadIdx.x]) = float2{s0, s1};
    __syncthreads();
    if (owner) {
      dh = dcarry + *reinterpret_cast<const f32x4*>(&dsum[i * C::UP + 16 * ut + 4 * g]);
      if (C::UP == 16) dh += *reinterpret_cast<const f32x4*>(&dsum[SLICE * 16 + i * 16 + 4 * g]);  // the odd sources' half
    }
    STAMP(1);
  }
  if (owner) bacc.flush(p, dir, H, j0, i);  // owner is wave-uniform
  STAMP_FLUSH();
  finish_launch(sync, gen);
}

// sync_ws sections, in granules
constexpr int SPLIT_MAX_PARTS = 12;  // the XCD-id rows of the workspace are laid out for the widest split (forward, small batches)
inline long gru_xid_granules(int B) { return ((2L * ceil_div(B, SLICE) * SPLIT_MAX_PARTS) + 7) / 8 * 8; }
inline long gru_fwd_granules(int B, int H) { return 2L * (2 * ceil_div(B, SLICE)) * SLICE * H; }
inline long gru_bwd_granules(int B, int H, int P) { return 2L * (2 * ceil_div(B, SLICE)) * P * SLICE * H; }

// pairs the grid is launched with: the next multiple of 8 when that still fits the co-residency bound.  Workgroup b runs on XCD
// b % 8 (round-robin dispatch), so with a multiple of 8 pairs and part-major numbering the P partners of a pair share an XCD and
// exchange their granules through its L2; a grid of 2 pairs x 6 parts (B = 16) had every partner on another XCD.
inline int gru_split_grid_pairs(int B, int P) {
  const int pairs = 2 * ceil_div(B, SLICE), pad = (pairs + 7) / 8 * 8;
  return pad * P <= split_max_wgs() ? pad : pairs;
}

// The recurrences of H = 192 over twelve parts of 16 units where the padded grid has room (up to 16 pairs = 128 clips, the
// reference's shipped batch of 16 among them): the MFMA stage of a step is at its floor for four waves (288 MFMAs per workgroup:
// 0.96 us of a 2.6 us step); half the units per workgroup halve it and the gate / publish stage: 2.6 -> 2.0 us per step.  The
// sweep is unchanged (every workgroup reads the whole 16 x H panel whatever the split).  The backward kernel likewise: its
// contraction over the part's 3 x 16 gate rows is 144 MFMAs per workgroup instead of 288, a workgroup still publishes one 16 x H
// panel of partial sums and sweeps P x 16 x 16 of them.  The forward contraction is then cut into four k slices instead of two, so a clip's state differs in the last
// bits between a batch of <= 128 and a larger one.  (Summing four fixed segments in both splits makes the bits equal -- measured:
// tests green -- but costs the six-part kernel, i.e. BASELINE config 2, 6 us per step for three more accumulators; not kept.)
inline int gru_split_small_parts(int B, int P) {
  if (P != 6) return P;
  const int pad = (2 * ceil_div(B, SLICE) + 7) / 8 * 8;
  return pad * 12 <= split_max_wgs() ? 12 : P;
}

// parts per (slice, direction) for a shape, or 0 when the single-workgroup form is the right one
inline int gru_split_parts(int B, int T, int H) {
  const int pairs = 2 * ceil_div(B, SLICE);
  if (T > SPLIT_MAX_T) return 0;
  const int P = H == 192 ? 6 : (H == 64 ? 2 : 0);
  return (P && pairs * P <= split_max_wgs()) ? P : 0;
}

}  // namespace
