// ROI normalise + TinyROICNN forward, fused: one persistent 512-thread workgroup per CU walks
// frames; a frame's uint8 pixels are read once from HBM and every activation stays in LDS.
//
// Replaces /root/reference/train_model_official.py:286-291 (normalise) and :212-229 (CNN); with
// standardize = 0 the live variant /root/reference/live_infer_official.py:126-127.
//
//   stage 0  u8 frame -> exact integer sum / sum of squares -> mean, unbiased std (clamp 1e-6)
//            -> xn = (u/255 - mu) / std  into a zero-haloed LDS image
//   stage 1  conv 1->8 on v_mfma_f32_16x16x4_f32: M = 16 pixels of row y, N = (8 channels) x (rows y, y+1),
//            K = 12 = the 4 input rows x 3 columns both output rows touch -- no padding in N or K;
//            ReLU + 2x2 max-pool: column pairs inside a lane, the row pair via one lane exchange (xor 8)
//   stage 2  conv 8->16 as implicit GEMM: M = 16 pixels of one row, N = 16 out channels, K = 72 = 9 taps
//            x 8 channels; a wave computes rows y and y+1 of one 16-pixel column block, so the max-pool
//            happens in its registers
//   stage 3  conv 16->24 the same way (M = 16 linear pixels, N = 24 -> two N tiles, K = 144), ReLU and
//            the global average reduced with wave shuffles
//   stage 4  fc 24 -> E, written straight into the caller's (B,T,x_dim+E) buffer (the torch.cat)
//
// For training the pooled maps, pool argmaxes and the conv3 sign mask are stashed (HBM is cheaper
// than recomputing them in the backward kernel: 67 KB per 64x64 frame), copied out of LDS in
// 16-byte pieces while the next stage computes.
#include "ss_common.h"
#include "roi_cnn_geom.h"

STAMP_TABLE(ss_debug_stamps_fwd)

int ss_cnn_max_wgs = 0;  // shared with roi_cnn_bwd.hip; set through ss_roi_cnn_set_max_workgroups
extern "C" int ss_roi_cnn_set_max_workgroups(int n) {
  if (n < 0 || n > 4096) return SS_ERR_ARG;
  ss_cnn_max_wgs = n;
  return SS_OK;
}

namespace {

#ifndef SS_FWD_NT
#define SS_FWD_NT 256
#endif
// (Round 3: -DSS_FWD_NT=384 = two 6-wave workgroups per CU, a third wave per SIMD to fill the issue slots two leave empty -- MFMA
// busy is 0.58 with each wave in MFMAs 29 % of its time.  168 registers per wave do not hold the 93 weight fragments plus a
// stage's working set: 33 spilled, 719 us per launch against 459.  LDS (66 KB per frame image) rules out a third workgroup.
// Also measured: the NEXT frame's pixel sums taken behind conv3, its mean / std by one thread of wave 1 beside the feature sums
// and its grey-level table beside the fc -- three barriers and the serial f64 section out of a frame's front: 458.8 / 461.8 us
// against 459.0 / 458.6.  And: conv2 / conv3 with two register sets of A values, the next nine k steps' ds_reads issued under the
// current nine's MFMAs and the next unit's first nine under this unit's last (counted lgkmcnt waits in the ISA): 457.9 / 459.8
// against 459.0 / 458.8.  A workgroup's own latency chain is not what bounds the CU's rate.)
constexpr int NT = SS_FWD_NT;  // threads per workgroup; TWO workgroups per CU (LDS image <= 80 KB): while one is in an epilogue, a
                         // barrier or the statistics, the other's MFMAs keep the matrix pipes busy
constexpr int NWV = NT / 64;

struct CnnFwdParams {
  const uint8_t* R;
  int N;
  int standardize;
  const float *w1, *b1, *w2, *b2, *w3, *b3, *wfc, *bfc;
  int E;
  float* out;
  int ld_out;
  // training stash (all null together)
  float* st_a1;     // [N][8][P1]  haloed LDS image of the pooled-1 map, as is
  uint8_t* st_i1;   // [N][8][I1S] (plane stride Geom::I1S = H2*W2 + 16 bytes)
  float* st_a2;     // [N][16][P2] haloed LDS image of the pooled-2 map, as is
  uint8_t* st_i2;   // [N][H4][W4][16]  pixel-major: the backward pass consumes it next to its pixel-major da2
  uint8_t* st_m3;   // [N][P][32]  pixel-major, channels 24..31 unused: the backward turns 16 bytes into 16 floats of one pixel
  float* st_feat;   // [N][ST_FEAT]  24 averaged conv3 features, 24 counts of positive conv3 outputs (for d b3), mean, std
  const int* frames;  // null, or [0] = how many frames to walk, [1 ...] their numbers (ss_roi_active_frames): padded frames are skipped
};

// conv1 with lane-local pool windows (round 4, below) reads the frame with lanes that walk a 2 x 2 window x 4 windows: a row
// stride == 8 (mod 32) keeps the two rows of a window and the four windows on different banks (W + 2 == 2 (mod 32) put row 1 of
// window w on the banks of row 0 of window w + 1)
// MEASURED AND NOT USED (A/B on one box, tools/cnn_ab.py, 64 x 64: 437.7 us per launch against 427.3; 48 x 96: 535.1 / 538.5):
// the 128 extra MFMAs per frame cost what the 17 instructions saved per pooled output gain -- the same null result as round 1's
// lane-local pool, now with the cheap epilogue.  Kept behind the macro as the record of that experiment.
#ifndef SS_CONV1_LOCAL
#define SS_CONV1_LOCAL 0
#endif
// Diagnostic (-DSS_FWD_STOP=k, tools/fwd_stage_pmc.py): a frame ends behind stage k, so that the difference between two builds'
// hardware counters is one stage's share (LDS bank conflicts per stage: rocprofv3 has no per-stage view of a persistent kernel).
#ifdef SS_FWD_STOP
#define STAGE_END(k) if (SS_FWD_STOP == (k)) continue
#else
#define STAGE_END(k)
#endif

template <class G>
constexpr int fwd_xs() { return SS_CONV1_LOCAL ? G::XS + (8 - G::XS % 32 + 32) % 32 : G::XS; }

template <class G>
struct FwdLds {
  // region U holds the normalised frame + pool-1 argmaxes while conv1 runs, then the pooled-2 map + pool-2 argmaxes
  // (conv2 writes them when conv1's inputs are dead): the image stays under half a CU's LDS
  static constexpr int o_xh = 0;                                // [(H+2)][XS]
  static constexpr int o_i1 = ((G::H + 2) * fwd_xs<G>() + 3) & ~3;    // bytes [8][I1S]
  static constexpr int o_a2 = 0;                                // [16][P2]
  static constexpr int o_i2 = 16 * G::P2;                       // bytes [P][16]
  static constexpr int u_end1 = o_i1 + 2 * G::I1S, u_end2 = o_i2 + 4 * G::P;
  static constexpr int o_a1 = ((u_end1 > u_end2 ? u_end1 : u_end2) + 3) & ~3;  // [8][P1]
  static constexpr int o_m3 = o_a1;                      // bytes [P][32] conv3 sign mask, staged for the stash over the
                                                         // pooled-1 map (dead once conv2 is through)
  static_assert(8 * G::P <= 8 * G::P1, "mask staging fits the pooled-1 planes");
  static constexpr int o_misc = o_a1 + 8 * G::P1 + 256;  // the 256 floats in front hold the grey-level table
  static constexpr int total = o_misc + 512;
};

template <class G>
__global__ __launch_bounds__(NT, NT == 256 ? 2 : 3) void roi_cnn_fwd_kernel(CnnFwdParams p) {
  STAMP_ENTRY;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using LL = FwdLds<G>;
  constexpr int H = G::H, W = G::W, H2 = G::H2, W2 = G::W2, W4 = G::W4, HW = G::HW, P = G::P;
  constexpr int XS = fwd_xs<G>(), S1 = G::S1, P1 = G::P1, S2 = G::S2, P2 = G::P2;
  constexpr int NCH = (HW / 16 + NT - 1) / NT;  // 16-byte pixel chunks per thread
  float* xh = lds + LL::o_xh;
  float* a1 = lds + LL::o_a1;
  float* a2 = lds + LL::o_a2;
  uint8_t* i1s = reinterpret_cast<uint8_t*>(lds + LL::o_i1);
  uint8_t* i2s = reinterpret_cast<uint8_t*>(lds + LL::o_i2);
  uint8_t* m3s = reinterpret_cast<uint8_t*>(lds + LL::o_m3);
  float* misc = lds + LL::o_misc;
  float* s_b1 = misc;                       // [8]
  float* s_b2 = misc + 16;                  // [16]
  float* s_b3 = misc + 32;                  // [24] (+8 pad)
  float* s_feat = misc + 64;                // [24]
  float* s_stat = misc + 96;                // mu, std
  unsigned* s_red = reinterpret_cast<unsigned*>(misc + 104);  // [NWV][2]
  float* s_fp = misc + 128;                 // [NWV][32] per-wave partial channel sums
  float* s_cp = misc + 128 + NWV * 32;      // [NWV][32] per-wave counts of positive outputs
  float* s_xn = misc - 256;                 // [256] normalised value of every uint8 level (own 256-float block)

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const bool stash = p.st_a1 != nullptr;

  // ---- one-time: load weights into MFMA B fragments, zero LDS (halos).  W2 and W3 go through LDS (coalesced 16-byte loads, then
  // 90 ds_read_b32 per lane): read straight from global memory the 90 strided 4-byte loads per lane were most of the 30 k cycles
  // a workgroup spent in front of its frame loop (14 us of a 470 us launch, stage timers)
  static_assert(LL::total >= 1152 + 3456, "weight staging area");
  if (((reinterpret_cast<uintptr_t>(p.w2) | reinterpret_cast<uintptr_t>(p.w3)) & 15) == 0) {  // (views of a flat bucket need not be)
    const f32x4* g2 = reinterpret_cast<const f32x4*>(p.w2);
    const f32x4* g3 = reinterpret_cast<const f32x4*>(p.w3);
    for (int q = tid; q < 1152 / 4; q += NT) reinterpret_cast<f32x4*>(lds)[q] = g2[q];
    for (int q = tid; q < 3456 / 4; q += NT) reinterpret_cast<f32x4*>(lds + 1152)[q] = g3[q];
  } else {
    for (int q = tid; q < 1152; q += NT) lds[q] = p.w2[q];
    for (int q = tid; q < 3456; q += NT) lds[1152 + q] = p.w3[q];
  }
  __syncthreads();
  const float* w2s = lds;
  const float* w3s = lds + 1152;
  // conv1: k = 4kk+g -> (ry = k/3, kx = k%3) over the 4x3 input window of output rows y, y+1;
  // column i = (c = i&7, s = i>>3): W1[c][ky = ry - s][kx], zero outside the 3x3 kernel
#if SS_CONV1_LOCAL
  // lane-local pool windows: M = the 16 pixels of four 2 x 2 windows (D rows 4g + r = position r of window g: a lane ends up with
  // ONE whole window), N = (8 channels) x (row pairs yq and yq + 1 of a 4-row band), K = the 5 x 3 input window both row pairs
  // touch = 15 -> 16: k = 4kk+g -> (ry = k/3, kx = k%3); column i = (c = i&7, v = i>>3): W1[c][ky = ry - 2v][kx], zero outside
  constexpr int K1S = 4;
  float bw1[K1S];
  int aoff1[K1S];
#pragma unroll
  for (int kk = 0; kk < K1S; ++kk) {
    const int k = 4 * kk + g, ry = k / 3, kx = k % 3, c = i & 7, v = i >> 3, ky = ry - 2 * v;
    bw1[kk] = (k < 15 && ky >= 0 && ky <= 2) ? p.w1[c * 9 + ky * 3 + kx] : 0.f;
    aoff1[kk] = k < 15 ? ry * XS + kx : 0;
  }
#else
  float bw1[3];
  int aoff1[3];
#pragma unroll
  for (int kk = 0; kk < 3; ++kk) {
    const int k = 4 * kk + g, ry = k / 3, kx = k % 3, c = i & 7, s = i >> 3, ky = ry - s;
    bw1[kk] = (ky >= 0 && ky <= 2) ? p.w1[c * 9 + ky * 3 + kx] : 0.f;
    aoff1[kk] = ry * XS + kx;
  }
#endif
  float bw2[18];  // conv2: k-step kk -> tap = kk/2, c = 4*(kk%2)+g ; n = i
#pragma unroll
  for (int kk = 0; kk < 18; ++kk) bw2[kk] = w2s[i * 72 + (4 * (kk & 1) + g) * 9 + (kk >> 1)];
  // conv3: 24 output channels on 16-wide tiles.  Channels 0..15 fill one tile (k-step kk: tap = kk/4, c = 4*(kk%4)+g; n = i).
  // Channels 16..23 would waste half of a second one (round 3: 25 % of conv3's MFMAs multiplied zero columns), so they share a
  // tile between the pixel rows y and y+1 -- column i = (channel 16 + (i&7), row y + (i>>3)) -- over the 4 x 3 window both rows
  // touch, conv1's trick: K = 4 rows x 3 columns x 16 channels = 192 with zero weights where a row lies outside its 3 x 3
  // kernel.  120 MFMAs per 32 pixels instead of 144, and the 48 A values feed all three accumulators.
  float bw3a[36], bw3b[48];
#pragma unroll
  for (int kk = 0; kk < 36; ++kk) {
    const int c = 4 * (kk & 3) + g, tap = kk >> 2;
    bw3a[kk] = w3s[i * 144 + c * 9 + tap];
  }
#pragma unroll
  for (int kk = 0; kk < 48; ++kk) {  // kk = (ry * 3 + kx) * 4 + cg
    const int c = 4 * (kk & 3) + g, ry = (kk >> 2) / 3, kx = (kk >> 2) % 3, ky = ry - (i >> 3);
    bw3b[kk] = (ky >= 0 && ky <= 2) ? w3s[(16 + (i & 7)) * 144 + c * 9 + ky * 3 + kx] : 0.f;
  }
  __syncthreads();
  for (int q = tid; q < LL::total; q += NT) lds[q] = 0.f;
  __syncthreads();
  if (tid < 8) s_b1[tid] = p.b1[tid];
  if (tid < 16) s_b2[tid] = p.b2[tid];
  if (tid < 24) s_b3[tid] = p.b3[tid];

  const float level_f = (float)(tid & 255) / 255.0f;  // fl(u / 255) of this thread's grey level: the same for every frame
  uint4 px[NCH];
  auto load_frame = [&](int n) {
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int q = tid + k * NT;
      if (q * 16 < HW) px[k] = reinterpret_cast<const uint4*>(p.R + (long)n * HW)[q];
    }
  };
  // the frames this launch walks: all N, or the listed ones (rows of a clip's padding never reach the packed recurrence)
  const int n_walk = p.frames ? min(p.frames[0], p.N) : p.N;
  auto frame_at = [&](int it) { return p.frames ? (int)min((unsigned)p.frames[1 + it], (unsigned)(p.N - 1)) : it; };
  // frame numbers are read two frames ahead (a scalar load that misses would otherwise be waited for inside a stage)
  int n = (int)blockIdx.x < n_walk ? frame_at(blockIdx.x) : -1;
  int n_next = (int)(blockIdx.x + gridDim.x) < n_walk ? frame_at(blockIdx.x + gridDim.x) : -1, n_after = -1;
  if (n >= 0) load_frame(n);
  // (Round 3 experiment: the second workgroup of every CU started 16 k / 32 k / 49 k cycles late, so that its vector-heavy stages
  // -- statistics, conv1 -- would meet the first one's MFMA-heavy ones -- conv2, conv3 -- instead of its own kind: 471.8 / 469.7 /
  // 473.2 us per launch against 462.6, i.e. the sleep itself and nothing else.  The two workgroups of a CU do not get in each
  // other's way whatever their phase: each is bound by its own chain of reads, MFMAs and barriers, not by a shared pipe.)
  STAMP_DECL;

  for (int it = blockIdx.x; it < n_walk; it += gridDim.x, n = n_next, n_next = n_after) {
    const int it2 = it + 2 * (int)gridDim.x;
    n_after = it2 < n_walk ? frame_at(it2) : -1;
    STAMP(15);
    // ---------------- stage 0: statistics + normalise
    unsigned su = 0, sq = 0;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int q = tid + k * NT;
      if (q * 16 < HW) {
        // four pixels per instruction: v_dot4_u32_u8 (exact integer sums; a 64x64 frame's sum of squares is < 2^29)
        const unsigned wds[4] = {px[k].x, px[k].y, px[k].z, px[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          su = __builtin_amdgcn_udot4(wds[e], 0x01010101u, su, false);
          sq = __builtin_amdgcn_udot4(wds[e], wds[e], sq, false);
        }
      }
    }
    su = wave_sum_u32(su);
    sq = wave_sum_u32(sq);
    if (lane == 0) { s_red[2 * wv] = su; s_red[2 * wv + 1] = sq; }
    __syncthreads();
    if (tid == 0) {
      unsigned long long tsu = 0, tsq = 0;
      for (int k = 0; k < NWV; ++k) { tsu += s_red[2 * k]; tsq += s_red[2 * k + 1]; }
      float mu = 0.f, sd = 1.f;
      if (p.standardize) {
        // exact integer sums -> mean and unbiased variance in double.  Products with the reciprocals of the (compile-time) pixel
        // counts instead of f64 divides, one f32 root instead of an f64 one: the serial section every other wave of the
        // workgroup waits for was 133 vector instructions, and its result is rounded to f32 anyway (the reference computes
        // mean and std in f32 altogether).  For a constant frame mu == fl(u / 255) still holds exactly -- tsu / HW is u to
        // within 1e-16, so the f32 rounding returns u -- and xn == 0.
        constexpr double inv_n = 1.0 / (double)HW, inv_n1 = 1.0 / ((double)HW - 1.0);
        const double mean_u = (double)tsu * inv_n;
        mu = (float)mean_u / 255.0f;
        const double var = ((double)tsq - (double)tsu * mean_u) * inv_n1;
        sd = sqrtf((float)(var > 0.0 ? var : 0.0)) / 255.0f;
        sd = fmaxf(sd, 1e-6f);
      }
      s_stat[0] = mu;
      s_stat[1] = sd;
      if (stash) {
        p.st_feat[(long)n * ST_FEAT + 48] = mu;
        p.st_feat[(long)n * ST_FEAT + 49] = sd;
      }
    }
    __syncthreads();
    if (tid < 256) s_xn[tid] = p.standardize ? (level_f - s_stat[0]) / s_stat[1] : level_f;  // one IEEE divide per grey level
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int q = tid + k * NT;
      if (q * 16 < HW) {
        const int lin = q * 16;
        float* dst = xh + (lin / W + 1) * XS + (lin % W + 1);
        const unsigned wds[4] = {px[k].x, px[k].y, px[k].z, px[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int b = 0; b < 4; ++b) dst[4 * e + b] = s_xn[(wds[e] >> (8 * b)) & 255u];
      }
    }
    // the zero border of the frame: region U held the previous frame's pooled-2 map
    for (int q = tid; q < 2 * XS + 2 * H; q += NT) {
      int cell;
      if (q < XS) cell = q;                                   // top row
      else if (q < 2 * XS) cell = (H + 1) * XS + (q - XS);    // bottom row
      else if (q < 2 * XS + H) cell = (q - 2 * XS + 1) * XS;  // left column
      else cell = (q - 2 * XS - H + 1) * XS + W + 1;          // right column
      xh[cell] = 0.f;
    }
    // prefetch the next frame's bytes while this one is computed
    if (n_next >= 0) load_frame(n_next);
    __syncthreads();
    STAMP(0);
    STAGE_END(0);

    // ---------------- stage 1: conv1 (MFMA) + ReLU + pool -> a1 (haloed), argmax bytes.
    // The pooling epilogue is the stage's cost (the forward kernel issues 3.3 other vector instructions per MFMA, most of
    // them here, and they do not hide behind the MFMAs of the CU's other workgroup): it is written for instruction count.
    //  * a pass = UC neighbouring 16-pixel blocks of one row pair: one address per pass, immediates per chain;
    //  * max first, bias and ReLU once on the winner (max(relu(x_i + b)) = relu(max(x_i + b)));
    //  * the argmax bits are lane MASKS (v_cmp writes an SGPR pair): the row / column bookkeeping and the exchange with
    //    the partner lane (i ^ 8: a byte swap of the mask) run on the scalar unit, two vector instructions turn the
    //    final masks into the index byte;
    //  * the MFMAs of the next pass are issued before the epilogue of this one (two accumulator sets).
    {
#if SS_CONV1_LOCAL
      // Round 4: the 2 x 2 max-pool inside ONE lane.  The row-pair form below leaves a lane with two half windows: column pairs in
      // the lane, the row pair in lane i ^ 8 -- a DPP exchange, two selects and the partner's argmax bits swapped on the scalar unit:
      // 16 vector + 16 scalar instructions per pooled output, and conv1 was 22 k of a frame's 64 k cycles for 3 k cycles of MFMAs
      // (profiles/round4 stage timers; f32 MFMAs and the other vector instructions of a SIMD do not overlap).  Here a lane's four
      // accumulator values ARE one window (M tile = four windows x four positions, N = 8 channels x two row pairs of a 4-row band,
      // K = 5 x 3 -> 16): 4 MFMAs per 64 outputs instead of 3, and the epilogue is 8 vector + 3 scalar instructions -- the bias rides
      // in the accumulator's initial value, v_max3 folds the ReLU into the last maximum.
      constexpr int UC = 4;                       // windows groups per pass: 32 frame columns
      constexpr int XP = W / 32, passes = (H / 4) * XP;
      static_assert(W % 32 == 0 && H % 4 == 0, "conv1 pass split");
      const int c = i & 7, v = i >> 3;
      const int wvu = __builtin_amdgcn_readfirstlane(wv);
      const float bias = s_b1[c];
      const float* xa = xh + ((i >> 1) & 1) * XS + 2 * (i >> 2) + (i & 1);  // A row i = position i & 3 of window i >> 2
      float* a1w = a1 + c * P1 + (v + 1) * S1 + g + 1;                      // D: window g of channel c, pooled row 2 yq + v
      uint8_t* i1w = i1s + c * G::I1S + v * W2 + g;
      auto mm = [&](int ps, f32x4 (&acc)[UC]) {
        const float* ap = xa + (4 * (ps / XP)) * XS + 32 * (ps % XP);
        float av[UC][K1S];
#pragma unroll
        for (int kk = 0; kk < K1S; ++kk)
#pragma unroll
          for (int q = 0; q < UC; ++q) av[q][kk] = (ap + aoff1[kk])[8 * q];
        SS_SCHED_FENCE();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q = 0; q < UC; ++q) acc[q] = f32x4{bias, bias, bias, bias};
#pragma unroll
        for (int kk = 0; kk < K1S; ++kk)
#pragma unroll
          for (int q = 0; q < UC; ++q) acc[q] = mfma16(av[q][kk], bw1[kk], acc[q]);
        __builtin_amdgcn_s_setprio(0);
        SS_SCHED_FENCE();
      };
      auto epi = [&](int ps, const f32x4 (&acc)[UC]) {
        const int yq = ps / XP, xb = ps % XP;
        float* aw = a1w + 2 * yq * S1 + 16 * xb;
        uint8_t* iw = i1w + 2 * yq * W2 + 16 * xb;
#pragma unroll
        for (int q = 0; q < UC; ++q) {
          const float x00 = acc[q][0], x01 = acc[q][1], x10 = acc[q][2], x11 = acc[q][3];  // (row, column) of the window
          const unsigned long long c0 = __ballot(x01 > x00), c1 = __ballot(x11 > x10);
          const float m0 = fmaxf(x00, x01), m1 = fmaxf(x10, x11);
          const unsigned long long t1 = __ballot(m1 > m0);  // strictly: ties go to the first in row-major order
          const unsigned long long b0 = (t1 & c1) | (~t1 & c0);
          const float best = fmaxf(fmaxf(m0, m1), 0.f);
          int hi, bi;
          unsigned long long carry_out;
          asm("v_cndmask_b32_e64 %0, 0, 2, %1" : "=v"(hi) : "s"(t1));
          asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(bi), "=s"(carry_out) : "v"(hi), "s"(b0));
          aw[4 * q] = best;
          iw[4 * q] = (uint8_t)bi;
        }
      };
#else
      constexpr int XT = W / 16;
      constexpr int UC = (XT % 4 == 0) ? 4 : (XT % 3 == 0) ? 3 : 2;
      static_assert(XT % UC == 0, "conv1 pass split");
      constexpr int XP = XT / UC, passes = (H / 2) * XP;
      const int c = i & 7, s = i >> 3;
      const int wvu = __builtin_amdgcn_readfirstlane(wv);
      const float bias = s_b1[c];
      constexpr unsigned long long S = 0xFF00FF00FF00FF00ull, LO = 0x00FF00FF00FF00FFull;  // lanes of row s = 1; low bytes
      const float* xa = xh + i;
      float* a1w = a1 + c * P1 + S1 + 2 * g + s + 1;
      uint8_t* i1w = i1s + c * G::I1S + 2 * g + s;
      auto mm = [&](int ps, f32x4 (&acc)[UC]) {
        const float* ap = xa + (2 * (ps / XP)) * XS + 16 * UC * (ps % XP);
        float av[UC][3];
#pragma unroll
        for (int kk = 0; kk < 3; ++kk)
#pragma unroll
          for (int q = 0; q < UC; ++q) av[q][kk] = (ap + aoff1[kk])[16 * q];
        SS_SCHED_FENCE();
        __builtin_amdgcn_s_setprio(1);  // conv1's twelve MFMAs ahead of the partner's vector work, its own epilogue not
#pragma unroll
        for (int q = 0; q < UC; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 3; ++kk)
#pragma unroll
          for (int q = 0; q < UC; ++q) acc[q] = mfma16(av[q][kk], bw1[kk], acc[q]);
        __builtin_amdgcn_s_setprio(0);
        SS_SCHED_FENCE();
      };
      auto epi = [&](int ps, const f32x4 (&acc)[UC]) {
        const int yp = ps / XP, xt0 = UC * (ps % XP);
        float* aw = a1w + yp * S1 + 8 * xt0;
        uint8_t* iw = i1w + yp * W2 + 8 * xt0;
#pragma unroll
        for (int q = 0; q < UC; ++q) {
          // lane holds 4 pixels x0+4g+r of row 2yp+s for channel c: column pairs (0,1) and (2,3)
          const float x0 = acc[q][0] + bias, x1 = acc[q][1] + bias, x2 = acc[q][2] + bias, x3 = acc[q][3] + bias;
          const unsigned long long m0 = __ballot(x1 > x0), m1 = __ballot(x3 > x2);
          const float p0 = fmaxf(x0, x1), p1 = fmaxf(x2, x3);
          // row s = 0 lanes finish column pair 0, row s = 1 lanes finish column pair 1; the partner (other row, same
          // channel) is lane ^ 8: a rotate by 8 inside the 16-lane DPP row, no LDS round trip
          const float own = s ? p1 : p0, send = s ? p0 : p1;
          const float recv = SS_DPP_F(send, 0x128);
          const unsigned long long oc = (m1 & S) | (m0 & ~S);   // column bit of the own pair
          const unsigned long long sc = (m0 & S) | (m1 & ~S);   // column bit of the pair the partner finishes
          const unsigned long long rc = ((sc & LO) << 8) | ((sc >> 8) & LO);  // the partner's bit for the own pair
          const unsigned long long rg = __ballot(recv > own), og = __ballot(own > recv);
          const unsigned long long t1 = (rg & ~S) | (og & S);   // row 1 wins (strictly: ties go to the first in row-major order)
          const unsigned long long r1c = (oc & S) | (rc & ~S), r0c = (rc & S) | (oc & ~S);
          const unsigned long long b0 = (t1 & r1c) | (~t1 & r0c);
          const float best = fmaxf(fmaxf(own, recv), 0.f);
          int hi, bi;
          unsigned long long carry_out;
          asm("v_cndmask_b32_e64 %0, 0, 2, %1" : "=v"(hi) : "s"(t1));
          asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(bi), "=s"(carry_out) : "v"(hi), "s"(b0));
          aw[8 * q] = best;
          iw[8 * q] = (uint8_t)bi;
        }
      };
#endif
      f32x4 accA[UC], accB[UC];
      if (wvu < passes) {
        mm(wvu, accA);
#pragma unroll 1
        for (int ps = wvu;;) {  // (wave-uniform exits: a wave's share of the passes need not be even)
          const int p1 = ps + NWV, p2 = p1 + NWV;
          if (p1 < passes) mm(p1, accB);
          epi(ps, accA);
          if (p1 >= passes) break;
          if (p2 < passes) mm(p2, accA);
          epi(p1, accB);
          if (p2 >= passes) break;
          ps = p2;
        }
      }
    }
    __syncthreads();
    STAMP(1);
    STAGE_END(1);
    if (stash) {  // pool-1 argmaxes first: conv2 is about to write the pooled-2 map over them
      for (int q = tid; q < G::I1S / 2; q += NT)
        reinterpret_cast<uint4*>(p.st_i1 + (long)n * 8 * G::I1S)[q] = reinterpret_cast<const uint4*>(i1s)[q];
    }
    __syncthreads();
    // the zero border of the pooled-2 planes (region U held the frame and the pool-1 argmaxes); conv2 writes interiors only
    for (int q = tid; q < 16 * (2 * S2 + 2 * G::H4); q += NT) {
      constexpr int per = 2 * S2 + 2 * G::H4;
      const int pl = q / per, r = q % per;
      int cell;
      if (r < S2) cell = r;
      else if (r < 2 * S2) cell = (G::H4 + 1) * S2 + (r - S2);
      else if (r < 2 * S2 + G::H4) cell = (r - 2 * S2 + 1) * S2;
      else cell = (r - 2 * S2 - G::H4 + 1) * S2 + W4 + 1;
      a2[pl * P2 + cell] = 0.f;
    }
    if (stash) {  // pooled-1 map (haloed image as it stands: the backward loads it linearly)
      f32x4* dst = reinterpret_cast<f32x4*>(p.st_a1 + (long)n * 8 * P1);
      for (int q = tid; q < 2 * P1; q += NT) dst[q] = reinterpret_cast<const f32x4*>(a1)[q];
    }

    // ---------------- stage 2: conv2 (MFMA) + ReLU + pool -> a2 (haloed)
    {
      constexpr int XT = W2 / 16;
      constexpr int units = (H2 / 2) * XT;
      const float bias = s_b2[i];
      // The CU's other workgroup is usually in a different stage: while this wave streams MFMAs its issue slots must not
      // go to the partner's vector instructions (each holds the SIMD's issue port for 4+ cycles and the in-order wave
      // then misses its MFMA slot).  Raised priority through the MFMA-dense stages, the default for epilogue-heavy conv1,
      // the statistics and the copies: 472 -> 462 us per launch, the largest single gain of the round for two lines.
      __builtin_amdgcn_s_setprio(2);
      for (int u = wv; u < units; u += NWV) {
        const int yp = u / XT, xt = u % XT;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        const float* base = a1 + g * P1 + (2 * yp) * S1 + 16 * xt + i;
#pragma unroll
        for (int kb = 0; kb < 18; kb += 6) {
          float a0[6], a1v[6];
#pragma unroll
          for (int u = 0; u < 6; ++u) {
            const int kk = kb + u, tap = kk >> 1, ky = tap / 3, kx = tap % 3;
            const float* ap = base + 4 * (kk & 1) * P1 + ky * S1 + kx;
            a0[u] = ap[0];
            a1v[u] = ap[S1];
          }
          SS_SCHED_FENCE();
#pragma unroll
          for (int u = 0; u < 6; ++u) {
            acc0 = mfma16(a0[u], bw2[kb + u], acc0);
            acc1 = mfma16(a1v[u], bw2[kb + u], acc1);
          }
          SS_SCHED_FENCE();
        }
        // pooling epilogue, written for instruction count like conv1's: the window's four values sit in one lane
        // (acc0 = row 2yp, acc1 = row 2yp + 1; column pairs 2e, 2e + 1), max first, ReLU once, argmax bits as lane masks
        float* aw = a2 + i * P2 + (yp + 1) * S2 + 8 * xt + 2 * g + 1;
        uint8_t* iw = i2s + (yp * W4 + 8 * xt + 2 * g) * 16 + i;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float x00 = acc0[2 * e] + bias, x01 = acc0[2 * e + 1] + bias;
          const float x10 = acc1[2 * e] + bias, x11 = acc1[2 * e + 1] + bias;
          const unsigned long long c0 = __ballot(x01 > x00), c1 = __ballot(x11 > x10);
          const float m0 = fmaxf(x00, x01), m1 = fmaxf(x10, x11);
          const unsigned long long t1 = __ballot(m1 > m0);  // strictly: ties go to the first in row-major order
          const unsigned long long b0 = (t1 & c1) | (~t1 & c0);
          const float best = fmaxf(fmaxf(m0, m1), 0.f);
          int hi, bi;
          unsigned long long carry_out;
          asm("v_cndmask_b32_e64 %0, 0, 2, %1" : "=v"(hi) : "s"(t1));
          asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(bi), "=s"(carry_out) : "v"(hi), "s"(b0));
          aw[e] = best;
          iw[16 * e] = (uint8_t)bi;
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    STAMP(2);
    STAGE_END(2);
    if (stash) {
      f32x4* dst = reinterpret_cast<f32x4*>(p.st_a2 + (long)n * 16 * P2);
      for (int q = tid; q < 4 * P2; q += NT) dst[q] = reinterpret_cast<const f32x4*>(a2)[q];
      for (int q = tid; q < P; q += NT)
        reinterpret_cast<uint4*>(p.st_i2 + (long)n * 16 * P)[q] = reinterpret_cast<const uint4*>(i2s)[q];
    }
    STAMP(3);

    // ---------------- stage 3: conv3 (MFMA) + ReLU + global average.  A unit = 16 pixels of an even row y (linear over the even
    // rows) and their neighbours in row y + 1: accumulators a0 / a1 = channels 0..15 of the two rows, ab = channels 16..23 of both.
    {
      constexpr int units = P / 32;
      float fa = 0.f, fb = 0.f;  // per-lane partial channel sums (a: channel i; b: channel 16 + (i&7), row y + (i>>3))
      float ca = 0.f, cb = 0.f;  // and counts of positive outputs
      const int c8 = i & 7, sB = i >> 3;
      const float bias_a = s_b3[i], bias_b = s_b3[16 + c8];
      __builtin_amdgcn_s_setprio(2);
      for (int u = wv; u < units; u += NWV) {
        const int qa = 16 * u + i;  // this lane's A row: pixel (2 (qa / W4), qa % W4)
        const float* base = a2 + g * P2 + (2 * (qa / W4)) * S2 + (qa % W4);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, accb = acc0;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          float av[4][4];  // [ry][cg]
#pragma unroll
          for (int ry = 0; ry < 4; ++ry)
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) av[ry][cg] = base[4 * cg * P2 + ry * S2 + kx];
          SS_SCHED_FENCE();
#pragma unroll
          for (int cg = 0; cg < 4; ++cg)
#pragma unroll
            for (int ry = 0; ry < 4; ++ry) {
              if (ry < 3) acc0 = mfma16(av[ry][cg], bw3a[(ry * 3 + kx) * 4 + cg], acc0);
              if (ry > 0) acc1 = mfma16(av[ry][cg], bw3a[((ry - 1) * 3 + kx) * 4 + cg], acc1);
              accb = mfma16(av[ry][cg], bw3b[(ry * 3 + kx) * 4 + cg], accb);
            }
          SS_SCHED_FENCE();
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v0 = acc0[r] + bias_a, v1 = acc1[r] + bias_a, vb = accb[r] + bias_b;
          const bool p0 = v0 > 0.f, p1 = v1 > 0.f, pb = vb > 0.f;
          fa += fmaxf(v0, 0.f) + fmaxf(v1, 0.f);  // (one v_max + v_add instead of compare, select, add)
          fb += fmaxf(vb, 0.f);
          ca += (p0 ? 1.f : 0.f) + (p1 ? 1.f : 0.f);
          cb += pb ? 1.f : 0.f;
          if (stash) {  // D row 4g+r = pixel, column i = channel: pixel-major bytes, 32 per pixel (24..31 zero)
            const int qd = 16 * u + 4 * g + r;
            uint8_t* mp = m3s + ((2 * (qd / W4)) * W4 + qd % W4) * 32;
            mp[i] = p0;
            mp[W4 * 32 + i] = p1;
            mp[sB * W4 * 32 + 16 + c8] = pb;
            mp[sB * W4 * 32 + 24 + c8] = 0;
          }
        }
      }
      __builtin_amdgcn_s_setprio(0);
      // reduce over the 4 lane groups (rows of the tiles), the two rows of the shared tile, then over waves through LDS
      fa += __shfl_xor(fa, 16, 64); fa += __shfl_xor(fa, 32, 64);
      fb += __shfl_xor(fb, 16, 64); fb += __shfl_xor(fb, 32, 64); fb += __shfl_xor(fb, 8, 64);
      ca += __shfl_xor(ca, 16, 64); ca += __shfl_xor(ca, 32, 64);
      cb += __shfl_xor(cb, 16, 64); cb += __shfl_xor(cb, 32, 64); cb += __shfl_xor(cb, 8, 64);
      if (g == 0) {
        s_fp[wv * 32 + i] = fa; s_cp[wv * 32 + i] = ca;
        s_fp[wv * 32 + 16 + i] = i < 8 ? fb : 0.f; s_cp[wv * 32 + 16 + i] = i < 8 ? cb : 0.f;
      }
    }
    __syncthreads();
    STAMP(4);
    STAGE_END(4);
    if (stash)
      for (int q = tid; q < 2 * P; q += NT)
        reinterpret_cast<uint4*>(p.st_m3 + (long)n * 32 * P)[q] = reinterpret_cast<const uint4*>(m3s)[q];
    if (tid < 24) {
      float s = 0.f;
      for (int k = 0; k < NWV; ++k) s += s_fp[k * 32 + tid];
      s /= (float)P;
      s_feat[tid] = s;
      if (stash) {
        float cnt = 0.f;
        for (int k = 0; k < NWV; ++k) cnt += s_cp[k * 32 + tid];
        p.st_feat[(long)n * ST_FEAT + tid] = s;
        p.st_feat[(long)n * ST_FEAT + 24 + tid] = cnt;
      }
    }
    __syncthreads();
    // ---------------- stage 4: fc
    if (tid < p.E) {
      float o = p.bfc[tid];
      for (int c = 0; c < 24; ++c) o += s_feat[c] * p.wfc[tid * 24 + c];
      p.out[(long)n * p.ld_out + tid] = o;
    }
    STAMP(5);
  }
  STAMP_FLUSH();
}

template <class G>
int launch_fwd(const CnnFwdParams& p, hipStream_t s) {
  constexpr size_t lds_bytes = (size_t)FwdLds<G>::total * sizeof(float);
  static_assert(lds_bytes <= 80 * 1024, "two workgroups per CU: the LDS image must stay under half a CU");
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(roi_cnn_fwd_kernel<G>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            80 * 1024) != hipSuccess)
      return SS_ERR_LAUNCH;
    attr_set = true;
  }
  const int cap = 2 * (ss_cnn_max_wgs > 0 ? ss_cnn_max_wgs : ss_device_cus());  // the cap counts CUs
  const int grid = p.N < cap ? p.N : cap;
  hipLaunchKernelGGL(roi_cnn_fwd_kernel<G>, dim3(grid), dim3(NT), lds_bytes, s, p);
  return ss_launch_status();
}

}  // namespace

extern "C" int ss_roi_cnn_fwd_frames(const uint8_t* R, int N, int H, int W, int standardize, const float* w1,
                                     const float* b1, const float* w2, const float* b2, const float* w3,
                                     const float* b3, const float* wfc, const float* bfc, int E, float* out,
                                     int ld_out, float* st_a1, uint8_t* st_i1, float* st_a2, uint8_t* st_i2,
                                     uint8_t* st_m3, float* st_feat, const int* stash_sizes, const int* frames,
                                     ss_stream_t stream) {
  SS_REQUIRE(R && w1 && b1 && w2 && b2 && w3 && b3 && wfc && bfc && out, SS_ERR_ARG);
  SS_REQUIRE(N > 0 && E > 0 && ld_out >= E, SS_ERR_ARG);
  SS_REQUIRE(E <= 64, SS_ERR_UNSUPPORTED);
  const bool any = st_a1 || st_i1 || st_a2 || st_i2 || st_m3 || st_feat;
  const bool all = st_a1 && st_i1 && st_a2 && st_i2 && st_m3 && st_feat;
  SS_REQUIRE(!any || (all && stash_sizes), SS_ERR_ARG);
  CnnFwdParams p;
  p.R = R; p.N = N; p.standardize = standardize;
  p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.w3 = w3; p.b3 = b3; p.wfc = wfc; p.bfc = bfc;
  p.E = E; p.out = out; p.ld_out = ld_out;
  p.st_a1 = st_a1; p.st_i1 = st_i1; p.st_a2 = st_a2; p.st_i2 = st_i2; p.st_m3 = st_m3; p.st_feat = st_feat;
  p.frames = frames;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define SS_DISPATCH(HH, WW)                                                                                      \
  if (H == HH && W == WW) {                                                                                     \
    using G_ = Geom<HH, WW>;                                                                                    \
    SS_REQUIRE(!all || stash_sizes_match<G_>(stash_sizes), SS_ERR_ARG);                                         \
    return launch_fwd<G_>(p, s);                                                                                \
  }
  SS_CNN_SHAPES(SS_DISPATCH)
#undef SS_DISPATCH
  return SS_ERR_UNSUPPORTED;
}

extern "C" int ss_roi_cnn_fwd_stash(const uint8_t* R, int N, int H, int W, int standardize, const float* w1,
                                    const float* b1, const float* w2, const float* b2, const float* w3,
                                    const float* b3, const float* wfc, const float* bfc, int E, float* out,
                                    int ld_out, float* st_a1, uint8_t* st_i1, float* st_a2, uint8_t* st_i2,
                                    uint8_t* st_m3, float* st_feat, const int* stash_sizes,
                                    ss_stream_t stream) {
  return ss_roi_cnn_fwd_frames(R, N, H, W, standardize, w1, b1, w2, b2, w3, b3, wfc, bfc, E, out, ld_out, st_a1, st_i1, st_a2,
                               st_i2, st_m3, st_feat, stash_sizes, nullptr, stream);
}

extern "C" int ss_roi_cnn_stash_size(int H, int W, int* sizes) {
  SS_REQUIRE(sizes, SS_ERR_ARG);
#define SS_DISPATCH(HH, WW)               \
  if (H == HH && W == WW) {               \
    stash_sizes_of<Geom<HH, WW>>(sizes);  \
    return SS_OK;                         \
  }
  SS_CNN_SHAPES(SS_DISPATCH)
#undef SS_DISPATCH
  return SS_ERR_UNSUPPORTED;
}

extern "C" int ss_roi_cnn_fwd(const uint8_t* R, int N, int H, int W, int standardize, const float* w1,
                              const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                              const float* wfc, const float* bfc, int E, float* out, int ld_out,
                              ss_stream_t stream) {
  return ss_roi_cnn_fwd_stash(R, N, H, W, standardize, w1, b1, w2, b2, w3, b3, wfc, bfc, E, out, ld_out, nullptr,
                              nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}
