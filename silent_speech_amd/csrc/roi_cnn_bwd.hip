// Backward of the fused ROI normalise + TinyROICNN block w.r.t. its eight parameter tensors
// (the uint8 image has no gradient).  Autograd counterpart of
// /root/reference/train_model_official.py:212-229 as driven by loss.backward() (:437).
//
// One persistent 512-thread workgroup per CU walks frames.  Per frame it reads the forward's stash
// (pooled maps a1, a2, pool argmaxes, conv3 sign mask, averaged features: 67 KB for 64x64) plus the
// 4 KB uint8 frame, and runs four MFMA contractions and one VALU gather out of LDS:
//
//   S1  dW3[n][c][tap] += sum_q dy3[n][q-tap] * a2[c][q]       M=(tap,n)=216->224 N=c(16) K=pixels: the tap shift sits in
//                                                              the A operand, so 9 x 24 rows pack into 14 tiles (18 with M=n)
//   S2  da2[c][p] = sum_{n,tap} dy3[n][p-tap] * W3[n][c][tap]  M=pixels N=c(16) K=(tap,n)=216
//   T   dy2 (gradient before max-pool 2) expanded ONCE into a dense LDS image from da2 + 2-bit argmax
//   S3  dW2[n][c][tap] += sum_x dy2[n][x] * a1[c][x+tap]       M=n(16) N=(tap,c)=72->80 K=pixels
//   S4  da1[c][y..y+1][x] = sum dy2[n][.] * W2[n][c][.]        M=16 pixels of row y, N=(c, row y|y+1)=16,
//                                                              K=(4 source rows x 3 cols, n)=192: no N padding
//   S5  G[(c,o)][(ry,rx)] += sum_q [argmax1(c,q)=o] da1[c][q] * x[2q+(ry,rx)-1]   M=(8 ch x 4 window slots)=32,
//       N=16 shifted views of the normalised frame, K=pooled pixels; dW1[c][ky][kx] = sum_o G[(c,o)][o+(ky,kx)]
//       (the pool routes each gradient to ONE of 4 input positions, so conv1's weight gradient is a 32x16xK GEMM
//       on the pooled grid instead of an 8x9x4K one); x is rebuilt in LDS from a 256-entry grey-level table
//
// Weight-gradient partial sums stay in registers for the whole frame walk (K = pixels split over the
// 8 waves) and are reduced through LDS, then one float atomic per element per workgroup, at the end.
//
// Frame schedule (a frame starts at S1; seven workgroup barriers):
//   [E] pixels / pool-2 argmaxes out of staging | S1 (+ the pooled-1 map's DMA, + the NEXT frame's d_out row) | barrier |
//   S2 | [D] phase switch T: the dy2 scatter (two barriers) | S3 (+ per wave, without a barrier, the two waves of a SIMD
//   at different passes: the normalised frame, the pool-1 argmax bytes, dy2's zero columns, d feat of the next frame) |
//   barrier | S4 | barrier | S5 (+ the next frame's inputs by DMA with its first pass, + the next frame's front -- dy3
//   image, grey-level table -- per wave in the same way) | [E]
// Every MFMA stage issues its operand reads one pass ahead of the MFMAs: the two waves of a SIMD leave each barrier
// together and run in lockstep, so a wave's LDS latency is not covered by its partner.
#include <type_traits>
#include "ss_common.h"
#include "roi_cnn_geom.h"

STAMP_TABLE(ss_debug_stamps_bwd)

extern int ss_cnn_max_wgs;  // roi_cnn.hip

namespace {

constexpr int NT = 512;
constexpr int NWV = NT / 64;

template <class G>
struct BwdLds {
  static constexpr int o_a1h = 0;                         // [8][P1]   haloed pooled-1 map; S4 overwrites it with da1
  // phase area.  phase 1: a2h | dy3 (pixel-major [haloed pixel][24 ch]) | (free: second staging area) | i2 (| w3) ;
  // phase 2: dy2 | xh | i1
  static constexpr int o_ph = 8 * G::P1;
  static constexpr int o_dy3h = o_ph + 16 * G::P2;
  // dy3 is pixel-major, pixel stride DS floats.  S2 reads the 16 pixels of a row tile with one ds_read_b128 (channels 4g .. 4g+3)
  // and one ds_read_b64 (channels 16+2g, 17+2g) per lane.  The hardware serves a b128 read in groups of 16 lanes that are pixels
  // 0-3 and 12-15 at chunk g with pixels 4-11 at chunk g + 1 (MI355X_MICROARCH.md, LDS table): conflict-free iff DS/4 == 2 (mod 4),
  // i.e. 24 floats, no padding (round 2 had reasoned with 16 consecutive lanes and padded to 28: 2-way on the b128 reads,
  // conflict-free on the b64 ones; A/B on one box: 716.6 -> 715.0 us per launch -- S2 is not LDS-bound any more either way)
  static constexpr int DS = 24;
  static constexpr int o_da2m = o_dy3h + DS * G::P2;
  static constexpr int o_i2b = o_da2m + 16 * G::P;
  static constexpr int end1a = (o_i2b + 4 * G::P + 3) & ~3;
  // dy2 is pixel-major with a zero column left and right of every row: [H2][W2 + 2][16 channels].  A lane's four
  // k-steps of an S4 tap are then ONE ds_read_b128 at (lane base + immediate): no address arithmetic, no edge selects
  static constexpr int W2H = G::W2 + 2;
  static constexpr int o_xh = o_ph + 16 * G::H2 * W2H;
  // row stride of the normalised frame == 8 (mod 32): S5's B operand -- 16 shifted views (ry, rx) x 4 k-steps 2g apart --
  // then reads bank 8 ry + rx + 2g, conflict-free per 32 lanes (W + 2 put three rows on the same banks)
  static constexpr int XSB = G::XS + (8 - G::XS % 32 + 32) % 32;
  static constexpr int XHN = (G::H + 2) * XSB;
  static constexpr int o_i1b = (o_xh + XHN + 3) & ~3;
  // pool-1 argmax planes in LDS as they lie in the stash (plane stride H2*W2 + 16 bytes: four banks apart -- S5 reads one byte
  // per channel plane): the image arrives by linear LDS-DMA under S1, into a part of the phase-2 area that phase 1 never touches
  static constexpr int I1SB = G::I1S;
  static_assert(I1SB % 16 == 0, "argmax planes are moved in 16-byte pieces");
  static constexpr int end2 = (o_i1b + 2 * I1SB + 3) & ~3;
  // W3 (13.8 KB) stays resident behind the phase area when the CU's 160 KB allow it; otherwise it lives in the
  // phase-1 part and is re-staged from L2 every frame
  static constexpr int fixed = 192 * 16 + 512;             // w2t + misc
  static constexpr bool W3_RESIDENT = ((end1a > end2 ? end1a : end2) + 3456 + fixed) * 4 <= 160 * 1024;
  static constexpr int end1 = W3_RESIDENT ? end1a : end1a + 3456;
  static_assert(o_i1b >= end1, "the pool-1 argmax image is filled while phase 1 is live");
  static constexpr int ph_end = end1 > end2 ? end1 : end2;
  static constexpr int o_w3s = W3_RESIDENT ? ph_end : end1a;
  static constexpr int o_w2t = W3_RESIDENT ? ph_end + 3456 : ph_end;
  static constexpr int o_misc = o_w2t + 192 * 16;
  // staging of the NEXT frame's uint8 pixels and pool-2 argmaxes (they are consumed at the top of that frame, after the
  // dy3 image -- where the mask bytes are staged -- has been filled): the part of the da2m planes that the normalised
  // frame does not cover is free while S5 runs; shapes where that is too small get an area of their own
  static constexpr int stg2_need = G::HW / 4 + 4 * G::P;
  static constexpr bool STG2_IN_DA2M = o_xh - o_da2m >= stg2_need;
  static constexpr int o_stg2 = STG2_IN_DA2M ? o_da2m : o_misc + 512;
  static constexpr int total = o_misc + 512 + (STG2_IN_DA2M ? 0 : stg2_need);
  static_assert(total * 4 <= 160 * 1024, "LDS image exceeds a CU");
};

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
  const int* frames;  // null, or [0] = how many frames to walk, [1 ...] their numbers (ss_roi_active_frames): padded frames are skipped
};

// S1's A rows are (tap, n) pairs: dW3[n][c][tap] = sum_q dy3[n][q - tap] * a2[c][q] with the B operand a2[c][q] shared by
// every tap.  Two taps = 48 rows = three full tiles: tile 0 = (tap a, n 0..15), tile 1 = (tap a, n 16..23 | tap b, n 0..7),
// tile 2 = (tap b, n 8..23); the ninth tap takes a full tile and a half-empty one.  The four k-quarter waves of each half of
// the workgroup own taps {0..3} + (tap 8, n 0..15) resp. {4..7} + (tap 8, n 16..23): seven tiles either way.
// dy3 is pixel-major [haloed pixel][24]: byte offset of tap (ky, kx) relative to the lane base at (y, x) of the haloed image
constexpr int DY3_STRIDE = 24;  // == BwdLds::DS (asserted in the kernel)
template <int S2>
__device__ constexpr int s1_off(int tap) { return ((2 - tap / 3) * S2 + (2 - tap % 3)) * DY3_STRIDE; }

template <class G, int HALF, class F>
__device__ __forceinline__ void s1_rows(const float* dy3h, const float* a2h, int kg, int i, int g, f32x4 (&acc)[7], F&& on_row) {
  constexpr int S2 = G::S2, W4 = G::W4, P2 = G::P2;
  constexpr int rows = G::P / 4 / W4;  // whole rows of the pooled-2 grid per wave: row bases + immediates
  static_assert(rows * W4 * 4 == G::P && (W4 / 4) % 2 == 0, "S1 row split");
  constexpr int t0 = 4 * HALF;
  const int dA = i < 8 ? s1_off<S2>(t0) + 16 + i : s1_off<S2>(t0 + 1) + i - 8;
  const int dB = i < 8 ? s1_off<S2>(t0 + 2) + 16 + i : s1_off<S2>(t0 + 3) + i - 8;
  // one k-step (8 LDS reads, 7 MFMAs) per pass, the next pass's reads issued before this pass's MFMAs (see S3); fully
  // unrolled: every address is one of four lane bases + an immediate
  constexpr int KPR = W4 / 4, NP = rows * KPR;
  const float* lb = dy3h + (kg * rows * S2 + g) * DY3_STRIDE;
  const float* lbi = lb + i;
  const float* lbA = lb + dA;
  const float* lbB = lb + dB;
  const float* bp = a2h + i * P2 + (kg * rows + 1) * S2 + g + 1;
  float a[2][7], b[2];
#pragma unroll
  for (int ps = -1; ps < NP; ++ps) {
    if (ps + 1 < NP) {
      const int nx = ps + 1, r = nx / KPR, xq = nx % KPR, buf = nx & 1;
      if (xq == 0) on_row(r, rows);
      const int o = (r * S2 + 4 * xq) * DY3_STRIDE;
      b[buf] = bp[r * S2 + 4 * xq];
      a[buf][0] = lbi[o + s1_off<S2>(t0)];
      a[buf][1] = lbA[o];
      a[buf][2] = lbi[o + s1_off<S2>(t0 + 1) + 8];
      a[buf][3] = lbi[o + s1_off<S2>(t0 + 2)];
      a[buf][4] = lbB[o];
      a[buf][5] = lbi[o + s1_off<S2>(t0 + 3) + 8];
      a[buf][6] = lbi[o + s1_off<S2>(8) + (HALF ? 16 : 0)];  // half 1: rows 8..15 are padding (dropped at the flush)
    }
    SS_SCHED_FENCE();
    if (ps >= 0) {
#pragma unroll
      for (int t = 0; t < 7; ++t) acc[t] = mfma16(a[ps & 1][t], b[ps & 1], acc[t]);
    }
    SS_SCHED_FENCE();
  }
}

// LISTED: the frames are p.frames[1 ...] (a clip's padding frames have d feat == 0 and add nothing: they are not walked);
// otherwise frame `it` is frame number `it`.  Two kernels because the indirection is not free in THIS one: with five more scalar
// values alive through a frame (list pointer, count, three frame numbers) it spills 48 instead of 38 scalar registers and a launch
// over all frames took 692 instead of 682 us, 924 instead of 917 at 48 x 96 (A/B on one box, gpurun_out/r4_fr_ab2.log; both walks
// inlined into one kernel behind a branch on the count were worse: 689 and 949, r4_fr_ab3.log).  The host picks (engine.py).
template <class G, bool LISTED>
__global__ __launch_bounds__(NT) void roi_cnn_bwd_kernel(CnnBwdParams p) {
  STAMP_ENTRY;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using LL = BwdLds<G>;
  const int n_walk = LISTED ? min(p.frames[0], p.N) : p.N;
  auto frame_at = [&](int it) { return LISTED ? (int)min((unsigned)p.frames[1 + it], (unsigned)(p.N - 1)) : it; };
  if (LISTED && (int)blockIdx.x >= n_walk) return;  // (the grid is sized for N: a workgroup without frames has nothing to add)
  constexpr int P = G::P, HW2 = G::HW2, HW = G::HW, H = G::H, XS = LL::XSB;
  constexpr int W = G::W, W2 = G::W2, W4 = G::W4, H2 = G::H2, S1 = G::S1, S2 = G::S2, P1 = G::P1, P2 = G::P2, W2H = LL::W2H;
  constexpr int NCH = (HW / 8 + NT - 1) / NT;     // 8-byte pixel chunks per thread (every thread of a 64x64 frame has one)
  constexpr int I1S = G::I1S;
  // The pool-1 argmax image (8 planes as they lie in the stash) reaches LDS either by LDS-DMA under S1 (shapes that are short of
  // registers: 48 x 96 spilled 43 VGPRs with the bytes held in registers through S2, 24 without) or through registers, fetched in
  // front of S2 and stored inside S3 (64 x 64: measured 682 against 693 us per launch for the DMA form, A/B on one box)
  constexpr bool I1_BY_DMA = !LL::W3_RESIDENT;
  constexpr int NI1 = I1_BY_DMA ? 1 : (I1S / 2 + NT - 1) / NT;  // 16-byte pool-1 argmax chunks per thread
  constexpr int DS = LL::DS;
  static_assert(DS == DY3_STRIDE && DS % 4 == 0 && DS >= 24, "dy3 pixel stride");
  float* a1h = lds + LL::o_a1h;
  float* a2h = lds + LL::o_ph;
  float* dy3h = lds + LL::o_dy3h;
  uint8_t* i2b = reinterpret_cast<uint8_t*>(lds + LL::o_i2b);
  float* w3s = lds + LL::o_w3s;
  float* dy2 = lds + LL::o_ph;
  float* xh = lds + LL::o_xh;
  float* w2t = lds + LL::o_w2t;
  float* misc = lds + LL::o_misc;
  float* s_dout = misc;          // [64]
  float* s_feat = misc + 64;     // [32]
  float* s_dfeat = misc + 96;    // [32]  d feat[c] / P
  float* s_stat = misc + 128;    // mu, sd of this frame (prefetched: dead after the grey-level table is built)
  float* s_gb3 = misc + 160;     // [32]  accumulated over the frame walk
  float* s_cnt = misc + 448;     // [24] positive conv3 outputs per channel of this frame (from the forward's stash)
  float* s_xn = misc + 192;      // [256] normalised value of every uint8 level for this frame

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wvu = __builtin_amdgcn_readfirstlane(wv);
  const int i = lane & 15, g = lane >> 4;
  const int E = p.E;

  for (int q = tid; q < LL::total; q += NT) lds[q] = 0.f;
  __syncthreads();
  // S2's B operand, one ds_read_b128 + one ds_read_b64 per tap: the 24 output channels n of conv3 are the K index
  // (slot (e, g) of the first four k-steps carries n = 4g + e, of the last two n = 16 + 2g + e), column = input channel c:
  //   [0, 2304)    ((tap*4 + g)*16 + c)*4 + e  = W3[4g+e][c][tap]
  //   [2304, 3456) ((tap*4 + g)*16 + c)*2 + e  = W3[16+2g+e][c][tap]
  auto stage_w3 = [&]() {
    for (int q = tid; q < 3456; q += NT) {
      int nn, c, tap;
      if (q < 2304) {
        const int e = q & 3, r = q >> 2;
        c = r & 15;
        nn = 4 * ((r >> 4) & 3) + e;
        tap = r >> 6;
      } else {
        const int q2 = q - 2304, e = q2 & 1, r = q2 >> 1;
        c = r & 15;
        nn = 16 + 2 * ((r >> 4) & 3) + e;
        tap = r >> 6;
      }
      w3s[q] = p.w3[nn * 144 + c * 9 + tap];
    }
  };
  if (LL::W3_RESIDENT) stage_w3();
  // Shapes whose LDS image has no room to keep W3 (48 x 96: the slot lies under phase 2's normalised frame) re-stage it every
  // frame.  Round 3 did that with plain loads at the frame top -- seven dependent 4-byte L2 gathers per thread with every wave
  // waiting: 3.6 k of the frame's 78.8 k cycles (profiles/round3_f_stamp_64x64_vs_48x96.txt).  Now the same gather goes by
  // LDS-DMA with per-lane source addresses (global_load_lds_dword: lane l's word lands at base + 4 l, i.e. w3s[64 piece + l]),
  // issued at the frame top and landing under S1, which does not read W3; the wait sits in front of the barrier that ends S1.
  auto stage_w3_dma = [&]() {
    static_assert(3456 % 64 == 0, "whole wave pieces");
    for (int piece = wvu; piece < 3456 / 64; piece += NWV) {
      const int q = piece * 64 + lane;
      int nn, c, tap;
      if (q < 2304) {
        const int e = q & 3, r = q >> 2;
        c = r & 15;
        nn = 4 * ((r >> 4) & 3) + e;
        tap = r >> 6;
      } else {
        const int q2 = q - 2304, e = q2 & 1, r = q2 >> 1;
        c = r & 15;
        nn = 16 + 2 * ((r >> 4) & 3) + e;
        tap = r >> 6;
      }
      ss_dma4(p.w3 + nn * 144 + c * 9 + tap, (unsigned)((LL::o_w3s + piece * 64) * 4));
    }
  };
  // S4's B operand: k = (t*3+kx)*16 + n, column j = (c, s): W2[n][c][ky = s+2-t][kx], zero outside the 3x3 window.
  // Stored [tap tk][g][column j][e] with n = 4g + e: lane (j, g) takes its four k-steps of a tap in ONE ds_read_b128
  // (MFMA slot (e, g) carries n = 4g + e for both operands).
  for (int q = tid; q < 192 * 16; q += NT) {
    const int e = q & 3, j = (q >> 2) & 15, gg = (q >> 6) & 3, tk = q >> 8;
    const int n = 4 * gg + e, t = tk / 3, kx = tk % 3, c = j & 7, s = j >> 3;
    const int ky = s + 2 - t;
    w2t[q] = (ky >= 0 && ky <= 2) ? p.w2[n * 72 + c * 9 + ky * 3 + kx] : 0.f;
  }

  // persistent per-thread accumulators
  // S1 splits K (pixels) four ways and the 14 (tap, n) tiles in two halves over the 8 waves (s1_rows)
  f32x4 acc3[7], acc2[5];
#pragma unroll
  for (int a = 0; a < 7; ++a) acc3[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int s1_kg = wvu & 3, s1_half = wvu >> 2;
#pragma unroll
  for (int a = 0; a < 5; ++a) acc2[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 accG[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};  // S5: rows (c, window slot), cols (ry, rx)
  float accfc[3] = {0.f, 0.f, 0.f}, accbfc = 0.f, accb2 = 0.f, accb1 = 0.f;

  // S3 B-operand offsets: column idx = 16*nt + i  ->  tap = idx/8, c = idx%8.  Columns 72..79 are padding: their
  // results are dropped at the flush and a D column depends on its own B column only, so they read any valid cell.
  int boff[5];
#pragma unroll
  for (int nt = 0; nt < 5; ++nt) {
    const int idx = 16 * nt + i;
    const int tap = idx >> 3, c = idx & 7;
    boff[nt] = (idx < 72) ? c * P1 + (tap / 3) * S1 + (tap % 3) : 0;
  }

  __syncthreads();
  STAMP_DECL;

  // Every per-frame input is fetched ONE FRAME AHEAD by LDS-DMA, issued at the start of S5 of the previous frame: the
  // pooled-2 map straight into its place (phase 1's a2h lies under the dense dy2 image, dead by then), the conv3 sign mask
  // into the (equally dead) dy3 planes, the uint8 frame / pool-2 argmaxes into the second staging area, the d_out row and
  // the averaged features into their misc slots.  The front of the next frame -- d feat, the dy3 image, the grey-level
  // table: barriers and LDS round trips with no MFMA beside them -- then runs INSIDE the last quarter of S5 (top1 / top2
  // below): a frame starts at S1.  No registers are involved, so
  // nothing the compiler does with its own loads can wait on these (a register prefetch cost 2.8 us of serialised HBM
  // round trips per frame: spilled pointers and split destination registers each forced an s_waitcnt vmcnt(0)).
  constexpr int STG_M3 = LL::o_dy3h, STG_PX = LL::o_stg2, STG_I2 = STG_PX + HW / 4;   // float offsets
  static_assert(8 * P <= DS * P2 && HW % 16 == 0 && P % 16 == 0, "staging");
  uint2 px[NCH];
  // fc weight column of this thread's channel (c = tid / 16; e = tid % 16 + 16 k): d feat needs nothing staged per frame
  float wq[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) wq[k] = (tid < 24 * 16 && (tid & 15) + 16 * k < E) ? p.wfc[((tid & 15) + 16 * k) * 24 + (tid >> 4)] : 0.f;
  // FAST (one 16-byte mask item per thread, whole rows of the pooled-2 grid per wave -- the 64x64 frames of BASELINE.json):
  // the front of the next frame needs no workgroup barrier.  Every wave fetches the 1 KB of mask bytes of ITS pixels into
  // the dy3 cells of those same pixels, reads them back after its own s_waitcnt and fills its rows; d feat is made two stages
  // earlier, inside S3 (its inputs are requested in S1).  The two waves of a SIMD then run their fronts at DIFFERENT passes of S5, one
  // wave's vector / LDS work under the other's MFMAs (tools/microbench/mfma_valu: a wave's v_fma stream does not slow the
  // partner's MFMAs at all); with barriers in it the front costs its full length wherever it is put.
  // Round 4: any shape whose waves own whole rows of the pooled-2 grid -- wave w rows [w H4 / NWV, (w + 1) H4 / NWV), one or
  // two mask items per lane (48 x 96: 12 rows over 8 waves, 48 or 96 items per wave) -- as long as a wave's mask bytes (32 per
  // pixel) fit into the cells of its first row (96 bytes per pixel): up to three rows per wave.  The 48 x 96 frames of the
  // reference ran the generic form (two workgroup barriers inside S5, the front at full length: S5 15.2 k cycles against 8.1 k).
  constexpr int H4 = G::H4;
  constexpr int RPW_MAX = (H4 + NWV - 1) / NWV;          // rows a wave owns at most
  constexpr int NM3F = (RPW_MAX * 2 * W4 + 63) / 64;      // mask items per lane at most
  constexpr bool FAST = H4 >= NWV && RPW_MAX <= 3 && 2 * W4 <= 64;
  const int own_r0 = (wvu * H4) / NWV, own_r1 = ((wvu + 1) * H4) / NWV;  // this wave's rows of the pooled-2 grid
  const int own_items = (own_r1 - own_r0) * 2 * W4;
  auto m3_own_dst = [&]() {  // float offset of this wave's first interior pixel in the dy3 image
    return LL::o_dy3h + ((own_r0 + 1) * S2 + 1) * DS;
  };
  static_assert(!FAST || RPW_MAX * 2 * W4 * 16 <= W4 * DS * 4, "a wave's mask piece stays inside its first row");
  auto misc_dma = [&](int nf) {  // d_out row, averaged features, counts, mean / std: one wave, four tiny DMAs
    if (wvu == NWV - 1) {
      if (lane < E) ss_dma4(p.d_out + (long)nf * p.ld_dout + lane, (unsigned)((LL::o_misc) * 4));
      if (lane < 24) ss_dma4(p.st_feat + (long)nf * ST_FEAT + lane, (unsigned)((LL::o_misc + 64) * 4));
      if (lane < 24) ss_dma4(p.st_feat + (long)nf * ST_FEAT + 24 + lane, (unsigned)((LL::o_misc + 448) * 4));
      if (lane < 2) ss_dma4(p.st_feat + (long)nf * ST_FEAT + 48 + lane, (unsigned)((LL::o_misc + 128) * 4));  // mean, std
    }
  };
  // (a 1 KB DMA instruction occupies the CU's vector-memory path for ~30 cycles; 39 of them back to back at a barrier
  // stalled every wave for 1.2 k cycles per frame, under MFMAs they are free)
  auto prefetch_frame = [&](int nf) {
    constexpr int A2_PIECES = (16 * P2 * 4 + 1023) / 1024, PX_PIECES = (HW + 1023) / 1024,
                  M3_PIECES = FAST ? 0 : (32 * P + 1023) / 1024, I2_PIECES = (16 * P + 1023) / 1024;
    const char* a2src = reinterpret_cast<const char*>(p.st_a2 + (long)nf * 16 * P2);
    const char* pxsrc = reinterpret_cast<const char*>(p.R + (long)nf * HW);
    const char* m3src = reinterpret_cast<const char*>(p.st_m3 + (long)nf * 32 * P);
    const char* i2src = reinterpret_cast<const char*>(p.st_i2 + (long)nf * 16 * P);
    if (FAST) {  // this wave's mask bytes into the cells of its own first row (front_own reads them back)
#pragma unroll
      for (int k = 0; k < NM3F; ++k)
        if (lane + 64 * k < own_items)
          ss_dma16(m3src + (2 * own_r0 * W4 + 64 * k + lane) * 16, (unsigned)(m3_own_dst() * 4 + k * 1024));
    }
    for (int piece = wvu; piece < A2_PIECES + PX_PIECES + M3_PIECES + I2_PIECES; piece += NWV) {
      int q = piece;
      if (q < A2_PIECES) {
        const int off = q * 1024 + lane * 16;
        if (off < 16 * P2 * 4) ss_dma16(a2src + off, (unsigned)(LL::o_ph * 4 + q * 1024));
        continue;
      }
      q -= A2_PIECES;
      if (q < PX_PIECES) {
        const int off = q * 1024 + lane * 16;
        if (off < HW) ss_dma16(pxsrc + off, (unsigned)(STG_PX * 4 + q * 1024));
        continue;
      }
      q -= PX_PIECES;
      if (q < M3_PIECES) {
        const int off = q * 1024 + lane * 16;
        if (off < 32 * P) ss_dma16(m3src + off, (unsigned)(STG_M3 * 4 + q * 1024));
        continue;
      }
      q -= M3_PIECES;
      const int off = q * 1024 + lane * 16;
      if (off < 16 * P) ss_dma16(i2src + off, (unsigned)(STG_I2 * 4 + q * 1024));
    }
    if (!FAST) misc_dma(nf);
  };
  constexpr int NM3 = (2 * P + NT - 1) / NT;  // 16-byte mask items (pixel, channel half) per thread
  // ---- the pieces of a frame's front
  auto front_dfeat = [&]() {  // needs the d_out row and the averaged features (landed and published)
    if (tid < 24 * 16) {  // d feat[c] = sum_e d_out[e] * Wfc[e][c]: 16 lanes per channel (one DPP row), then a shuffle tree
      float sacc = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) sacc += s_dout[(tid & 15) + 16 * k] * wq[k];  // slots beyond E stay zero
      sacc = row_sum(sacc);
      if ((tid & 15) == 0) s_dfeat[tid >> 4] = sacc / (float)P;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int idx = tid + k * NT;
      if (idx < E * 24) {
        const int e = idx / 24;
        accfc[k] += s_dout[e] * s_feat[idx - 24 * e];
      }
    }
    if (tid < E) accbfc += s_dout[tid];
  };
  auto front_table = [&]() {
    if (tid < 256) {  // the forward's xn = (u/255 - mu)/sd, once per grey level instead of once per pixel
      const float rr = (float)tid / 255.0f;
      s_xn[tid] = p.standardize ? (rr - s_stat[0]) / s_stat[1] : rr;
    }
  };
  // dy3 = mask3 * dfeat / P, pixel-major [haloed pixel][24 channels]: an item is 16 mask bytes = 16 channels of one
  // pixel -> four (two for channels 16..23) 16-byte stores
  auto fill_item = [&](const int item, const uint4& mw) {
    const int pix = item >> 1, half = item & 1;
    const unsigned wds[4] = {mw.x, mw.y, mw.z, mw.w};
    float* dst = dy3h + ((pix / W4 + 1) * S2 + (pix % W4) + 1) * DS + 16 * half;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (half == 0 || e < 2) {
        const f32x4 dv = *reinterpret_cast<const f32x4*>(&s_dfeat[16 * half + 4 * e]);
        f32x4 v;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) v[bb] = ((wds[e] >> (8 * bb)) & 1u) ? dv[bb] : 0.f;
        *reinterpret_cast<f32x4*>(dst + 4 * e) = v;
      }
    }
  };
  auto front_fill = [&](const uint4 (&m3w)[NM3]) {
#pragma unroll
    for (int k = 0; k < NM3; ++k)
      if (tid + k * NT < 2 * P) fill_item(tid + k * NT, m3w[k]);
  };
  auto front_gb3 = [&]() {
    if (tid < 24) s_gb3[tid] += s_dfeat[tid] * s_cnt[tid];  // d b3 = d feat x (number of positive conv3 outputs)
  };
  // generic path, top1: after the barrier behind which every wave's DMA pieces have landed.  Mask bytes into registers
  // (the dy3 image is about to be written over their staging area), d feat, the grey-level table.
  auto top1 = [&](uint4 (&m3w)[NM3]) {
#pragma unroll
    for (int k = 0; k < NM3; ++k)
      m3w[k] = (tid + k * NT < 2 * P) ? reinterpret_cast<const uint4*>(lds + STG_M3)[tid + k * NT] : uint4{0, 0, 0, 0};
    front_dfeat();
    front_table();
  };
  // generic path, top2: after the barrier behind which every mask byte is in a register and d feat is published: the halo
  // pixels of the dy3 image back to zero (the planes held the staged mask), every interior pixel written
  auto top2 = [&](const uint4 (&m3w)[NM3]) {
    for (int q = tid; q < (2 * S2 + 2 * G::H4) * 6; q += NT) {
      const int hp = q / 6, part = q - 6 * hp;
      int pixh;
      if (hp < S2) pixh = hp;                                           // top row
      else if (hp < 2 * S2) pixh = (G::H4 + 1) * S2 + (hp - S2);        // bottom row
      else if (hp < 2 * S2 + G::H4) pixh = (hp - 2 * S2 + 1) * S2;      // left column
      else pixh = (hp - 2 * S2 - G::H4 + 1) * S2 + W4 + 1;              // right column
      reinterpret_cast<f32x4*>(dy3h + pixh * DS)[part] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    front_fill(m3w);
    front_gb3();
  };
  // FAST path: one wave's share of the front, no barrier.  Its own DMA pieces have landed (s_waitcnt), its mask bytes come
  // back out of the cells of its own pixels, which it then fills; it zeroes the halo cells of its own rows (waves 0 and
  // NWV - 1 the top / bottom row as well); d feat was published a stage ago
  auto front_own = [&]() {
    ss_dma_wait();
    uint4 m3w[NM3F];
#pragma unroll
    for (int k = 0; k < NM3F; ++k)
      m3w[k] = (lane + 64 * k < own_items) ? *reinterpret_cast<const uint4*>(lds + m3_own_dst() + (lane + 64 * k) * 4) : uint4{0, 0, 0, 0};
    for (int q = lane; q < 2 * (own_r1 - own_r0) * 6; q += 64) {  // left / right halo pixel of each own row, six 16-byte pieces each
      const int hp = q / 6, part = q - 6 * hp;
      const int pixh = (own_r0 + (hp >> 1) + 1) * S2 + ((hp & 1) ? W4 + 1 : 0);
      reinterpret_cast<f32x4*>(dy3h + pixh * DS)[part] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (wvu == 0 || wvu == NWV - 1) {
      const int row = wvu == 0 ? 0 : G::H4 + 1;
      for (int q = lane; q < S2 * DS / 4; q += 64) reinterpret_cast<f32x4*>(dy3h + row * S2 * DS)[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < NM3F; ++k)
      if (lane + 64 * k < own_items) fill_item(2 * own_r0 * W4 + lane + 64 * k, m3w[k]);
    front_table();
    front_gb3();
  };
  // frame numbers are read two frames ahead: a scalar load that misses costs ~500 cycles, and the wait for it (lgkmcnt, shared with
  // LDS) would otherwise sit in the middle of a stage (measured: 1 % of a launch)
  int n = frame_at(blockIdx.x);
  int n_next = LISTED && (int)(blockIdx.x + gridDim.x) < n_walk ? frame_at(blockIdx.x + gridDim.x) : -1, n_after = -1;
  // (without a list the next frame's number is n + gridDim.x wherever it is needed, and nothing is carried from frame to frame)
  auto has_next = [&]() { return LISTED ? n_next >= 0 : n + (int)gridDim.x < p.N; };
  auto next_frame = [&]() { return LISTED ? n_next : n + (int)gridDim.x; };
  auto advance = [&]() {
    if constexpr (LISTED) { n = n_next; n_next = n_after; }
    else n += gridDim.x;
  };
  if (LISTED || (int)blockIdx.x < p.N) {  // the first frame's inputs and front, outside the pipeline
    prefetch_frame(n);
    if (FAST) misc_dma(n);
    ss_dma_wait();
    __syncthreads();
    if (FAST) {
      front_dfeat();
      __syncthreads();
      front_own();
    } else {
      uint4 m3w[NM3];
      top1(m3w);
      __syncthreads();
      top2(m3w);
    }
    __syncthreads();
  }

  for (int it = blockIdx.x; it < n_walk; it += gridDim.x, advance()) {
    if constexpr (LISTED) {
      const int it2 = it + 2 * (int)gridDim.x;
      n_after = it2 < n_walk ? frame_at(it2) : -1;
    }
    STAMP(15);
    // ---------------- frame top: d feat, the dy3 image and the grey-level table of this frame were made during S5 of the
    // previous one (top1 / top2); what is left are two copies out of the second staging area, needed at the phase switch
#pragma unroll
    for (int k = 0; k < NCH; ++k)
      if ((tid + k * NT) * 8 < HW) px[k] = reinterpret_cast<const uint2*>(lds + STG_PX)[tid + k * NT];
    if (tid * 16 < 16 * P) reinterpret_cast<uint4*>(i2b)[tid] = reinterpret_cast<const uint4*>(lds + STG_I2)[tid];
    if (!LL::W3_RESIDENT) stage_w3_dma();  // (read in S2, behind the wait + barrier that end S1)
    STAMP(0);

    // ---------------- S1: dW3
    // the pooled-1 map arrives by LDS-DMA (1 KB per wave instruction) under S1 / S2, one share per row iteration; it is
    // waited for before S3.  a1h held da1 of the previous frame, dead since barrier E
    auto a1_dma = [&](int part, int nparts) {
      // FAST: the next frame's d_out row / features / statistics are requested here (d feat is made inside S3)
      if (FAST && part == 0 && has_next()) misc_dma(next_frame());
      constexpr int BYTES = 8 * P1 * 4, A1_PIECES = (BYTES + 1023) / 1024;
      constexpr int I1_BYTES = 8 * I1S, I1_PIECES = I1_BY_DMA ? (I1_BYTES + 1023) / 1024 : 0;  // the pool-1 argmax image rides along
      const char* src = reinterpret_cast<const char*>(p.st_a1 + (long)n * 8 * P1);
      const char* isrc = reinterpret_cast<const char*>(p.st_i1 + (long)n * 8 * I1S);
      for (int piece = wvu + NWV * part; piece < A1_PIECES + I1_PIECES; piece += NWV * nparts) {
        if (piece < A1_PIECES) {
          const int off = piece * 1024 + lane * 16;
          if (off < BYTES) ss_dma16(src + off, (unsigned)(LL::o_a1h * 4 + piece * 1024));
        } else {
          const int off = (piece - A1_PIECES) * 1024 + lane * 16;
          if (off < I1_BYTES) ss_dma16(isrc + off, (unsigned)(LL::o_i1b * 4 + (piece - A1_PIECES) * 1024));
        }
      }
    };
    if (s1_half == 0) s1_rows<G, 0>(dy3h, a2h, s1_kg, i, g, acc3, a1_dma);
    else s1_rows<G, 1>(dy3h, a2h, s1_kg, i, g, acc3, a1_dma);
    // S2's epilogue reads the pool-2 argmaxes that the frame top copied out of staging (and, for shapes whose W3 is
    // re-staged per frame, S2 reads what stage_w3 wrote): no wave starts S2 before every wave has made its copies.
    // (Found the hard way: this used to be a barrier of the diagnostic build only.  While S2 still wrote a da2m image over
    // the staged pixels, hipcc sank the frame top's LDS read of them to the end of S1 -- legal for one wave -- where a
    // faster wave's stores overtook it: d W1 moved by up to 4e-2 relative between identical launches;
    // tools/bwd_determinism.py and the kernel test now check that.)
    if (!LL::W3_RESIDENT) ss_dma_wait();  // this wave's pieces of W3 (and of the pooled-1 map issued so far) have landed
    __syncthreads();
    STAMP(3);
    // pool-1 argmaxes for S5: fetched here so that HBM answers under S2 (register form)
    uint4 ix1[NI1];
#pragma unroll
    for (int k = 0; k < NI1; ++k)
      if (!I1_BY_DMA && (tid + k * NT) * 16 < 8 * I1S)
        ix1[k] = reinterpret_cast<const uint4*>(p.st_i1 + (long)n * 8 * I1S)[tid + k * NT];
    // ---------------- S2: da2 (masked by a2 > 0), kept in registers ; db2.  ONE pass per wave over all its pixel tiles (tile
    // w, w + NWV, w + 2 NWV: two at 64 x 64, two or three at 48 x 96), which share the W3 reads.  (Round 3 ran 48 x 96's 18 tiles
    // as a pass of sixteen and a second pass of two: a second set of B reads, a second epilogue and two register sets of results
    // held through the phase switch -- S2 13.0 k cycles for an MFMA floor of 8.6 k.)
    constexpr int S2_TILES = P / 16;
    constexpr int S2_MAXT = (S2_TILES + NWV - 1) / NWV;  // tiles of a wave at most
    constexpr int S2_FULL = S2_TILES % NWV;              // waves [0, S2_FULL) hold S2_MAXT tiles, the others one fewer (0: all S2_MAXT)
    static_assert(S2_MAXT >= 1 && S2_MAXT <= 3, "S2 tile split");
    float s2v[S2_MAXT][4];
    unsigned s2o[S2_MAXT];  // four argmax bytes each
    {
      auto s2_pass = [&](auto nt_c) {
        constexpr int NTW = decltype(nt_c)::value;  // tiles of this wave
        int w3off = (LL::o_w3s + (g * 16 + i) * 4) * 4, w3off8 = (LL::o_w3s + 2304 + (g * 16 + i) * 2) * 4;
        asm volatile("" : "+v"(w3off), "+v"(w3off8));  // W3 may lie beyond the 64 KB reach of a ds_read immediate (see S5)
        const float* bt16 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds) + w3off);
        const float* bt8 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds) + w3off8);
        // per tap and pixel tile: channels 4g..4g+3 in one ds_read_b128 (k-steps 0..3), channels 16+2g, 17+2g in one
        // ds_read_b64 (k-steps 4, 5); the tap shift is an immediate
        const float* ap[NTW];
        f32x4 acc[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          const int pp = 16 * (wvu + NWV * j) + i;
          ap[j] = dy3h + ((pp / W4 + 2) * S2 + (pp % W4 + 2)) * DS;
          acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // a tap is two steps -- channels 0..15 (ds_read_b128 each, 4 NTW MFMAs) and 16..23 (ds_read_b64 each, 2 NTW MFMAs) -- and
        // every step's reads are issued before the MFMAs of the step in front of it (see S3)
        f32x4 b16, a16[NTW];
        float2 b8, a8[NTW];
        auto back_of = [](int tap) { return ((tap / 3) * S2 + (tap % 3)) * DS; };
        b16 = *reinterpret_cast<const f32x4*>(bt16);
#pragma unroll
        for (int j = 0; j < NTW; ++j) a16[j] = *reinterpret_cast<const f32x4*>(ap[j] + 4 * g);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int back = back_of(tap);
          b8 = *reinterpret_cast<const float2*>(bt8 + tap * 128);
#pragma unroll
          for (int j = 0; j < NTW; ++j) a8[j] = *reinterpret_cast<const float2*>(ap[j] - back + 16 + 2 * g);
          SS_SCHED_FENCE();
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[j] = mfma16(a16[j][e], b16[e], acc[j]);
          SS_SCHED_FENCE();
          if (tap + 1 < 9) {
            const int nb = back_of(tap + 1);
            b16 = *reinterpret_cast<const f32x4*>(bt16 + (tap + 1) * 256);
#pragma unroll
            for (int j = 0; j < NTW; ++j) a16[j] = *reinterpret_cast<const f32x4*>(ap[j] - nb + 4 * g);
          }
          SS_SCHED_FENCE();
#pragma unroll
          for (int j = 0; j < NTW; ++j) acc[j] = mfma16(a8[j].x, b8.x, acc[j]);
#pragma unroll
          for (int j = 0; j < NTW; ++j) acc[j] = mfma16(a8[j].y, b8.y, acc[j]);
          SS_SCHED_FENCE();
        }
        // the masked gradient and the pool-2 argmax of its four pixels stay in registers: this wave scatters them into the
        // dense dy2 image itself, right behind barrier D (no da2m image, no second barrier)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          unsigned o = 0;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int pq = 16 * (wvu + NWV * j) + 4 * g + r;
            const float av = a2h[i * P2 + (pq / W4 + 1) * S2 + (pq % W4) + 1];
            const float v = av > 0.f ? acc[j][r] : 0.f;
            s2v[j][r] = v;
            o |= (unsigned)i2b[pq * 16 + i] << (8 * r);
            accb2 += v;
          }
          s2o[j] = o;
        }
      };
      if (S2_FULL == 0 || wvu < S2_FULL) s2_pass(std::integral_constant<int, S2_MAXT>{});
      else if constexpr (S2_MAXT > 1) s2_pass(std::integral_constant<int, (S2_MAXT > 1 ? S2_MAXT - 1 : 1)>{});
    }
    if (FAST && wvu == NWV - 1) ss_dma_wait();  // the next frame's d_out row has landed: published by barrier D
    __syncthreads();  // D: dy3h / a2h / w3 / the pool-2 argmaxes are dead
    STAMP(4);

    // ---------------- T: phase switch.  The dense dy2 image is written over the phase-1 area (every read of it ended before
    // barrier D): each wave expands the pixels it computed in S2 -- the value goes to its pool winner's cell, zeros to the
    // other three.
    {
#pragma unroll
      for (int j = 0; j < S2_MAXT; ++j) {
        const int tile = wvu + NWV * j;
        if (tile < S2_TILES) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int pq = 16 * tile + 4 * g + r, qy = pq / W4, qx = pq % W4;
            // a wave's store covers 4 pixels (one per lane group g) x 16 channels and a dy2 pixel is 16 banks wide: odd
            // lane groups write the right-hand window column first, so that the 32 lanes of a half spread over all banks
            const int par = g & 1;
            float* d0 = dy2 + ((2 * qy) * W2H + 2 * qx + 1 + par) * 16 + i;
            float* d1 = dy2 + ((2 * qy) * W2H + 2 * qx + 2 - par) * 16 + i;
            const float v = s2v[j][r];
            const int o = (s2o[j] >> (8 * r)) & 255u;
            d0[0] = o == par ? v : 0.f;
            d1[0] = o == 1 - par ? v : 0.f;
            d0[16 * W2H] = o == 2 + par ? v : 0.f;
            d1[16 * W2H] = o == 3 - par ? v : 0.f;
          }
        }
      }
      STAMP(10);
    }
    ss_dma_wait();    // this wave's share of the pooled-1 map has landed in LDS
    __syncthreads();  // T done
    STAMP(5);

    // The normalised frame (interior by table lookup, halo cells zeroed: the area held phase-1 data) and the pool-1 argmax
    // bytes are needed by S5 only; the frame is made by every thread from its OWN registers (px) and the grey-level table (the
    // argmax bytes arrive by DMA since round 4):
    // like the next frame's front they run per wave, without a barrier, inside S3 -- the two waves of a SIMD at
    // different passes, one wave's table lookups and LDS stores under its partner's MFMAs (they were 2.4 k cycles of
    // the phase switch, with every wave storing and nobody multiplying).  The zero columns of dy2 and the next frame's
    // d feat ride along.
    auto xh_own = [&]() {
      // normalised frame: interior by table lookup, halo cells zeroed (the area held phase-1 data)
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int q = tid + k * NT;
        if (q * 8 < HW) {
          const int lin = q * 8;
          float* dst = xh + (lin / W + 1) * XS + (lin % W + 1);
          const unsigned wds[2] = {px[k].x, px[k].y};
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int b = 0; b < 4; ++b) dst[4 * e + b] = s_xn[(wds[e] >> (8 * b)) & 255u];
        }
      }
      for (int q = tid; q < 2 * XS + 2 * H; q += NT) {
        int cell;
        if (q < XS) cell = q;                                   // top row
        else if (q < 2 * XS) cell = (H + 1) * XS + (q - XS);    // bottom row
        else if (q < 2 * XS + H) cell = (q - 2 * XS + 1) * XS;  // left column
        else cell = (q - 2 * XS - H + 1) * XS + W + 1;          // right column (cells beyond it are never read)
        xh[cell] = 0.f;
      }
#pragma unroll
      for (int k = 0; k < NI1; ++k)
        if (!I1_BY_DMA && (tid + k * NT) * 16 < 8 * I1S) {  // the stash's plane layout is the LDS layout
          uint32_t* dst = reinterpret_cast<uint32_t*>(lds + LL::o_i1b) + 4 * (tid + k * NT);
          dst[0] = ix1[k].x; dst[1] = ix1[k].y; dst[2] = ix1[k].z; dst[3] = ix1[k].w;
        }
      // the zero columns of dy2 (the area held phase-1 data): S3 reads interior columns only, S4 is behind a barrier
      for (int q = tid; q < 2 * H2 * 4; q += NT) {
        const int y = q >> 3, side = (q >> 2) & 1, part = q & 3;
        reinterpret_cast<f32x4*>(dy2)[(y * W2H + (side ? W2 + 1 : 0)) * 4 + part] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      // FAST: d feat of the NEXT frame (its d_out row landed before barrier D; s_dfeat of THIS frame was last read in S5 of
      // the previous one, the new value is read in S5 of this one, two barriers on)
      if (FAST && has_next()) front_dfeat();
    };
    // ---------------- S3: dW2.  A wave owns whole rows of the pooled-1 grid: one A base and five B bases per wave,
    // everything else is an immediate offset (VALU work between MFMAs is not hidden by them).  The two waves of a SIMD
    // leave the barrier together and then run in lockstep -- both read, both multiply -- so a wave's operand reads are
    // NOT covered by its partner's MFMAs: each pass issues the reads of the next one before its own MFMAs (two register
    // sets), fully unrolled so that every address is base + immediate.
    {
      constexpr int kpw = HW2 / NWV, rows = kpw / W2;
      static_assert(rows * W2 == kpw && (W2 / 4) % 2 == 0, "S3 row split");
      constexpr int PPR = W2 / 8, NP = rows * PPR;  // passes of two k-steps (12 LDS reads, 10 MFMAs)
      const int y0 = wvu * rows;
      const float* ap = dy2 + (y0 * W2H + 1 + g) * 16 + i;
      const float* bpn[5];
#pragma unroll
      for (int nt = 0; nt < 5; ++nt) bpn[nt] = a1h + y0 * S1 + g + boff[nt];
      float a[2][2], b[2][2][5];
#pragma unroll
      for (int ps = -1; ps < NP; ++ps) {
        if (ps + 1 < NP) {
          const int nx = ps + 1, r = nx / PPR, xq = 2 * (nx % PPR), buf = nx & 1;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            a[buf][u] = ap[r * W2H * 16 + (xq + u) * 64];
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) b[buf][u][nt] = bpn[nt][r * S1 + (xq + u) * 4];
          }
        }
        SS_SCHED_FENCE();
        if (ps >= 0) {
          const int buf = ps & 1;
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) acc2[nt] = mfma16(a[buf][u], b[buf][u][nt], acc2[nt]);
        }
        SS_SCHED_FENCE();
        if (ps == NP / 4 && wvu < NWV / 2) xh_own();          // (wave-uniform; the next pass's reads are in flight)
        if (ps == (3 * NP) / 4 && wvu >= NWV / 2) xh_own();
      }
    }
    __syncthreads();  // S4 overwrites a1 with da1 in place; S3 reads a1 rows of the neighbouring waves' bands as well
    STAMP(6);
    // ---------------- S4 + S5: da1 for rows y, y+1 of a 16-pixel column block (two blocks per pass share the W2
    // table reads); dW1 / db1 in the epilogue
    {
      constexpr int xt_n = W2 / 16;
      constexpr int chains = (H2 / 2) * xt_n;
      const int c = i & 7, s = i >> 3;
      // (a wave needs its two chains: one chain of dependent MFMAs alone runs at half rate -- measured with single-chain
      // passes for waves 4..7, meant to stagger the epilogues of a SIMD's two waves: S4 15.5 k -> 20.3 k cycles)
      // A pass takes two NEIGHBOURING chains: with two column blocks per row pair both lie on the same rows, so both
      // are interior, both on the first row pair or both on the last one (see the tap loop below)
      constexpr int npairs = (chains + 1) / 2;
#pragma unroll 1
      for (int pj = wvu; pj < npairs; pj += NWV) {  // wvu: pass control and the edge tests are scalar
        const int ch = 2 * pj, ch2 = ch + 1;
        const bool two = ch2 < chains;
        const int chb = two ? ch2 : ch;
        const int y0_ = 2 * (ch / xt_n), x0_ = 16 * (ch % xt_n) + i;
        const int y1_ = 2 * (chb / xt_n), x1_ = 16 * (chb % xt_n) + i;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1x = acc0;
        // source pixel of tap (t, kx) for output (y, x) is (y - 1 + t, x + 1 - kx): lane base + immediate; the zero
        // columns take care of x = -1 / W2, rows outside the map exist only for t = 0 (y = 0) and t = 3 (y = H2 - 2)
        // and are skipped (wave-uniform)
        const float* ab0 = dy2 + ((y0_ - 1) * W2H + x0_) * 16 + 4 * g;
        const float* ab1 = dy2 + ((y1_ - 1) * W2H + x1_) * 16 + 4 * g;
        const float* bb = w2t + (g * 16 + i) * 4;
        // one tap (3 x ds_read_b128, 8 MFMAs) per step, the next tap's reads issued before this tap's MFMAs (see S3).
        // Source rows above / below the map (t = 0 of the first row pair, t = 3 of the last) contribute nothing.  Four
        // copies of the tap loop: KIND 0 both chains interior (12 taps, no test), 1 both on the first row pair (taps 3..11),
        // 2 both on the last (taps 0..8), 3 anything else: every tap's row tested, scalar branches around the reads
        // (all passes through KIND 3 cost 15 us per launch: the stage ends when its slowest wave does)
        float* cell0 = a1h + c * P1 + (y0_ + s + 1) * S1 + x0_ - i + 4 * g + 1;
        float* cell1 = a1h + c * P1 + (y1_ + s + 1) * S1 + x1_ - i + 4 * g + 1;
        float mk0[4], mk1[4];
        auto taps = [&](auto kind_tag) {
          constexpr int KIND = decltype(kind_tag)::value;
          constexpr int T0 = KIND == 1 ? 3 : 0, T1 = KIND == 2 ? 9 : 12;
          f32x4 a0[2], a1[2], b[2];
#pragma unroll
          for (int tk = T0 - 1; tk < T1; ++tk) {
            if (tk + 1 < T1) {
              const int nx = tk + 1, t = nx / 3, kx = nx % 3, buf = nx & 1;
              const int off = (t * W2H + 2 - kx) * 16;
              const bool ok0 = KIND != 3 || (!(t == 0 && y0_ == 0) && !(t == 3 && y0_ == H2 - 2));
              const bool ok1 = KIND != 3 || (!(t == 0 && y1_ == 0) && !(t == 3 && y1_ == H2 - 2));
              a0[buf] = a1[buf] = f32x4{0.f, 0.f, 0.f, 0.f};
              if (ok0) a0[buf] = *reinterpret_cast<const f32x4*>(ab0 + off);
              if (ok1) a1[buf] = *reinterpret_cast<const f32x4*>(ab1 + off);
              b[buf] = *reinterpret_cast<const f32x4*>(bb + nx * 256);
            }
            if (tk == T1 - 3) {  // the a1 cells whose sign masks the result: read under the last taps, not in the epilogue
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                mk0[r] = cell0[r];
                mk1[r] = cell1[r];
              }
            }
            SS_SCHED_FENCE();
            if (tk >= T0) {
              const int buf = tk & 1;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                acc0 = mfma16(a0[buf][e], b[buf][e], acc0);
                acc1x = mfma16(a1[buf][e], b[buf][e], acc1x);
              }
            }
            SS_SCHED_FENCE();
          }
        };
        {
          const bool first0 = y0_ == 0, first1 = y1_ == 0, last0 = y0_ == H2 - 2, last1 = y1_ == H2 - 2;
          if (!(first0 || first1 || last0 || last1)) taps(std::integral_constant<int, 0>{});
          else if (first0 && first1 && !last0 && !last1) taps(std::integral_constant<int, 1>{});
          else if (last0 && last1 && !first0 && !first1) taps(std::integral_constant<int, 2>{});
          else taps(std::integral_constant<int, 3>{});
        }
        // D: row 4g+r -> pixel x0+4g+r of row y+s ; column i -> channel c.  Mask by a1 > 0 and leave da1 IN PLACE of
        // a1 (each cell is read and rewritten by exactly one lane; S3 finished with a1 before the barrier above).
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if (half == 1 && !two) break;
          float* cell = half ? cell1 : cell0;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d = (half ? mk1[r] : mk0[r]) > 0.f ? (half ? acc1x[r] : acc0[r]) : 0.f;
            cell[r] = d;
            accb1 += d;
          }
        }
      }
    }
    __syncthreads();  // da1 complete
    STAMP(9);
    // ---------------- S5: conv1 weight gradient on the pooled grid (K = pooled pixels, whole rows per wave: row
    // bases + immediates)
    {
      constexpr int kpw = HW2 / NWV, rows = kpw / W2;
      static_assert(rows * W2 == kpw && (W2 / 4) % 4 == 0, "S5 row split");
      const int y0 = wvu * rows;
      const int c = i & 7, os = i >> 3;         // A rows: tile 0 -> window slot os, tile 1 -> slot 2 + os
      const int boff5 = (i >> 2) * XS + (i & 3);  // B column (ry, rx)
      // i1b and xh lie beyond the 64 KB reach of a ds_read immediate: keep the whole byte offset in a register the
      // compiler cannot split, so that the per-read constants stay immediates instead of one v_add each
      const float* dp = a1h + c * P1 + (y0 + 1) * S1 + g + 1;
      int ioff = LL::o_i1b * 4 + c * LL::I1SB + y0 * W2 + g;
      int xoff = (LL::o_xh + 2 * y0 * XS + 2 * g + boff5) * 4;
      asm volatile("" : "+v"(ioff), "+v"(xoff));
      const uint8_t* ip = reinterpret_cast<const uint8_t*>(lds) + ioff;
      const float* xp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds) + xoff);
      constexpr int KP = 4;                  // k-steps per pass: 3 KP LDS reads, 2 KP selects, 2 KP MFMAs
      constexpr int PPR = W2 / 4 / KP, NP = rows * PPR;
      constexpr int FRONT = NP - (NP + 3) / 4;  // pass after which the next frame's front runs (the DMA has ~3/4 of S5 to land)
      constexpr int FRONT_A = NP / 2;           // FAST: waves 0 .. NWV/2 - 1 run theirs here
      const bool more = has_next();
      // fully unrolled, the next pass's reads issued before this pass's selects and MFMAs (see S3)
      float d[2][KP], xv[2][KP];
      int ix[2][KP];
#pragma unroll
      for (int ps = -1; ps < NP; ++ps) {
        if (ps + 1 < NP) {
          const int nx = ps + 1, r = nx / PPR, xq = KP * (nx % PPR), buf = nx & 1;
          // dy2 / dy3 areas are dead (S4 is through): the next frame's inputs are requested with the first pass
          if (nx == 0 && more) prefetch_frame(next_frame());
#pragma unroll
          for (int u = 0; u < KP; ++u) {
            d[buf][u] = dp[r * S1 + (xq + u) * 4];
            ix[buf][u] = ip[r * W2 + (xq + u) * 4];
            xv[buf][u] = xp[2 * r * XS + (xq + u) * 8];
          }
        }
        if (ps >= 0) {
          const int buf = ps & 1;
          // every operand is selected into a register of its own BEFORE the MFMA block: a select that rewrites the
          // source register of the MFMA in front of it waits for that MFMA (measured: 2.7x the MFMA time)
          float s0[KP], s1[KP];
#pragma unroll
          for (int u = 0; u < KP; ++u) {
            s0[u] = ix[buf][u] == os ? d[buf][u] : 0.f;
            s1[u] = ix[buf][u] == 2 + os ? d[buf][u] : 0.f;
          }
          SS_SCHED_FENCE();
#pragma unroll
          for (int u = 0; u < KP; ++u) {
            accG[0] = mfma16(s0[u], xv[buf][u], accG[0]);
            accG[1] = mfma16(s1[u], xv[buf][u], accG[1]);
          }
        }
        SS_SCHED_FENCE();
        if (FAST) {  // the two waves of a SIMD (w, w + NWV / 2) take different passes
          if (ps == FRONT_A - 1 && more && wvu < NWV / 2) front_own();
          if (ps == FRONT - 1 && more && wvu >= NWV / 2) front_own();
        } else if (ps == FRONT - 1 && more) {  // (wave-uniform; the reads of pass FRONT are already in flight)
          ss_dma_wait();
          __syncthreads();  // F1: every wave's pieces of the next frame have landed
          uint4 m3w[NM3];
          top1(m3w);
          __syncthreads();  // F2: mask bytes are in registers, d feat is published
          top2(m3w);
        }
      }
    }
    __syncthreads();  // E
    STAMP(7);
  }

  // ---------------- flush: reduce the per-wave partials through LDS, then one atomic per element
  float* r_w3 = lds;            // [3456]
  float* r_w2 = r_w3 + 3456;    // [1152]
  float* r_w1 = r_w2 + 1152;    // [72]
  float* r_b1 = r_w1 + 72;      // [8]
  float* r_b2 = r_b1 + 8;       // [16]
  float* r_fc = r_b2 + 16;      // [E*24]
  const int rtot = 3456 + 1152 + 72 + 8 + 16 + E * 24;
  // (r_* overlay a1h and the phase area, both dead; s_gb3 in misc is untouched)
  // d W3 and d W2 -- 48 of a lane's 56 partial sums -- go through one LDS REGION PER WAVE with plain stores and are summed over the
  // eight regions by the threads that then issue the global atomics.  The first version added all partials into one image with LDS
  // float atomics: ds_add_f32 retires well under one lane-operation per clock (measured on the config-5 weight gradients: 110 k of
  // them in 178 k cycles), 25 k of them here.  Shapes whose LDS image is too small for eight regions keep the atomics.
  constexpr int RW = 3456 + 1152;
  constexpr bool REGIONS = 8 * RW + 96 + 64 * 24 <= LL::o_misc;
  float* r_w3w = REGIONS ? lds + wvu * RW : r_w3;  // this wave's region (d W3, then d W2)
  float* r_w2w = r_w3w + 3456;
  if (REGIONS) {
    r_w1 = lds + 8 * RW;
    r_b1 = r_w1 + 72;
    r_b2 = r_b1 + 8;
    r_fc = r_b2 + 16;
    for (int q = tid; q < (8 * RW + 96) / 4; q += NT) reinterpret_cast<f32x4*>(lds)[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  } else {
    for (int q = tid; q < rtot; q += NT) lds[q] = 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * g + r, t0 = 4 * s1_half + 2 * (t / 3);
      int nn, tap;
      if (t == 6) { nn = s1_half ? 16 + row : row; tap = 8; }
      else if (t % 3 == 0) { nn = row; tap = t0; }
      else if (t % 3 == 1) { nn = row < 8 ? 16 + row : row - 8; tap = row < 8 ? t0 : t0 + 1; }
      else { nn = 8 + row; tap = t0 + 1; }
      if (nn < 24) {
        if (REGIONS) r_w3w[nn * 144 + i * 9 + tap] = acc3[t][r];  // (a wave holds every element at most once)
        else atomicAdd(&r_w3[nn * 144 + i * 9 + tap], acc3[t][r]);
      }
    }
#pragma unroll
  for (int nt = 0; nt < 5; ++nt) {
    const int idx = 16 * nt + i;
    if (idx < 72) {
      const int tap = idx >> 3, c = idx & 7;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (REGIONS) r_w2w[(4 * g + r) * 72 + c * 9 + tap] = acc2[nt][r];
        else atomicAdd(&r_w2[(4 * g + r) * 72 + c * 9 + tap], acc2[nt][r]);
      }
    }
  }
  // S5 accumulators: row 16mt+4g+r = (c = row&7, window slot o = row>>3), column i = (ry, rx):
  // dW1[c][ky][kx] gets G[(c,o)][(oy+ky, ox+kx)]
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * mt + 4 * g + r, c = row & 7, o = row >> 3;
      const int ky = (i >> 2) - (o >> 1), kx = (i & 3) - (o & 1);
      if (ky >= 0 && ky <= 2 && kx >= 0 && kx <= 2) atomicAdd(&r_w1[c * 9 + ky * 3 + kx], accG[mt][r]);
    }
  {
    float s = accb1;  // lanes with equal (i & 7) share a channel: fold i^8 and the 4 lane groups, then the waves
    s += __shfl_xor(s, 8, 64);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lane < 8) atomicAdd(&r_b1[i & 7], s);
  }
  {
    float s = accb2;
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (g == 0) atomicAdd(&r_b2[i], s);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int idx = tid + k * NT;
    if (idx < E * 24) r_fc[idx] = accfc[k];
  }
  __syncthreads();
  if (REGIONS) {
    for (int q = tid; q < RW; q += NT) {
      float s8[8];
#pragma unroll
      for (int w = 0; w < 8; ++w) s8[w] = lds[w * RW + q];
      const float sum = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
      atomicAdd(q < 3456 ? &p.g_w3[q] : &p.g_w2[q - 3456], sum);
    }
  } else {
    for (int q = tid; q < 3456; q += NT) atomicAdd(&p.g_w3[q], r_w3[q]);
    for (int q = tid; q < 1152; q += NT) atomicAdd(&p.g_w2[q], r_w2[q]);
  }
  if (tid < 72) atomicAdd(&p.g_w1[tid], r_w1[tid]);
  if (tid < 8) atomicAdd(&p.g_b1[tid], r_b1[tid]);
  if (tid < 16) atomicAdd(&p.g_b2[tid], r_b2[tid]);
  if (tid < 24) atomicAdd(&p.g_b3[tid], s_gb3[tid]);
  for (int q = tid; q < E * 24; q += NT) atomicAdd(&p.g_wfc[q], r_fc[q]);
  if (tid < E) atomicAdd(&p.g_bfc[tid], accbfc);
  STAMP_FLUSH();
}

template <class G>
int launch_bwd(const CnnBwdParams& p, hipStream_t s) {
  using LL = BwdLds<G>;
  constexpr size_t lds_bytes = (size_t)LL::total * sizeof(float);
  static_assert(lds_bytes <= 160 * 1024, "ROI size does not fit the CU's LDS");
  // the K = pixel splits need whole 4-pixel k-steps per wave; mask / argmax-2 maps are one 16-byte piece per thread
  static_assert(G::P % (4 * NWV) == 0 && G::HW2 % (4 * NWV) == 0 && 24 * G::P <= 16 * NT, "unsupported ROI size");
  static bool attr_set = false;
  if (!attr_set) {
    for (const void* fn : {reinterpret_cast<const void*>(roi_cnn_bwd_kernel<G, false>),
                           reinterpret_cast<const void*>(roi_cnn_bwd_kernel<G, true>)})
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return SS_ERR_LAUNCH;
    attr_set = true;
  }
  const int cap = ss_cnn_max_wgs > 0 ? ss_cnn_max_wgs : ss_device_cus();
  const int grid = p.N < cap ? p.N : cap;
  if (p.frames) hipLaunchKernelGGL((roi_cnn_bwd_kernel<G, true>), dim3(grid), dim3(NT), lds_bytes, s, p);
  else hipLaunchKernelGGL((roi_cnn_bwd_kernel<G, false>), dim3(grid), dim3(NT), lds_bytes, s, p);
  return ss_launch_status();
}

}  // namespace

extern "C" int ss_roi_cnn_bwd_frames(const uint8_t* R, int N, int H, int W, int standardize, const float* w1,
                                     const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                                     const float* wfc, const float* bfc, int E, const float* st_a1, const uint8_t* st_i1,
                                     const float* st_a2, const uint8_t* st_i2, const uint8_t* st_m3, const float* st_feat,
                                     const int* stash_sizes, const float* d_out, int ld_dout, float* g_w1, float* g_b1, float* g_w2,
                                     float* g_b2, float* g_w3, float* g_b3, float* g_wfc, float* g_bfc, const int* frames,
                                     ss_stream_t stream) {
  (void)w1; (void)b1; (void)b2; (void)b3; (void)bfc;  // the stashed activations already contain their effect
  SS_REQUIRE(R && w2 && w3 && wfc && st_a1 && st_i1 && st_a2 && st_i2 && st_m3 && st_feat && stash_sizes && d_out, SS_ERR_ARG);
  SS_REQUIRE(g_w1 && g_b1 && g_w2 && g_b2 && g_w3 && g_b3 && g_wfc && g_bfc, SS_ERR_ARG);
  SS_REQUIRE(N > 0 && E > 0 && ld_dout >= E, SS_ERR_ARG);
  SS_REQUIRE(E <= 64, SS_ERR_UNSUPPORTED);
  CnnBwdParams p;
  p.R = R; p.N = N; p.standardize = standardize; p.w2 = w2; p.w3 = w3; p.wfc = wfc; p.E = E;
  p.st_a1 = st_a1; p.st_i1 = st_i1; p.st_a2 = st_a2; p.st_i2 = st_i2; p.st_m3 = st_m3; p.st_feat = st_feat;
  p.d_out = d_out; p.ld_dout = ld_dout;
  p.g_w1 = g_w1; p.g_b1 = g_b1; p.g_w2 = g_w2; p.g_b2 = g_b2; p.g_w3 = g_w3; p.g_b3 = g_b3; p.g_wfc = g_wfc; p.g_bfc = g_bfc;
  p.frames = frames;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // the stash must have been laid out by a forward kernel compiled against the same Geom as this file
#define SS_DISPATCH(HH, WW)                                                                             \
  if (H == HH && W == WW) {                                                                            \
    using G_ = Geom<HH, WW>;                                                                           \
    SS_REQUIRE(stash_sizes_match<G_>(stash_sizes), SS_ERR_ARG);                                        \
    return launch_bwd<G_>(p, s);                                                                       \
  }
  SS_CNN_SHAPES(SS_DISPATCH)
#undef SS_DISPATCH
  return SS_ERR_UNSUPPORTED;
}

extern "C" int ss_roi_cnn_bwd(const uint8_t* R, int N, int H, int W, int standardize, const float* w1,
                              const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                              const float* wfc, const float* bfc, int E, const float* st_a1, const uint8_t* st_i1,
                              const float* st_a2, const uint8_t* st_i2, const uint8_t* st_m3, const float* st_feat,
                              const int* stash_sizes, const float* d_out, int ld_dout, float* g_w1, float* g_b1, float* g_w2, float* g_b2,
                              float* g_w3, float* g_b3, float* g_wfc, float* g_bfc, ss_stream_t stream) {
  return ss_roi_cnn_bwd_frames(R, N, H, W, standardize, w1, b1, w2, b2, w3, b3, wfc, bfc, E, st_a1, st_i1, st_a2, st_i2, st_m3,
                               st_feat, stash_sizes, d_out, ld_dout, g_w1, g_b1, g_w2, g_b2, g_w3, g_b3, g_wfc, g_bfc, nullptr,
                               stream);
}
