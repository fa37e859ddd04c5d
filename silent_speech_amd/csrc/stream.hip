// Sliding-window serving for many camera streams on one GPU (SURVEY 8f-4 / BASELINE config 4).
//
// The reference keeps, per camera, `deque(maxlen=max_t)` of per-frame features, and every PRED_EVERY frames -- once
// WARMUP_MIN frames are in -- zero-pads the deque to (max_t, D) and runs the model on it
// (/root/reference/inactive/live_feed.py:155, :163-164, :201-213); important_landmarks.py:131-144 gates on a smoothed
// mouth openness with hysteresis.  Here S streams share device-resident rings: (S, max_t, D) features and
// (S, max_t, H*W) ROI bytes, a head and a count per stream.  A tick is: push one frame for any subset of streams,
// update their gates, and build the frame map (oldest -> newest, -1 = padding) of the streams that are due; the
// windows themselves are gathered by ss_batch_gather_f32/u8 (batch.hip) and go through the forward pass unchanged.
#include "ss_common.h"

namespace {

// one workgroup per pushed frame: rows of D floats and, optionally, frame_bytes bytes
__global__ __launch_bounds__(256) void ring_push_kernel(float* __restrict__ ring_x, uint8_t* __restrict__ ring_r, int max_t, int D,
                                                        int chunks, const int32_t* __restrict__ ids, const float* __restrict__ feats,
                                                        const uint8_t* __restrict__ rois, int32_t* __restrict__ head,
                                                        int32_t* __restrict__ count, int32_t* __restrict__ frames_seen) {
  const int s = ids[blockIdx.x];
  const int h = head[s];
  float* dx = ring_x + ((long)s * max_t + h) * D;
  const float* sx = feats + (long)blockIdx.x * D;
  for (int d = threadIdx.x; d < D; d += 256) dx[d] = sx[d];
  if (ring_r) {
    uint4* dr = reinterpret_cast<uint4*>(ring_r) + ((long)s * max_t + h) * chunks;
    const uint4* sr = reinterpret_cast<const uint4*>(rois) + (long)blockIdx.x * chunks;
    for (int c = threadIdx.x; c < chunks; c += 256) dr[c] = sr[c];
  }
  __syncthreads();  // every thread has read head[s] before it moves
  if (threadIdx.x == 0) {
    head[s] = (h + 1 == max_t) ? 0 : h + 1;
    count[s] = min(count[s] + 1, max_t);
    frames_seen[s] += 1;
  }
}

// frame map of the windows of n selected streams, oldest frame first; lengths[i] = frames in the ring
__global__ __launch_bounds__(256) void ring_window_map_kernel(const int32_t* __restrict__ ids, int n, int max_t,
                                                              const int32_t* __restrict__ head, const int32_t* __restrict__ count,
                                                              int32_t* __restrict__ frame_map, int64_t* __restrict__ lengths) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= n * max_t) return;
  const int i = q / max_t, j = q - i * max_t;
  const int s = ids[i], c = count[s];
  int row = -1;
  if (j < c) {
    int pos = head[s] - c + j;
    if (pos < 0) pos += max_t;
    row = s * max_t + pos;
  }
  frame_map[q] = row;
  if (j == 0) lengths[i] = c;
}

// Mouth openness of the pushed frames, in the reference's float64 (its landmark coordinates are Python floats):
//   mode 0  important_landmarks.py:131-133  |y[bottom] - y[top]| / (dist2d(eye_l, eye_r) + 1e-6), dist2d = :64-67
//   mode 1  inactive/live_test_5.py:92-94   max(y) - min(y) over the first K landmarks of the row
// lm (n, K, 2) f32 normalised landmarks (x, y).  Products and the sum are rounded one by one (no FMA) as Python does;
// the reference takes the root with ``** 0.5`` (libm pow), this kernel with the correctly rounded sqrt: they agree to one
// unit in the last place of a double (measured on 3e5 random spans: 0.09 % differ, by exactly one ulp).
__global__ __launch_bounds__(256) void mouth_openness_kernel(const float* __restrict__ lm, int n, int K, int mode, int i_top,
                                                             int i_bot, int i_eye_l, int i_eye_r, double* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* f = lm + (long)i * K * 2;
  if (mode == 0) {
    const double gap = fabs((double)f[2 * i_bot + 1] - (double)f[2 * i_top + 1]);
    const double dx = (double)f[2 * i_eye_l] - (double)f[2 * i_eye_r], dy = (double)f[2 * i_eye_l + 1] - (double)f[2 * i_eye_r + 1];
    const double span = __dadd_rn(__dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy))), 1e-6);
    out[i] = __ddiv_rn(gap, span);
  } else if (mode == 2) {
    // inactive/live_feed.py:69-78 on float32 landmark arrays: np.linalg.norm of a 2-vector is sqrt(x*x + y*y) with every
    // operation rounded to float32 (no FMA: pinned by tests/golden/serving.npz), the width gets + 1e-6 in float32
    const float gx = __fsub_rn(f[2 * i_top], f[2 * i_bot]), gy = __fsub_rn(f[2 * i_top + 1], f[2 * i_bot + 1]);
    const float wx = __fsub_rn(f[2 * i_eye_r], f[2 * i_eye_l]), wy = __fsub_rn(f[2 * i_eye_r + 1], f[2 * i_eye_l + 1]);
    const float gap = ss_sqrt_rn_f32(__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy)));
    const float wid = __fadd_rn(ss_sqrt_rn_f32(__fadd_rn(__fmul_rn(wx, wx), __fmul_rn(wy, wy))), 1e-6f);
    out[i] = (double)__fdiv_rn(gap, wid);
  } else {
    float lo = f[1], hi = f[1];
    for (int k = 1; k < K; ++k) {
      lo = fminf(lo, f[2 * k + 1]);
      hi = fmaxf(hi, f[2 * k + 1]);
    }
    out[i] = (double)hi - (double)lo;
  }
}

// EMA + hysteresis on the mouth openness of the pushed streams (important_landmarks.py:136-144).  ``mouth_ema`` is a
// Python float there: float64, two products and one sum, each rounded (no FMA).
__global__ __launch_bounds__(256) void mouth_gate_kernel(const int32_t* __restrict__ ids, int n, const double* __restrict__ openness,
                                                         double alpha, double open_thr, double close_thr, double* __restrict__ ema,
                                                         uint8_t* __restrict__ state_open) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int s = ids[i];
  const double e = __dadd_rn(__dmul_rn(1.0 - alpha, ema[s]), __dmul_rn(alpha, openness[i]));
  ema[s] = e;
  if (state_open[s]) {
    if (e < close_thr) state_open[s] = 0;
  } else if (e > open_thr) {
    state_open[s] = 1;
  }
}

// Openness-gated clip segmentation (inactive/live_test_5.py:146-152 constants, :233-272 state machine; "NO FACE" reset
// :293-301).  One workgroup per pushed frame: thread 0 advances the stream's state
//   state[s] = {speaking, above_ct, below_ct, clip_len}
// then all threads append the frame's feature row (and ROI bytes) to the stream's clip buffer when the machine says so.
// append_row[i] = row of the clip buffer this frame went to (-1: not appended); emit_len[i] = frames of the clip that
// ended with this frame when it has at least min_clip of them (else 0): those clips go to the classifier.
__global__ __launch_bounds__(256) void clip_gate_kernel(const int32_t* __restrict__ ids, const double* __restrict__ openv,
                                                        const uint8_t* __restrict__ face, double open_thresh, int start_n, int end_n,
                                                        int max_clip, int min_clip, int32_t* __restrict__ state, int D, int chunks,
                                                        const float* __restrict__ feats, const uint8_t* __restrict__ rois,
                                                        float* __restrict__ clip_x, uint8_t* __restrict__ clip_r,
                                                        int32_t* __restrict__ append_row, int32_t* __restrict__ emit_len) {
  __shared__ int sh_row;
  const int i = blockIdx.x, s = ids[i];
  if (threadIdx.x == 0) {
    int32_t* st = state + 4 * (long)s;
    int speaking = st[0], above = st[1], below = st[2], len = st[3];
    int row = -1, emit = 0;
    if (face && !face[i]) {  // no face in this frame: everything is dropped
      speaking = 0; above = below = 0; len = 0;
    } else {
      if (openv[i] > open_thresh) { above += 1; below = 0; } else { below += 1; above = 0; }
      if (!speaking) {
        if (above >= start_n) { speaking = 1; len = 0; above = below = 0; }
      } else {
        row = len;
        len += 1;
        if (below >= end_n || len >= max_clip) {
          speaking = 0; above = below = 0;
          if (len >= min_clip) emit = len;
        }
      }
    }
    st[0] = speaking; st[1] = above; st[2] = below; st[3] = len;
    append_row[i] = row;
    emit_len[i] = emit;
    sh_row = row;
  }
  __syncthreads();
  const int row = sh_row;
  if (row < 0) return;
  float* dx = clip_x + ((long)s * max_clip + row) * D;
  const float* sx = feats + (long)i * D;
  for (int d = threadIdx.x; d < D; d += 256) dx[d] = sx[d];
  if (clip_r) {
    uint4* dr = reinterpret_cast<uint4*>(clip_r) + ((long)s * max_clip + row) * chunks;
    const uint4* sr = reinterpret_cast<const uint4*>(rois) + (long)i * chunks;
    for (int c = threadIdx.x; c < chunks; c += 256) dr[c] = sr[c];
  }
}

// a camera frame without a face: the frame counter runs on, the buffer keeps what it holds (inactive/live_feed.py:173, 179-185)
__global__ void ring_tick_kernel(const int32_t* __restrict__ ids, int n, int32_t* __restrict__ frames_seen) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) frames_seen[ids[k]] += 1;
}

}  // namespace

extern "C" int ss_ring_tick(const int32_t* stream_ids, int n, int32_t* frames_seen, ss_stream_t stream) {
  SS_REQUIRE(stream_ids && frames_seen && n > 0, SS_ERR_ARG);
  hipLaunchKernelGGL(ring_tick_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), stream_ids, n,
                     frames_seen);
  return ss_launch_status();
}

extern "C" int ss_ring_push(float* ring_x, uint8_t* ring_r, int n_streams, int max_t, int D, int frame_bytes,
                            const int32_t* stream_ids, int n, const float* feats, const uint8_t* rois, int32_t* head,
                            int32_t* count, int32_t* frames_seen, ss_stream_t stream) {
  SS_REQUIRE(ring_x && stream_ids && feats && head && count && frames_seen, SS_ERR_ARG);
  SS_REQUIRE(n_streams > 0 && max_t > 0 && D > 0 && n > 0 && n <= n_streams, SS_ERR_ARG);
  SS_REQUIRE((ring_r == nullptr) == (rois == nullptr), SS_ERR_ARG);
  SS_REQUIRE(!ring_r || (frame_bytes > 0 && (frame_bytes & 15) == 0), SS_ERR_UNSUPPORTED);
  hipLaunchKernelGGL(ring_push_kernel, dim3(n), dim3(256), 0, static_cast<hipStream_t>(stream), ring_x, ring_r, max_t, D,
                     frame_bytes / 16, stream_ids, feats, rois, head, count, frames_seen);
  return ss_launch_status();
}

extern "C" int ss_ring_window_map(const int32_t* stream_ids, int n, int max_t, const int32_t* head, const int32_t* count,
                                  int32_t* frame_map, int64_t* lengths, ss_stream_t stream) {
  SS_REQUIRE(stream_ids && head && count && frame_map && lengths && n > 0 && max_t > 0, SS_ERR_ARG);
  hipLaunchKernelGGL(ring_window_map_kernel, dim3(ceil_div(n * max_t, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     stream_ids, n, max_t, head, count, frame_map, lengths);
  return ss_launch_status();
}

extern "C" int ss_mouth_openness(const float* lm, int n, int K, int mode, int i_top, int i_bot, int i_eye_l, int i_eye_r,
                                 double* openness, ss_stream_t stream) {
  SS_REQUIRE(lm && openness && n > 0 && K > 0 && mode >= 0 && mode <= 2, SS_ERR_ARG);
  if (mode != 1) SS_REQUIRE(i_top >= 0 && i_top < K && i_bot >= 0 && i_bot < K && i_eye_l >= 0 && i_eye_l < K && i_eye_r >= 0 && i_eye_r < K, SS_ERR_ARG);
  hipLaunchKernelGGL(mouth_openness_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), lm, n, K, mode,
                     i_top, i_bot, i_eye_l, i_eye_r, openness);
  return ss_launch_status();
}

extern "C" int ss_mouth_gate(const int32_t* stream_ids, int n, const double* openness, double alpha, double open_thr,
                             double close_thr, double* ema, uint8_t* state_open, ss_stream_t stream) {
  SS_REQUIRE(stream_ids && openness && ema && state_open && n > 0, SS_ERR_ARG);
  hipLaunchKernelGGL(mouth_gate_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), stream_ids, n,
                     openness, alpha, open_thr, close_thr, ema, state_open);
  return ss_launch_status();
}

extern "C" int ss_clip_gate(const int32_t* stream_ids, int n, const double* openv, const uint8_t* face_present, double open_thresh,
                            int start_n, int end_n, int max_clip, int min_clip, int32_t* state, int D, int frame_bytes,
                            const float* feats, const uint8_t* rois, float* clip_x, uint8_t* clip_r, int32_t* append_row,
                            int32_t* emit_len, ss_stream_t stream) {
  SS_REQUIRE(stream_ids && openv && state && feats && clip_x && append_row && emit_len, SS_ERR_ARG);
  SS_REQUIRE(n > 0 && D > 0 && start_n > 0 && end_n > 0 && max_clip > 0 && min_clip >= 0, SS_ERR_ARG);
  SS_REQUIRE((clip_r == nullptr) == (rois == nullptr), SS_ERR_ARG);
  SS_REQUIRE(!clip_r || (frame_bytes > 0 && (frame_bytes & 15) == 0), SS_ERR_UNSUPPORTED);
  hipLaunchKernelGGL(clip_gate_kernel, dim3(n), dim3(256), 0, static_cast<hipStream_t>(stream), stream_ids, openv, face_present,
                     open_thresh, start_n, end_n, max_clip, min_clip, state, D, frame_bytes / 16, feats, rois, clip_x, clip_r,
                     append_row, emit_len);
  return ss_launch_status();
}
