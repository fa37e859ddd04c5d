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

// EMA + hysteresis on the mouth openness of the pushed streams (important_landmarks.py:136-144)
__global__ __launch_bounds__(256) void mouth_gate_kernel(const int32_t* __restrict__ ids, int n, const float* __restrict__ openness,
                                                         float alpha, float open_thr, float close_thr, float* __restrict__ ema,
                                                         uint8_t* __restrict__ state_open) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int s = ids[i];
  const float e = __fadd_rn(__fmul_rn(1.0f - alpha, ema[s]), __fmul_rn(alpha, openness[i]));  // no FMA: the Python restatement rounds twice
  ema[s] = e;
  if (state_open[s]) {
    if (e < close_thr) state_open[s] = 0;
  } else if (e > open_thr) {
    state_open[s] = 1;
  }
}

}  // namespace

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

extern "C" int ss_mouth_gate(const int32_t* stream_ids, int n, const float* openness, float alpha, float open_thr,
                             float close_thr, float* ema, uint8_t* state_open, ss_stream_t stream) {
  SS_REQUIRE(stream_ids && openness && ema && state_open && n > 0, SS_ERR_ARG);
  hipLaunchKernelGGL(mouth_gate_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), stream_ids, n,
                     openness, alpha, open_thr, close_thr, ema, state_open);
  return ss_launch_status();
}
