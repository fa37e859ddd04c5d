// Flat-bucket optimiser step: global-norm clip folded into Adam, plus the small streaming helpers
// (row copy, column sums).  Replaces clip_grad_norm_ + torch.optim.Adam.step of
// /root/reference/train_model_official.py:403, 438-439.  All HBM-bound: 16-byte lanes, grid-stride.
#include <math.h>

#include "ss_common.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  const long n4 = n >> 2;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  // four 16-byte loads in flight per thread (a dependent chain of single loads made the 28 MB bucket of config 5 a 28 us kernel:
  // 1 TB/s; one launch's atomics -- one per workgroup on one address, ~12 ns each -- bound the workgroup count from above)
  const long stride = (long)gridDim.x * 256;
  long q = (long)blockIdx.x * 256 + threadIdx.x;
  for (; q + 3 * stride < n4; q += 4 * stride) {
    const f32x4 v0 = x4[q], v1 = x4[q + stride], v2 = x4[q + 2 * stride], v3 = x4[q + 3 * stride];
    acc += (v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2] + v0[3] * v0[3]) + (v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2] + v1[3] * v1[3]) +
           (v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2] + v2[3] * v2[3]) + (v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2] + v3[3] * v3[3]);
  }
  for (; q < n4; q += stride) {
    const f32x4 v = x4[q];
    acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    float v = x[(n4 << 2) + threadIdx.x];
    acc += v * v;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

struct AdamParams {
  float* p;
  const float* g;
  float* m;
  float* v;
  long n;
  const float* sumsq;
  float grad_scale, max_norm, step_size, beta1, beta2, eps, inv_sqrt_bc2;
};

__global__ __launch_bounds__(256) void adam_clip_kernel(AdamParams a) {
  const float total = sqrtf(a.sumsq[0]) * a.grad_scale;
  const float coef = a.grad_scale * fminf(1.0f, a.max_norm / (total + 1e-6f));
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < a.n; q += (long)gridDim.x * 256) {
    const float g = a.g[q] * coef;
    const float m = a.beta1 * a.m[q] + (1.0f - a.beta1) * g;
    const float v = a.beta2 * a.v[q] + (1.0f - a.beta2) * g * g;
    a.m[q] = m;
    a.v[q] = v;
    const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
    a.p[q] -= a.step_size * (m / denom);
  }
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst,
                                                        int ld_dst, int rows, int cols) {
  const long total = (long)rows * cols;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
    int r = (int)(q / cols), c = (int)(q % cols);
    dst[(long)r * ld_dst + c] = src[(long)r * ld_src + c];
  }
}

// The rows (b, t) of a padded batch that belong to a clip: frames[0] = how many, frames[1 ...] = their numbers b*T + t in ascending
// order; `emb` columns [0, E) of every OTHER row are cleared (the ROI CNN skips those frames -- the packed recurrence never reads
// their rows, pack_padded_sequence at train_model_official.py:300, but the input-projection GEMM does and must find numbers there).
// Block `blk` of `nblk` takes a contiguous range of clips; what lies in front of it is summed by the block itself (B loads).
template <class LenT>
__device__ void active_frames_block(const LenT* __restrict__ len, int B, int T, int* __restrict__ frames, float* __restrict__ emb,
                                    int ld, int E, int blk, int nblk) {
  __shared__ int s_part[4];
  const int chunk = (B + nblk - 1) / nblk, b0 = blk * chunk, b1 = min(B, b0 + chunk);
  if (b0 >= B) return;
  const int tid = threadIdx.x;
  int part = 0;
  for (int b = tid; b < b0; b += 256) part += min(max((int)len[b], 0), T);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if ((tid & 63) == 0) s_part[tid >> 6] = part;
  __syncthreads();
  int base = s_part[0] + s_part[1] + s_part[2] + s_part[3];
  for (int b = b0; b < b1; ++b) {
    const int l = min(max((int)len[b], 0), T);
    for (int t = tid; t < T; t += 256) {
      const int row = b * T + t;
      if (t < l) frames[1 + base + t] = row;
      else if (emb)
        for (int e = 0; e < E; ++e) emb[(long)row * ld + e] = 0.f;
    }
    base += l;
  }
  if (b1 == B && tid == 0) frames[0] = base;
}

__global__ __launch_bounds__(256) void active_frames_kernel(const int32_t* __restrict__ len, int B, int T, int* __restrict__ frames,
                                                            float* __restrict__ emb, int ld, int E) {
  active_frames_block(len, B, T, frames, emb, ld, E, blockIdx.x, gridDim.x);
}

// Everything a training step does before its first real kernel, in ONE launch (six tiny launches cost ~6 us each on the
// step's critical path): clear the gradient bucket, clear the loss / sum-of-squares scalars and the hit counter, narrow the
// int64 lengths to the int32 the kernels read, and copy the landmark features into their columns of the GRU input.
struct ProloguePar {
  float* grads; long n_grads;
  float* scal; int n_scal;
  int32_t* correct;
  const int64_t* len64; int32_t* len32; int B;
  const float* X; int ld_x; float* Z; int ld_z; int rows, cols;
  int* frames; int E;  // the list of frames that belong to a clip (active_frames_block), from the int64 lengths; null: skipped
};
__global__ __launch_bounds__(256) void train_prologue_kernel(ProloguePar a) {
  const long tid = (long)blockIdx.x * 256 + threadIdx.x, nthreads = (long)gridDim.x * 256;
  f32x4* g4 = reinterpret_cast<f32x4*>(a.grads);
  const long n4 = a.n_grads >> 2;
  for (long q = tid; q < n4; q += nthreads) g4[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (long q = (n4 << 2) + tid; q < a.n_grads; q += nthreads) a.grads[q] = 0.f;
  if (a.X) {
    const long total = (long)a.rows * a.cols;
    for (long q = tid; q < total; q += nthreads) {
      const int r = (int)(q / a.cols), c = (int)(q - (long)r * a.cols);
      a.Z[(long)r * a.ld_z + c] = a.X[(long)r * a.ld_x + c];
    }
  }
  if (blockIdx.x == 0) {
    if ((int)threadIdx.x < a.n_scal) a.scal[threadIdx.x] = 0.f;
    if (threadIdx.x == 0 && a.correct) a.correct[0] = 0;
    if (a.len64)
      for (int b = threadIdx.x; b < a.B; b += 256) a.len32[b] = (int32_t)a.len64[b];
  }
  if (a.frames) active_frames_block(a.len64, a.B, a.rows / a.B, a.frames, a.Z + a.cols, a.ld_z, a.E, blockIdx.x, gridDim.x);
}

// clears two float buffers (16-byte aligned) in one launch: the atomically summed destinations of the d layer_in GEMMs
__global__ __launch_bounds__(256) void zero2_kernel(float* __restrict__ a, long na, float* __restrict__ b, long nb) {
  const long tid = (long)blockIdx.x * 256 + threadIdx.x, nthreads = (long)gridDim.x * 256;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4* a4 = reinterpret_cast<f32x4*>(a);
  for (long q = tid; q < (na >> 2); q += nthreads) a4[q] = z;
  for (long q = (na & ~3L) + tid; q < na; q += nthreads) a[q] = 0.f;
  f32x4* b4 = reinterpret_cast<f32x4*>(b);
  for (long q = tid; q < (nb >> 2); q += nthreads) b4[q] = z;
  for (long q = (nb & ~3L) + tid; q < nb; q += nthreads) b[q] = 0.f;
}

// grid (ceil(cols/64), row chunks); thread (c = tid&63, rr = tid>>6) strides rows by 4
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ A, int rows, int cols, int lda,
                                                     int rows_per_block, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rr = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float acc = 0.f;
  if (c < cols)
    for (int r = r0 + rr; r < r1; r += 4) acc += A[(long)r * lda + c];
  red[rr][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rr == 0 && c < cols) atomicAdd(&out[c], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// Tall and narrow (millions of rows x 8..64 packed columns: the per-pixel gradients of the layer-by-layer CNN): the matrix as one
// flat stream, a thread's stride a multiple of `cols` so that it stays on one column, four loads in flight, one LDS fold and
// `cols` atomics per workgroup.  (colsum_kernel gives a column a lane and a block 32 rows: 8 of 64 lanes busy and 2.2 M
// workgroups adding onto the same 8 addresses for a 70 M x 8 matrix -- 37 ms.)
__global__ __launch_bounds__(256) void colsum_flat_kernel(const float* __restrict__ A, long total, int cols, int tpb,
                                                          float* __restrict__ out) {
  __shared__ float red[256];
  const int t = threadIdx.x;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (t < tpb) {
    const long stride = (long)gridDim.x * tpb;
    long i = (long)blockIdx.x * tpb + t;
    for (; i + 3 * stride < total; i += 4 * stride) {
      a0 += A[i]; a1 += A[i + stride]; a2 += A[i + 2 * stride]; a3 += A[i + 3 * stride];
    }
    for (; i < total; i += stride) a0 += A[i];
  }
  red[t] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (t < cols) {
    float s = 0.f;
    for (int k = t; k < tpb; k += cols) s += red[k];
    atomicAdd(&out[t], s);
  }
}

// GRU bias gradients of one layer, both directions, from dG (2, N, 4, H):
//   d b_ih[dir] += colsum(dG[dir][:, 0:3H]);  d b_hh[dir] += colsum(dG[dir][:, 0:2H] | dG[dir][:, 3H:4H])
// grid (4H/64, row chunks, 2); thread (c = tid&63, rr = tid>>6) strides rows by 4
__global__ __launch_bounds__(256) void gru_bias_grad_kernel(const float* __restrict__ dG, int N, int H, int rows_per_block,
                                                            float* __restrict__ gbi0, float* __restrict__ gbh0,
                                                            float* __restrict__ gbi1, float* __restrict__ gbh1) {
  __shared__ float red[4][64];
  const int dir = blockIdx.z;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rr = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(N, r0 + rows_per_block);
  const float* A = dG + (long)dir * N * 4 * H;
  float acc = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;  // four loads in flight per lane
  int r = r0 + rr;
  for (; r + 12 < r1; r += 16) {
    const float* a = A + (long)r * 4 * H + c;
    acc += a[0];
    acc1 += a[(long)16 * H];
    acc2 += a[(long)32 * H];
    acc3 += a[(long)48 * H];
  }
  for (; r < r1; r += 4) acc += A[(long)r * 4 * H + c];
  acc += acc1 + acc2 + acc3;
  red[rr][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rr == 0) {
    const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    float* gbi = dir ? gbi1 : gbi0;
    float* gbh = dir ? gbh1 : gbh0;
    if (c < 3 * H) atomicAdd(&gbi[c], s);
    if (c < 2 * H) atomicAdd(&gbh[c], s);
    else if (c >= 3 * H) atomicAdd(&gbh[c - H], s);
  }
}

}  // namespace

extern "C" int ss_gru_bias_grad(const float* d_g, int N, int H, float* g_bih_f, float* g_bhh_f, float* g_bih_r,
                                float* g_bhh_r, ss_stream_t stream) {
  SS_REQUIRE(d_g && g_bih_f && g_bhh_f && g_bih_r && g_bhh_r && N > 0 && H > 0 && (H % 16) == 0, SS_ERR_ARG);
  const int rpb = 128;
  dim3 grid(4 * H / 64, ceil_div(N, rpb), 2);
  hipLaunchKernelGGL(gru_bias_grad_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), d_g, N, H, rpb, g_bih_f,
                     g_bhh_f, g_bih_r, g_bhh_r);
  return ss_launch_status();
}

extern "C" int ss_sumsq_f32(const float* x, long n, float* sumsq, ss_stream_t stream) {
  SS_REQUIRE(x && sumsq && n > 0, SS_ERR_ARG);
  SS_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, SS_ERR_ARG);
  // every block ends in one float atomic on the same word (~11 ns each, serialised): 128 blocks, not 1024
  int blocks = (int)(((n >> 2) + 255) / 256);
  blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, n, sumsq);
  return ss_launch_status();
}

extern "C" int ss_adam_clip(float* p, const float* g, float* m, float* v, long n, const float* sumsq,
                            float grad_scale, float max_norm, float lr, float beta1, float beta2, float eps, int step,
                            ss_stream_t stream) {
  SS_REQUIRE(p && g && m && v && sumsq && n > 0 && step >= 1, SS_ERR_ARG);
  AdamParams a;
  a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.sumsq = sumsq;
  a.grad_scale = grad_scale; a.max_norm = max_norm;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  a.step_size = (float)((double)lr / bc1);
  a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  int blocks = (int)((n + 255) / 256);
  blocks = blocks > 2048 ? 2048 : blocks;
  hipLaunchKernelGGL(adam_clip_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return ss_launch_status();
}

extern "C" int ss_copy_rows_f32(const float* src, int ld_src, float* dst, int ld_dst, int rows, int cols,
                                ss_stream_t stream) {
  SS_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= cols, SS_ERR_ARG);
  long total = (long)rows * cols;
  int blocks = (int)((total + 255) / 256);
  blocks = blocks > 2048 ? 2048 : blocks;
  hipLaunchKernelGGL(copy_rows_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, ld_src, dst,
                     ld_dst, rows, cols);
  return ss_launch_status();
}

extern "C" int ss_train_prologue(float* grads, long n_grads, float* scalars, int n_scalars, int32_t* correct,
                                 const int64_t* lengths64, int32_t* lengths32, int B, const float* X, int ld_x, float* Z,
                                 int ld_z, int rows, int cols, int* frames, int emb_cols, ss_stream_t stream) {
  SS_REQUIRE(grads && n_grads > 0 && (reinterpret_cast<uintptr_t>(grads) & 15) == 0, SS_ERR_ARG);
  SS_REQUIRE(n_scalars >= 0 && n_scalars <= 256 && (n_scalars == 0 || scalars), SS_ERR_ARG);
  SS_REQUIRE(!lengths64 || (lengths32 && B > 0), SS_ERR_ARG);
  SS_REQUIRE(!X || (Z && rows > 0 && cols > 0 && ld_x >= cols && ld_z >= cols), SS_ERR_ARG);
  SS_REQUIRE(!frames || (lengths64 && X && emb_cols >= 0 && ld_z >= cols + emb_cols && rows % B == 0), SS_ERR_ARG);
  ProloguePar a;
  a.grads = grads; a.n_grads = n_grads; a.scal = scalars; a.n_scal = n_scalars; a.correct = correct;
  a.len64 = lengths64; a.len32 = lengths32; a.B = B;
  a.X = X; a.ld_x = ld_x; a.Z = Z; a.ld_z = ld_z; a.rows = rows; a.cols = cols;
  a.frames = frames; a.E = emb_cols;
  long work = (n_grads >> 2) > (X ? (long)rows * cols : 0) ? (n_grads >> 2) : (long)rows * cols;
  int blocks = (int)((work + 255) / 256);
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipLaunchKernelGGL(train_prologue_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return ss_launch_status();
}

extern "C" int ss_zero_f32x2(float* a, long na, float* b, long nb, ss_stream_t stream) {
  SS_REQUIRE(na >= 0 && nb >= 0 && (na == 0 || a) && (nb == 0 || b), SS_ERR_ARG);
  SS_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0, SS_ERR_ARG);
  if (na + nb == 0) return SS_OK;
  long blocks = ((na > nb ? na : nb) / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(zero2_kernel, dim3((int)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a, na, b, nb);
  return ss_launch_status();
}

extern "C" int ss_colsum_f32(const float* A, int rows, int cols, int lda, float* out, ss_stream_t stream) {
  SS_REQUIRE(A && out && rows > 0 && cols > 0 && lda >= cols, SS_ERR_ARG);
  if (lda == cols && cols <= 64 && (long)rows * cols >= (1L << 20)) {
    const int tpb = 256 - 256 % cols;  // threads that work: a multiple of cols
    const long total = (long)rows * cols;
    long blocks = total / ((long)tpb * 16);
    const long cap = 8L * ss_device_cus();
    blocks = blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
    hipLaunchKernelGGL(colsum_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), A, total, cols, tpb, out);
    return ss_launch_status();
  }
  int rpb = 32;  // 256 rows in one block is a chain of 16 memory latencies (18 us for a 256 x 384 sum); 8 blocks meet in atomics
  while (ceil_div(rows, rpb) > 32768) rpb *= 2;  // (and the grid's y dimension has a limit)
  dim3 grid(ceil_div(cols, 64), ceil_div(rows, rpb));
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), A, rows, cols, lda, rpb, out);
  return ss_launch_status();
}

extern "C" int ss_roi_active_frames(const int32_t* lengths, int B, int T, int* frames, float* emb, int ld_emb, int emb_cols,
                                    ss_stream_t stream) {
  SS_REQUIRE(lengths && frames && B > 0 && T > 0, SS_ERR_ARG);
  SS_REQUIRE((long)B * T < (1L << 31), SS_ERR_UNSUPPORTED);
  SS_REQUIRE(!emb || (emb_cols > 0 && ld_emb >= emb_cols), SS_ERR_ARG);
  const int blocks = B < 256 ? B : 256;
  hipLaunchKernelGGL(active_frames_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), lengths, B, T, frames, emb,
                     ld_emb, emb_cols);
  return ss_launch_status();
}

extern "C" int ss_abi_version(void) { return 3; }

extern "C" const char* ss_status_string(int status) {
  switch (status) {
    case SS_OK: return "ok";
    case SS_ERR_ARG: return "invalid argument";
    case SS_ERR_LAUNCH: return "kernel launch failed";
    case SS_ERR_UNSUPPORTED: return "unsupported shape";
    default: return "unknown status";
  }
}
