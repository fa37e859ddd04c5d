// Shared device helpers for the Silent-Speech hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ss_hotpath.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SS_WAVE 64

// launch-site error handling: every entry point returns 0 or a negative ss_status
#define SS_REQUIRE(cond, code) \
  do {                         \
    if (!(cond)) return (code); \
  } while (0)

static inline int ss_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SS_OK : SS_ERR_LAUNCH;
}

// v_mfma_f32_16x16x4_f32: A[i=lane&15][k=lane>>4], B[k=lane>>4][j=lane&15],
// D[row=(lane>>4)*4+reg][col=lane&15].  Exact f32 fma chain.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Fence for the instruction scheduler: keeps a batch of ds_reads issued ahead of the MFMAs that consume them
// (left alone, hipcc sinks every read next to its MFMA and waits lgkmcnt(0) in between: one LDS latency per MFMA).
#define SS_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// v_exp_f32 + v_rcp_f32 (1 ulp each): the gates sit on the recurrence's critical path between two workgroup barriers,
// an IEEE divide there costs ~10 VALU instructions per element; the logit tolerance (1e-3) leaves 4 orders of margin
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// tanh via one exp; saturates cleanly (exp->inf gives 1, exp->0 gives -1)
__device__ __forceinline__ float tanh_f(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- diagnostic build only (-DSS_STAMP): per-stage cycle shares of the persistent kernels.
// Thread 0 of every workgroup accumulates s_memtime deltas between barriers; the real build has no stamps.
#ifdef SS_STAMP
#define SS_STAMP_SLOTS 16
// one table per translation unit (no relocatable device code): STAMP_TABLE(fn) defines it and its host reader
#define STAMP_TABLE(reader)                                                                           \
  __device__ unsigned long long ss_stamp_buf[256 * SS_STAMP_SLOTS];                                   \
  extern "C" int reader(unsigned long long* host_out) {                                               \
    if (hipDeviceSynchronize() != hipSuccess) return SS_ERR_LAUNCH;                                   \
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ss_stamp_buf), sizeof(unsigned long long) * 256 * SS_STAMP_SLOTS) == hipSuccess \
               ? SS_OK                                                                                \
               : SS_ERR_LAUNCH;                                                                       \
  }
#define STAMP_DECL unsigned long long st_last = clock64(), st_acc[SS_STAMP_SLOTS] = {0}
#define STAMP(k)                               \
  do {                                         \
    if (threadIdx.x == 0) {                    \
      unsigned long long t_ = clock64();       \
      st_acc[k] += t_ - st_last;               \
      st_last = t_;                            \
    }                                          \
  } while (0)
#define STAMP_SYNC(k) \
  do {                \
    __syncthreads();  \
    STAMP(k);         \
  } while (0)
// stamp after every outstanding memory operation of the wave has returned (delimits a load phase)
#define STAMP_WAIT(k)                \
  do {                               \
    __builtin_amdgcn_s_waitcnt(0);   \
    STAMP(k);                        \
  } while (0)
#define STAMP_FLUSH()                                                                       \
  do {                                                                                      \
    if (threadIdx.x == 0 && blockIdx.x < 256)                                               \
      for (int k_ = 0; k_ < SS_STAMP_SLOTS; ++k_) ss_stamp_buf[blockIdx.x * SS_STAMP_SLOTS + k_] = st_acc[k_]; \
  } while (0)
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_SYNC(k)
#define STAMP_WAIT(k)
#define STAMP_FLUSH()
#define STAMP_TABLE(reader)
#endif
