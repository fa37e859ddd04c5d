// Diagnostic micro-benchmark: per-step cost of an all-gather between P workgroups (one per CU) through global memory,
// the exchange a recurrence split over several CUs would need.  Every step each workgroup publishes its share of a
// 16 x 192 float panel as (value, step) pairs and then every wave polls the whole panel until all stamps match.
//   hipcc --offload-arch=gfx950 -O3 -o xcu_sync xcu_sync.hip && ./xcu_sync
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SCOPE>
__device__ __forceinline__ unsigned long long ld(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE);
}
template <int SCOPE>
__device__ __forceinline__ void st(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE);
}

// panel: [2 parity][G groups][3072 pairs]
template <int SCOPE>
__global__ __launch_bounds__(256) void bench(unsigned long long* panel, int G, int P, int steps, int mfmas, long long* cyc,
                                             int* err, float* sink) {
  const int grp = blockIdx.x % G, part = blockIdx.x / G;
  const int tid = threadIdx.x, lane = tid & 63;
  const int share = 3072 / P;  // pairs this workgroup publishes per step
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float carry = 1.0f + part;
  long long t0 = wall_clock64();
  for (int t = 1; t <= steps; ++t) {
    unsigned long long* pan = panel + ((size_t)(t & 1) * G + grp) * 3072;
    for (int q = tid; q < share; q += 256) {
      unsigned long long w = ((unsigned long long)(unsigned)t << 32) | __float_as_uint(carry + q);
      st<SCOPE>(&pan[part * share + q], w);
    }
    // every wave gathers the whole panel: 48 pairs per lane
    float sum = 0.f;
    unsigned long long v[48];
    int spins = 0;
    bool ok, dead = false;
    do {
      ok = true;
#pragma unroll
      for (int j = 0; j < 48; ++j) v[j] = ld<SCOPE>(&pan[j * 64 + lane]);
#pragma unroll
      for (int j = 0; j < 48; ++j) ok &= (unsigned)(v[j] >> 32) == (unsigned)t;
      ok = __all(ok);
      if (!ok && ((++spins & 1023) == 0)) {  // bounded spin: a lost partner ends the kernel instead of hanging it
        if (spins > 2000000 && lane == 0) atomicAdd(err, 1);
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { dead = true; break; }
      }
    } while (!ok);
    if (dead) break;
#pragma unroll
    for (int j = 0; j < 48; ++j) sum += __uint_as_float((unsigned)v[j]);
    // stand-in for the step's matrix work
    for (int m = 0; m < mfmas; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sum, 1.0f, acc, 0, 0, 0);
    carry = acc[0] * 1e-30f + 1.0f;
  }
  long long t1 = wall_clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc[1] == 123.f) sink[0] = acc[1];
}

int main() {
  const int G = 32, steps = 200;
  unsigned long long* panel;
  long long* cyc;
  int* err;
  float* sink;
  (void)hipMalloc(&panel, sizeof(unsigned long long) * 2 * G * 3072);
  (void)hipMalloc(&cyc, sizeof(long long) * 1024);
  (void)hipMalloc(&err, sizeof(int));
  (void)hipMalloc(&sink, 4);
  int rate_khz = 0;
  (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
  printf("wall clock %d kHz\n", rate_khz);
  for (int scope = 0; scope < 2; ++scope)
    for (int P : {1, 2, 3, 4, 6, 12})
      for (int mfmas : {0, 72, 144}) {
        (void)hipMemset(panel, 0, sizeof(unsigned long long) * 2 * G * 3072);
        (void)hipMemset(err, 0, sizeof(int));
        if (scope == 0)
          hipLaunchKernelGGL(bench<__HIP_MEMORY_SCOPE_AGENT>, dim3(G * P), dim3(256), 0, 0, panel, G, P, steps, mfmas, cyc, err, sink);
        else
          hipLaunchKernelGGL(bench<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(G * P), dim3(256), 0, 0, panel, G, P, steps, mfmas, cyc, err,
                             sink);
        hipError_t e = hipDeviceSynchronize();
        std::vector<long long> h(G * P);
        int herr = 0;
        (void)hipMemcpy(h.data(), cyc, sizeof(long long) * G * P, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto x : h) mean += x;
        mean /= h.size();
        printf("scope=%s P=%2d mfma/step=%3d : %.2f us/step  (err=%d, %s)\n", scope ? "workgroup" : "agent", P, mfmas,
               mean / steps / (rate_khz * 1e-3), herr, hipGetErrorString(e));
        fflush(stdout);
      }
  return 0;
}
