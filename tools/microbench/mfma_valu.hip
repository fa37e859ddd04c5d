// Microbenchmark: does the vector-ALU work of one wave overlap the f32 MFMAs (v_mfma_f32_16x16x4_f32) of the OTHER wave on
// the same SIMD?  One 512-thread workgroup per CU (two waves per SIMD): waves 0..3 run MFMAs, waves 4..7 run VALU / LDS work.
//   mode 0: MFMA waves only      mode 1: VALU waves only      mode 2: both      mode 3: both, every wave does both (lockstep)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/mfma_valu tools/microbench/mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(int mode, int iters, float* out, unsigned long long* cyc) {
  __shared__ float lds[4096];
  const int wv = threadIdx.x >> 6;
  for (int q = threadIdx.x; q < 4096; q += 512) lds[q] = 1.0f;
  __syncthreads();
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  const float a = 1.0f + threadIdx.x, b = 0.5f;
  const bool do_m = (mode == 0 && wv < 4) || (mode == 2 && wv < 4) || mode == 3;
  const bool do_v = (mode == 1 && wv >= 4) || (mode == 2 && wv >= 4) || mode == 3;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (do_m) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u & 3], 0, 0, 0);
    }
    if (do_v) {
#pragma unroll
      for (int u = 0; u < 32; ++u) v[u & 7] = v[u & 7] * 1.0001f + 0.5f;   // 32 independent-ish v_fma
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = clock64();
  float s = 0;
  for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][1];
  for (int u = 0; u < 8; ++u) s += v[u];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  unsigned long long h[256 * 8];
  const int iters = 2000;
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) k<<<256, 512>>>(mode, iters, out, cyc);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int b = 0; b < 256; ++b) { for (int w = 0; w < 4; ++w) m += h[b * 8 + w]; for (int w = 4; w < 8; ++w) v += h[b * 8 + w]; }
    printf("mode %d: MFMA waves %.1f cycles/iter (8 MFMAs = 256 alone), VALU waves %.1f cycles/iter (32 v_fma = 128 alone)\n", mode,
           m / 1024 / iters, v / 1024 / iters);
  }
  return 0;
}
