// Microbenchmark: what do the two waves of a SIMD cost each other?  One 512-thread workgroup per CU = two waves per SIMD
// (wave w and wave w + 4 share SIMD w).  Wave class A = waves 0..3, class B = waves 4..7; each class runs one of
//   M  a stream of v_mfma_f32_16x16x4_f32 (two accumulator chains)        V  a stream of independent v_fma_f32
//   L  a stream of conflict-free ds_read_b32                               -  nothing
// and the table prints cycles per instruction of each class (clock64 around the loop).  Alone: M = 32, V = 4 (wave64 on a
// 16-lane SIMD), L = 2 (128 B per clock and CU, shared by the four SIMDs -> 8 per wave when all four read).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/issue_model tools/microbench/issue_model.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>  // 0 none, 1 MFMA, 2 VALU, 3 LDS
__device__ __forceinline__ float body(int iters, const float* lds, int prio) {
  f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  const float a = 1.0f + threadIdx.x, b = 0.5f;
  if (prio) __builtin_amdgcn_s_setprio(2);
  for (int it = 0; it < iters; ++it) {
    if (KIND == 1) {
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u & 1], 0, 0, 0);
    } else if (KIND == 2) {
#pragma unroll
      for (int u = 0; u < 64; ++u) v[u & 7] = v[u & 7] * 1.0001f + 0.5f;
    } else if (KIND == 3) {
#pragma unroll
      for (int u = 0; u < 32; ++u) v[u & 7] += lds[(threadIdx.x & 63) + 64 * (u & 15) + (it & 1)];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_s_setprio(0);
  float s = acc[0][0] + acc[1][1];
  for (int u = 0; u < 8; ++u) s += v[u];
  return s;
}

__global__ __launch_bounds__(512) void k(int ka, int kb, int prio_a, int prio_b, int iters, float* out, unsigned long long* cyc) {
  __shared__ float lds[2048];
  const int wv = threadIdx.x >> 6;
  for (int q = threadIdx.x; q < 2048; q += 512) lds[q] = 1.0f;
  __syncthreads();
  const int kind = wv < 4 ? ka : kb;
  const int pr = wv < 4 ? prio_a : prio_b;
  const unsigned long long t0 = clock64();
  float s = 0;
  if (kind == 1) s = body<1>(iters, lds, pr);
  else if (kind == 2) s = body<2>(iters, lds, pr);
  else if (kind == 3) s = body<3>(iters, lds, pr);
  const unsigned long long t1 = clock64();
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  static unsigned long long h[256 * 8];
  const int iters = 2000;
  const char* nm = "-MVL";
  const int per[4] = {1, 16, 64, 32};
  const int cases[][4] = {{1, 0, 0, 0}, {2, 0, 0, 0}, {3, 0, 0, 0}, {1, 1, 0, 0}, {2, 2, 0, 0}, {3, 3, 0, 0}, {1, 2, 0, 0}, {1, 2, 1, 0}, {1, 2, 0, 1},
                          {1, 3, 0, 0}, {1, 3, 1, 0}, {1, 3, 0, 1}, {2, 3, 0, 0}};
  for (auto& c : cases) {
    for (int rep = 0; rep < 2; ++rep) k<<<256, 512>>>(c[0], c[1], c[2], c[3], iters, out, cyc);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int blk = 0; blk < 256; ++blk) { for (int w = 0; w < 4; ++w) a += h[blk * 8 + w]; for (int w = 4; w < 8; ++w) b += h[blk * 8 + w]; }
    printf("A=%c%s B=%c%s : A %.2f cycles/instr", nm[c[0]], c[2] ? "(prio 2)" : "", nm[c[1]], c[3] ? "(prio 2)" : "", a / 1024 / iters / per[c[0]]);
    if (c[1]) printf(", B %.2f cycles/instr", b / 1024 / iters / per[c[1]]);
    printf("\n");
  }
  return 0;
}
