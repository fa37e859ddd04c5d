// Diagnostic micro-benchmark, second form of the cross-CU all-gather: data as plain floats (agent-scope 8-byte
// stores), one flag per (group, part) published after the data is acknowledged, consumers poll the P flags and then
// load the 16 x 192 panel once.  Also times a bare two-workgroup flag ping-pong.
//   hipcc --offload-arch=gfx950 -O3 -o xcu_sync2 xcu_sync2.hip && ./xcu_sync2
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define AG __HIP_MEMORY_SCOPE_AGENT

__global__ __launch_bounds__(64) void pingpong(unsigned* flags, int steps, long long* cyc, int* err) {
  const int me = blockIdx.x;  // 0 or 1 (x 8 apart in the grid so both land on one XCD, or adjacent for two XCDs)
  long long t0 = wall_clock64();
  if (threadIdx.x == 0) {
    for (int t = 1; t <= steps; ++t) {
      if ((t & 1) == me) {
        __hip_atomic_store(&flags[0], (unsigned)t, __ATOMIC_RELAXED, AG);
      } else {
        int spins = 0;
        while (__hip_atomic_load(&flags[0], __ATOMIC_RELAXED, AG) != (unsigned)t)
          if (++spins > 4000000) { atomicAdd(err, 1); t = steps; break; }
      }
    }
  }
  long long t1 = wall_clock64();
  if (threadIdx.x == 0) cyc[me] = t1 - t0;
}

// panel: [2 parity][G][3072 floats]; flags: [G][P]
__global__ __launch_bounds__(256) void gather(float* panel, unsigned* flags, int G, int P, int steps, int mfmas, int one_wave,
                                              long long* cyc, int* err, float* sink) {
  __shared__ float hs[3072];
  const int grp = blockIdx.x % G, part = blockIdx.x / G;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int share = 3072 / P;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float carry = 1.0f + part;
  bool dead = false;
  long long t0 = wall_clock64();
  for (int t = 1; t <= steps && !dead; ++t) {
    float* pan = panel + ((size_t)(t & 1) * G + grp) * 3072;
    for (int q = 2 * tid; q < share; q += 512) {
      unsigned long long w = ((unsigned long long)__float_as_uint(carry + q + 1) << 32) | __float_as_uint(carry + q);
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(&pan[part * share + q]), w, __ATOMIC_RELAXED, AG);
    }
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): the stores are acknowledged
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&flags[grp * P + part], (unsigned)t, __ATOMIC_RELAXED, AG);
    float sum = 0.f;
    if (!one_wave || wv == 0) {
      int spins = 0;
      bool ok;
      do {
        unsigned f = lane < P ? __hip_atomic_load(&flags[grp * P + lane], __ATOMIC_RELAXED, AG) : (unsigned)t;
        ok = __all(f == (unsigned)t);
        if (!ok && ((++spins & 1023) == 0)) {
          if (spins > 2000000 && lane == 0) atomicAdd(err, 1);
          if (__hip_atomic_load(err, __ATOMIC_RELAXED, AG)) { dead = true; break; }
        }
      } while (!ok);
      if (one_wave) {
#pragma unroll
        for (int j = 0; j < 24; ++j) {
          unsigned long long v = __hip_atomic_load(reinterpret_cast<unsigned long long*>(pan) + j * 64 + lane, __ATOMIC_RELAXED, AG);
          hs[2 * (j * 64 + lane)] = __uint_as_float((unsigned)v);
          hs[2 * (j * 64 + lane) + 1] = __uint_as_float((unsigned)(v >> 32));
        }
      } else {
        unsigned long long v[24];
#pragma unroll
        for (int j = 0; j < 24; ++j)
          v[j] = __hip_atomic_load(reinterpret_cast<unsigned long long*>(pan) + j * 64 + lane, __ATOMIC_RELAXED, AG);
#pragma unroll
        for (int j = 0; j < 24; ++j) sum += __uint_as_float((unsigned)v[j]) + __uint_as_float((unsigned)(v[j] >> 32));
      }
    }
    if (one_wave) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        f32x4 x = *reinterpret_cast<const f32x4*>(&hs[4 * (j * 64 + lane)]);
        sum += x[0] + x[1] + x[2] + x[3];
      }
    }
    for (int m = 0; m < mfmas; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sum, 1.0f, acc, 0, 0, 0);
    carry = acc[0] * 1e-30f + 1.0f;
    dead = __syncthreads_or(dead);
  }
  long long t1 = wall_clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc[1] == 123.f) sink[0] = acc[1];
}

int main() {
  const int G = 32, steps = 200;
  float* panel;
  unsigned* flags;
  long long* cyc;
  int* err;
  float* sink;
  (void)hipMalloc(&panel, sizeof(float) * 2 * G * 3072);
  (void)hipMalloc(&flags, sizeof(unsigned) * G * 16);
  (void)hipMalloc(&cyc, sizeof(long long) * 1024);
  (void)hipMalloc(&err, sizeof(int));
  (void)hipMalloc(&sink, 4);
  const double tick_us = 0.01;  // wall clock 100 MHz
  for (int apart : {8, 1}) {
    (void)hipMemset(flags, 0, sizeof(unsigned) * G * 16);
    (void)hipMemset(err, 0, sizeof(int));
    // two active workgroups `apart` positions from each other in dispatch order; the others exit at once
    hipLaunchKernelGGL(pingpong, dim3(2), dim3(64), 0, 0, flags, 2000, cyc, err);
    hipError_t e = hipDeviceSynchronize();
    long long h[2];
    int herr;
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost);
    printf("ping-pong (adjacent workgroups, run %d): %.2f us per one-way hop (err=%d, %s)\n", apart, h[0] * tick_us / 2000, herr,
           hipGetErrorString(e));
  }
  for (int one_wave = 0; one_wave < 2; ++one_wave)
    for (int P : {1, 2, 3, 4, 6})
      for (int mfmas : {0, 72, 144}) {
        (void)hipMemset(panel, 0, sizeof(float) * 2 * G * 3072);
        (void)hipMemset(flags, 0, sizeof(unsigned) * G * 16);
        (void)hipMemset(err, 0, sizeof(int));
        hipLaunchKernelGGL(gather, dim3(G * P), dim3(256), 0, 0, panel, flags, G, P, steps, mfmas, one_wave, cyc, err, sink);
        hipError_t e = hipDeviceSynchronize();
        std::vector<long long> h(G * P);
        int herr = 0;
        (void)hipMemcpy(h.data(), cyc, sizeof(long long) * G * P, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto x : h) mean += x;
        mean /= h.size();
        printf("gather %s P=%2d mfma/step=%3d : %.2f us/step  (err=%d, %s)\n", one_wave ? "one wave + LDS" : "every wave   ", P, mfmas,
               mean * tick_us / steps, herr, hipGetErrorString(e));
        fflush(stdout);
      }
  return 0;
}
