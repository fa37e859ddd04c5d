// Batch assembly on the device: clips stay resident in HBM as one ragged frame store, a training batch is a
// gather of frame rows into the padded (B, max_t, ...) tensors the model takes.
//
// Replaces what NPZWordDataset.__getitem__ and collate_fn do per clip on the host
// (/root/reference/train_model_official.py:122-204): additive feature noise (:143-145), interior frame drop
// (:146-152, expressed as a frame map), zero padding / trimming to max_t (:93-118), stacking (:174-204).  Which
// frames go where is decided on the host (a (B, max_t) int32 map, -1 = padding); the bytes never leave the GPU.
// Both kernels are pure HBM streams: 16 bytes per lane, one row per wave group.
#include "ss_common.h"

namespace {

// dst[r][:] = (map[r] >= 0 ? src[map[r]][:] : 0) + noise term (only on rows whose noise_map[r] >= 0):
//   noise != NULL : noise[noise_map[r]][:]           (host-drawn noise, the reference's np.random.normal)
//   noise == NULL : noise_std * N(0,1), Box-Muller on the Philox stream (seed, element index of dst)
__global__ __launch_bounds__(256) void batch_gather_f32_kernel(const float* __restrict__ src, int D,
                                                               const int32_t* __restrict__ frame_map, long rows,
                                                               const float* __restrict__ noise,
                                                               const int32_t* __restrict__ noise_map, float noise_std,
                                                               uint64_t seed, float* __restrict__ dst) {
  const long total = rows * D;
  for (long q = ((long)blockIdx.x * 256 + threadIdx.x) * 4; q < total; q += (long)gridDim.x * 256 * 4) {
    // D need not be a multiple of 4: walk the four elements of this 16-byte destination chunk
    float v[4];
    bool noisy[4];
    long r = q / D;           // one division per 16-byte chunk, then walk
    int d = (int)(q - r * D);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = 0.f;
      noisy[e] = false;
      if (q + e < total) {
        const int m = frame_map[r];
        if (m >= 0) x = src[(long)m * D + d];
        const int nm = noise_map ? noise_map[r] : -1;
        if (nm >= 0 && noise) x += noise[(long)nm * D + d];
        noisy[e] = m >= 0 && nm >= 0;
      }
      v[e] = x;
      if (++d == D) { d = 0; ++r; }
    }
    if (!noise && noise_std > 0.f && noise_map) {
      uint32_t rnd[4];
      const uint64_t ctr = (uint64_t)(q >> 2);
      philox4((uint32_t)ctr, (uint32_t)(ctr >> 32), 0x6e6f6973u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), rnd);
      // two Box-Muller pairs from four uniforms in (0, 1]
      const float u0 = ((float)rnd[0] + 1.0f) * 2.3283064e-10f, u1 = (float)rnd[1] * 2.3283064e-10f;
      const float u2 = ((float)rnd[2] + 1.0f) * 2.3283064e-10f, u3 = (float)rnd[3] * 2.3283064e-10f;
      const float ra = sqrtf(-2.0f * __logf(u0)), rb = sqrtf(-2.0f * __logf(u2));
      const float n[4] = {ra * __cosf(6.2831853f * u1), ra * __sinf(6.2831853f * u1), rb * __cosf(6.2831853f * u3),
                          rb * __sinf(6.2831853f * u3)};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (noisy[e]) v[e] += noise_std * n[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (q + e < total) dst[q + e] = v[e];
  }
}

// dst[r][0:frame_bytes] = map[r] >= 0 ? src[map[r]] : 0; frame_bytes % 16 == 0, one workgroup walks whole rows
__global__ __launch_bounds__(256) void batch_gather_u8_kernel(const uint8_t* __restrict__ src, int chunks /* 16-byte */,
                                                              const int32_t* __restrict__ frame_map, long rows,
                                                              uint8_t* __restrict__ dst) {
  for (long r = blockIdx.x; r < rows; r += gridDim.x) {
    const int m = frame_map[r];
    const uint4* s4 = reinterpret_cast<const uint4*>(src) + (long)(m < 0 ? 0 : m) * chunks;
    uint4* d4 = reinterpret_cast<uint4*>(dst) + r * chunks;
    for (int c = threadIdx.x; c < chunks; c += 256) d4[c] = m >= 0 ? s4[c] : uint4{0, 0, 0, 0};
  }
}

}  // namespace

extern "C" int ss_batch_gather_f32(const float* src, int D, const int32_t* frame_map, long rows, const float* noise,
                                   const int32_t* noise_map, float noise_std, uint64_t seed, float* dst,
                                   ss_stream_t stream) {
  SS_REQUIRE(src && frame_map && dst && D > 0 && rows > 0 && noise_std >= 0.f, SS_ERR_ARG);
  SS_REQUIRE(!noise || noise_map, SS_ERR_ARG);
  SS_REQUIRE((reinterpret_cast<uintptr_t>(dst) & 3) == 0, SS_ERR_ARG);
  const long chunks = (rows * D + 3) / 4;
  long blocks = (chunks + 255) / 256;
  blocks = blocks > 4096 ? 4096 : blocks;
  hipLaunchKernelGGL(batch_gather_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, D,
                     frame_map, rows, noise, noise_map, noise_std, seed, dst);
  return ss_launch_status();
}

extern "C" int ss_batch_gather_u8(const uint8_t* src, int frame_bytes, const int32_t* frame_map, long rows, uint8_t* dst,
                                  ss_stream_t stream) {
  SS_REQUIRE(src && frame_map && dst && frame_bytes > 0 && rows > 0, SS_ERR_ARG);
  SS_REQUIRE((frame_bytes & 15) == 0, SS_ERR_UNSUPPORTED);
  SS_REQUIRE((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0, SS_ERR_ARG);
  long blocks = rows > 8192 ? 8192 : rows;
  hipLaunchKernelGGL(batch_gather_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src,
                     frame_bytes / 16, frame_map, rows, dst);
  return ss_launch_status();
}
