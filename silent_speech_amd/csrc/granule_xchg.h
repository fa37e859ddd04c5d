// Cross-workgroup exchange primitives of the persistent recurrence kernels (gru_split.h: exact-f32, H <= 192; gru_bf16_pers.h:
// bf16 MFMA, H <= 512): 8-byte {payload, tag} granules, bounded sweeps, XCD discovery, the launch generation, fault injection.
// See gru_split.h for the protocol.
#pragma once

namespace {

constexpr int SYNC_HDR_WORDS = 64;

#define SS_AGENT __HIP_MEMORY_SCOPE_AGENT
#define NAN_F __builtin_nanf("")
typedef unsigned long long u64;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int AUX_SC1 = 16;  // cache-policy operand of the raw buffer intrinsics: bit 4 = sc1 (agent scope)

__device__ __forceinline__ rsrc_t granule_rsrc(u64* base, long granules) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(granules * 8), 0x00020000);
}

// Two adjacent granules in ONE 16-byte write-through store (each 8-byte half lands whole; a 16-byte sc1 store costs
// the fabric what an 8-byte one does).  `pair` = index of the granule pair.
__device__ __forceinline__ void store_granule_pair(rsrc_t rs, int pair, unsigned tag, float v0, float v1, bool same_xcd) {
  const u32x4 d = {__float_as_uint(v0), tag, __float_as_uint(v1), tag};
  if (same_xcd) __builtin_amdgcn_raw_buffer_store_b128(d, rs, pair * 16, 0, 0);  // stays in the shared L2 (wave-uniform branch)
  else __builtin_amdgcn_raw_buffer_store_b128(d, rs, pair * 16, 0, AUX_SC1);
}

// Which XCD this workgroup runs on, and whether all P partners of its pair share it.  xid: [pairs][P] granules in the
// sync header area, tag = generation base + 1023 (step tags are base + 1 .. base + 1022; never 0, the cleared state).
__device__ __forceinline__ bool partners_share_xcd(u64* xid, int pair, int part, int P, unsigned base, unsigned* errors, int lane,
                                                   bool* lost) {
  base += 1023u;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 15u;
  u64* mine = xid + (long)pair * P;
  if (threadIdx.x == 0) __hip_atomic_store(mine + part, ((u64)base << 32) | xcc, __ATOMIC_RELAXED, SS_AGENT);
  bool same = true;
  for (int spins = 0;; ++spins) {
    const u64 x = lane < P ? __hip_atomic_load(mine + lane, __ATOMIC_RELAXED, SS_AGENT) : (((u64)base << 32) | xcc);
    const bool ok = (unsigned)(x >> 32) == base;
    if (__all(ok)) {
      same = __all((unsigned)x == xcc);
      break;
    }
    if (spins > (1 << 20)) {
      if (lane == 0) atomicAdd(errors, 1u);
      same = false;
      *lost = true;
      break;
    }
  }
  return same;
}

// One wave re-reads its N granule pairs (pair stride 256: the whole workgroup sweeps a contiguous run) until every
// tag matches.
template <int N>
__device__ __forceinline__ bool sweep_granules(rsrc_t rs, int pair0, unsigned tag, float (&v)[2 * N], unsigned* errors, int lane) {
  for (int spins = 0;;) {
    bool ok = true;
    asm volatile("" ::: "memory");  // every pass really loads again
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, (pair0 + 256 * k) * 16, 0, AUX_SC1);
      v[2 * k] = __uint_as_float(x[0]);
      v[2 * k + 1] = __uint_as_float(x[2]);
      ok &= x[1] == tag && x[3] == tag;
    }
    if (__all(ok)) return true;
    if (++spins > (1 << 20)) {  // ~1 s of sweeping: a partner is gone
      if (lane == 0) atomicAdd(errors, 1u);
      return false;
    }
  }
}

// sync[2] counts bounded waits that gave up (never reset by the kernels: the host reads it where it synchronises anyway, see
// engine.check_gru_sync).  In band, a workgroup that lost a partner emits NaN for everything IT owns from that step on (and
// publishes NaN, so its partners do the same): a poison that needs no ordering against anybody else's stores.  (Rounds 1-2
// wrote the NaN over element 0 of the result, which another workgroup owns: two XCDs then hold dirty copies of one line and the
// order of their write-backs decides -- the test of this channel found the poison lost.)
__device__ __forceinline__ void finish_launch(unsigned* sync, unsigned gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = atomicAdd(&sync[1], 1u);
    if (done == gridDim.x - 1) {
      __hip_atomic_store(&sync[1], 0u, __ATOMIC_RELAXED, SS_AGENT);
      __hip_atomic_store(&sync[0], gen + 1u, __ATOMIC_RELEASE, SS_AGENT);
    }
  }
}

// Fault injection for the tests of the failure channel (tests/test_gpu_kernels.py::test_gru_lost_partner_reaches_the_host):
// sync[5] = 1 + the index of a workgroup that plays dead -- it publishes nothing and only arrives at the end, so its partners'
// bounded waits run out.  Zero (the cleared state) = off; the owner of the workspace sets it, the kernels never do.
__device__ __forceinline__ bool plays_dead(const unsigned* sync) {
  return __hip_atomic_load(&sync[5], __ATOMIC_RELAXED, SS_AGENT) == blockIdx.x + 1u;
}

}  // namespace
