// kz_device.h -- wave64 / workgroup primitives shared by the gfx950 kernels.
// CDNA4 only: wavefront = 64 lanes, ballots are 64-bit, LDS-staged scans.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KZ_WAVE 64
#define KZ_WG 256                 // default workgroup: 4 waves, one per SIMD of a CU

__device__ __forceinline__ int kz_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint64_t kz_ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ uint64_t kz_lanemask_lt() { return (1ULL << kz_lane()) - 1ULL; }

// inclusive wave scan (sum) of a 32-bit value
__device__ __forceinline__ uint32_t kz_wave_incl_sum(uint32_t v) {
  const int lane = kz_lane();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}
__device__ __forceinline__ uint32_t kz_wave_incl_max(uint32_t v) {
  const int lane = kz_lane();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v = v > t ? v : t;
  }
  return v;
}
__device__ __forceinline__ uint32_t kz_wave_sum(uint32_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// Workgroup exclusive sum over blockDim.x threads (multiple of 64, <= 1024).
// lds must hold >= 17 uint32. Returns exclusive prefix; *total = workgroup total.
__device__ __forceinline__ uint32_t kz_wg_excl_sum(uint32_t v, uint32_t* lds, uint32_t* total) {
  const int lane = kz_lane();
  const int wave = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  uint32_t inc = kz_wave_incl_sum(v);
  __syncthreads();
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int w = 0; w < nw; w++) { uint32_t t = lds[w]; lds[w] = run; run += t; }
    lds[16] = run;
  }
  __syncthreads();
  uint32_t base = lds[wave];
  *total = lds[16];
  return base + inc - v;
}

// Workgroup inclusive max-scan (values are "index+1 or 0" style monotone markers).
__device__ __forceinline__ uint32_t kz_wg_incl_max(uint32_t v, uint32_t* lds, uint32_t* total) {
  const int lane = kz_lane();
  const int wave = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  uint32_t inc = kz_wave_incl_max(v);
  __syncthreads();
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int w = 0; w < nw; w++) { uint32_t t = lds[w]; lds[w] = run; run = run > t ? run : t; }
    lds[16] = run;
  }
  __syncthreads();
  uint32_t base = lds[wave];
  *total = lds[16];
  return inc > base ? inc : base;
}

// match-any on an 8-bit digit within a wave: mask of lanes (among `valid` lanes) holding the same digit
// (per bit: x = the lane's bit spread over a word, bal = the lanes whose bit is set, the lanes that agree are ~(bal ^ x); in 32-bit
// halves the compiler spends six VALU instructions per bit, with `m &= bit ? bal : ~bal` on 64-bit masks nine: see bw_match in
// kz_bwt_fwd.hip, round 6)
__device__ __forceinline__ uint64_t kz_match8(uint32_t d, bool valid) {
  const uint64_t m0 = kz_ballot(valid);
  uint32_t mlo = (uint32_t)m0, mhi = (uint32_t)(m0 >> 32);
#pragma unroll
  for (int b = 0; b < 8; b++) {
    const uint32_t x = (uint32_t)(((int32_t)(d << (31 - b))) >> 31);
    const uint64_t bal = kz_ballot(x != 0u);
    mlo &= ~((uint32_t)bal ^ x);
    mhi &= ~((uint32_t)(bal >> 32) ^ x);
  }
  return valid ? (((uint64_t)mhi << 32) | mlo) : 0ULL;
}

__device__ __forceinline__ int kz_ilog2(uint32_t x) { return 31 - __clz(x); }
