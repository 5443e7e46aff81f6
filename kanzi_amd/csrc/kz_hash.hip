// kz_hash.hip -- per-block checksums (the -x32 / -x64 option) on gfx950.
//
// Replaces K/util/hash/XXHash32.java:60-131 and K/util/hash/XXHash64.java:58-143 as used by
// K/io/CompressedOutputStream.java:749-755 (hash of the ORIGINAL block, seed "KANZ" 0x4B414E5A) and
// K/io/CompressedInputStream.java:1349-1363 (verification).  XXHash64 keeps the reference's deviations
// from the standard XXH64 (32-bit rotate amounts when merging the 4 lanes, sign-extended 4-byte tail).
// The 4 accumulator lanes are sequential chains (rotate-multiply is not associative), so the
// parallelism is 4 lanes per block x blocks of the batch: one wave per block, rows of 256/512 bytes
// fetched coalesced and handed to lanes 0..3 with wave shuffles.
#include "kz_device.h"
#include "kz_internal.h"

typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint8_t u8;

#define KZ_SEED 0x4B414E5Au
#define P32_1 0x9E3779B1u
#define P32_2 0x85EBCA77u
#define P32_3 0xC2B2AE3Du
#define P32_4 0x27D4EB2Fu
#define P32_5 0x165667B1u
#define P64_1 0x9E3779B185EBCA87ULL
#define P64_2 0xC2B2AE3D27D4EB4FULL
#define P64_3 0x165667B19E3779F9ULL
#define P64_4 0x85EBCA77C2B2AE63ULL
#define P64_5 0x27D4EB2F165667C5ULL

__device__ __forceinline__ u32 h_rotl32(u32 x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ u64 h_rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ u64 h_round64(u64 acc, u64 val) { acc += val * P64_2; return h_rotl64(acc, 31) * P64_1; }
__device__ __forceinline__ u64 h_merge64(u64 acc, u64 val) { acc ^= h_round64(0, val); return acc * P64_1 + P64_4; }
__device__ __forceinline__ u32 h_le32(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }
__device__ __forceinline__ u64 h_le64(const u8* p) { return (u64)h_le32(p) | ((u64)h_le32(p + 4) << 32); }

// kind 1 = XXHash32 (result in low 32 bits), 2 = XXHash64
__global__ __launch_bounds__(64) void k_xxhash(const u8* __restrict__ data, int64_t stride, const int32_t* __restrict__ d_len,
                                                u64* __restrict__ d_hash, int kind) {
  const int b = blockIdx.x;
  const int length = d_len[b];
  const int lane = kz_lane();
  const u8* p = data + (int64_t)b * stride;
  if (kind == 1) {
    u32 h32;
    int idx = 0;
    if (length >= 16) {
      u32 v = (lane == 0) ? KZ_SEED + P32_1 + P32_2 : (lane == 1) ? KZ_SEED + P32_2 : (lane == 2) ? KZ_SEED : KZ_SEED - P32_1;
      const int nStripes = length >> 4;
      for (int s0 = 0; s0 < nStripes; s0 += 16) {
        const int base = s0 * 16 + lane * 4;
        const u32 w = (base + 4 <= length) ? h_le32(p + base) : 0u;       // 64 lanes x 4 B = 16 stripes
        const int cnt = min(16, nStripes - s0);
        for (int t = 0; t < cnt; t++) {
          const u32 val = __shfl(w, (4 * t + lane) & 63, 64);
          if (lane < 4) v = h_rotl32(v + val * P32_2, 13) * P32_1;
        }
      }
      idx = nStripes * 16;
      const u32 v1 = __shfl(v, 0, 64), v2 = __shfl(v, 1, 64), v3 = __shfl(v, 2, 64), v4 = __shfl(v, 3, 64);
      h32 = h_rotl32(v1, 1) + h_rotl32(v2, 7) + h_rotl32(v3, 12) + h_rotl32(v4, 18);
    } else h32 = KZ_SEED + P32_5;
    h32 += (u32)length;
    while (idx + 4 <= length) { h32 += h_le32(p + idx) * P32_3; h32 = h_rotl32(h32, 17) * P32_4; idx += 4; }
    while (idx < length) { h32 += (u32)p[idx] * P32_5; h32 = h_rotl32(h32, 11) * P32_1; idx++; }
    h32 ^= h32 >> 15; h32 *= P32_2; h32 ^= h32 >> 13; h32 *= P32_3; h32 ^= h32 >> 16;
    if (lane == 0) d_hash[b] = (u64)h32;
  } else {
    const u64 seed = (u64)KZ_SEED;
    u64 h64;
    int idx = 0;
    if (length >= 32) {
      u64 v = (lane == 0) ? seed + P64_1 + P64_2 : (lane == 1) ? seed + P64_2 : (lane == 2) ? seed : seed - P64_1;
      const int nStripes = length >> 5;
      for (int s0 = 0; s0 < nStripes; s0 += 16) {
        const int base = s0 * 32 + lane * 8;
        const u64 w = (base + 8 <= length) ? h_le64(p + base) : 0ULL;     // 64 lanes x 8 B = 16 stripes
        const int cnt = min(16, nStripes - s0);
        for (int t = 0; t < cnt; t++) {
          const int srcLane = (4 * t + lane) & 63;
          const u32 lo = __shfl((u32)w, srcLane, 64), hi = __shfl((u32)(w >> 32), srcLane, 64);
          if (lane < 4) v = h_round64(v, ((u64)hi << 32) | lo);
        }
      }
      idx = nStripes * 32;
      u64 vv[4];
      for (int k = 0; k < 4; k++) vv[k] = ((u64)(u32)__shfl((u32)(v >> 32), k, 64) << 32) | (u64)(u32)__shfl((u32)v, k, 64);
      // reference quirk (XXHash64.java:88-89): 32-bit rotate amounts on 64-bit values
      h64 = ((vv[0] << 1) | (vv[0] >> 31)) + ((vv[1] << 7) | (vv[1] >> 25)) + ((vv[2] << 12) | (vv[2] >> 20)) + ((vv[3] << 18) | (vv[3] >> 14));
      h64 = h_merge64(h64, vv[0]); h64 = h_merge64(h64, vv[1]); h64 = h_merge64(h64, vv[2]); h64 = h_merge64(h64, vv[3]);
    } else h64 = seed + P64_5;
    h64 += (u64)(long long)length;
    while (idx + 8 <= length) { h64 ^= h_round64(0, h_le64(p + idx)); h64 = h_rotl64(h64, 27) * P64_1 + P64_4; idx += 8; }
    while (idx + 4 <= length) { h64 ^= (u64)(long long)(int)h_le32(p + idx) * P64_1; h64 = h_rotl64(h64, 23) * P64_2 + P64_3; idx += 4; }   // sign-extended (:111)
    while (idx < length) { h64 ^= (u64)p[idx] * P64_5; h64 = h_rotl64(h64, 11) * P64_1; idx++; }
    h64 ^= h64 >> 33; h64 *= P64_2; h64 ^= h64 >> 29; h64 *= P64_3; h64 ^= h64 >> 32;
    if (lane == 0) d_hash[b] = h64;
  }
}

int kz_block_hashes(kz_ctx* ctx, const uint8_t* data, int64_t stride, const int32_t* d_len, int B, int kind, unsigned long long* d_hash) {
  KZ_LAUNCH(ctx, KID_XXHASH, k_xxhash, dim3(B), dim3(64), data, stride, d_len, d_hash, kind);
  KZ_HIP(hipGetLastError());
  return 0;
}
