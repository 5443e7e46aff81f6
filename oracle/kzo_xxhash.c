/* ORACLE (test infrastructure only; parity unpinned -- see kzo.h).
 * Block checksums restated from K/util/hash/XXHash32.java:60-131 and K/util/hash/XXHash64.java:58-143.
 * XXHash32 is the standard XXH32.  XXHash64 is NOT the standard XXH64: the reference combines the four
 * lanes with 32-bit rotate amounts applied to 64-bit values ((v1 << 1) | (v1 >>> 31), ...) and the 4-byte
 * tail multiplies a SIGN-EXTENDED int (:111) -- both quirks are kept.  Seed = 0x4B414E5A
 * (K/io/CompressedOutputStream.java:196-200). */
#include "kzo.h"
#include <string.h>

static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

#define P32_1 0x9E3779B1u
#define P32_2 0x85EBCA77u
#define P32_3 0xC2B2AE3Du
#define P32_4 0x27D4EB2Fu
#define P32_5 0x165667B1u

uint32_t kzo_xxhash32(const uint8_t* data, int length, uint32_t seed) {
  const uint8_t* p = data; const uint8_t* end = data + length;
  uint32_t h32;
  if (length >= 16) {
    const uint8_t* end16 = end - 16;
    uint32_t v1 = seed + P32_1 + P32_2, v2 = seed + P32_2, v3 = seed, v4 = seed - P32_1;
    do {
      v1 = rotl32(v1 + rd32(p) * P32_2, 13) * P32_1;
      v2 = rotl32(v2 + rd32(p + 4) * P32_2, 13) * P32_1;
      v3 = rotl32(v3 + rd32(p + 8) * P32_2, 13) * P32_1;
      v4 = rotl32(v4 + rd32(p + 12) * P32_2, 13) * P32_1;
      p += 16;
    } while (p <= end16);
    h32 = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
  } else h32 = seed + P32_5;
  h32 += (uint32_t)length;
  while (p + 4 <= end) { h32 += rd32(p) * P32_3; h32 = rotl32(h32, 17) * P32_4; p += 4; }
  while (p < end) { h32 += (uint32_t)(*p) * P32_5; h32 = rotl32(h32, 11) * P32_1; p++; }
  h32 ^= h32 >> 15; h32 *= P32_2; h32 ^= h32 >> 13; h32 *= P32_3;
  return h32 ^ (h32 >> 16);
}

#define P64_1 0x9E3779B185EBCA87ULL
#define P64_2 0xC2B2AE3D27D4EB4FULL
#define P64_3 0x165667B19E3779F9ULL
#define P64_4 0x85EBCA77C2B2AE63ULL
#define P64_5 0x27D4EB2F165667C5ULL
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t round64(uint64_t acc, uint64_t val) { acc += val * P64_2; return rotl64(acc, 31) * P64_1; }   /* :133-136 */
static inline uint64_t merge64(uint64_t acc, uint64_t val) { acc ^= round64(0, val); return acc * P64_1 + P64_4; }

uint64_t kzo_xxhash64(const uint8_t* data, int length, uint64_t seed) {
  const uint8_t* p = data; const uint8_t* end = data + length;
  uint64_t h64;
  if (length >= 32) {
    const uint8_t* end32 = end - 32;
    uint64_t v1 = seed + P64_1 + P64_2, v2 = seed + P64_2, v3 = seed, v4 = seed - P64_1;
    do {
      v1 = round64(v1, rd64(p)); v2 = round64(v2, rd64(p + 8)); v3 = round64(v3, rd64(p + 16)); v4 = round64(v4, rd64(p + 24));
      p += 32;
    } while (p <= end32);
    /* reference quirk (:88-89): 32-bit rotate amounts on 64-bit values */
    h64 = ((v1 << 1) | (v1 >> 31)) + ((v2 << 7) | (v2 >> 25)) + ((v3 << 12) | (v3 >> 20)) + ((v4 << 18) | (v4 >> 14));
    h64 = merge64(h64, v1); h64 = merge64(h64, v2); h64 = merge64(h64, v3); h64 = merge64(h64, v4);
  } else h64 = seed + P64_5;
  h64 += (uint64_t)(int64_t)length;
  while (p + 8 <= end) { h64 ^= round64(0, rd64(p)); h64 = rotl64(h64, 27) * P64_1 + P64_4; p += 8; }
  while (p + 4 <= end) { h64 ^= (uint64_t)(int64_t)(int32_t)rd32(p) * P64_1; h64 = rotl64(h64, 23) * P64_2 + P64_3; p += 4; }   /* :111 sign-extended */
  while (p < end) { h64 ^= (uint64_t)(*p) * P64_5; h64 = rotl64(h64, 11) * P64_1; p++; }
  h64 ^= h64 >> 33; h64 *= P64_2; h64 ^= h64 >> 29; h64 *= P64_3;
  return h64 ^ (h64 >> 32);
}
