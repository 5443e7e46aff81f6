/*
 * kzo_fsd.c -- ORACLE (test infrastructure only; see kzo.h).  Fixed-step delta codec, transform id MM (15):
 *   K/transform/FSDCodec.java:60-244 (forward), :246-318 (inverse), :320-323 (getMaxEncodedLength).
 * *dataType is the block's "dataType" context entry (read and written like the reference does); NULL = no context.
 */
#include "kzo.h"
#include <string.h>

#define FSD_MIN_LENGTH 1024
#define FSD_ESCAPE 0xFF
#define FSD_DELTA 0
#define FSD_XOR 1

int kzo_fsd_max_encoded_len(int n) { return n + ((n >> 4) > 64 ? (n >> 4) : 64); }

int kzo_fsd_forward(int* dataType, const uint8_t* src, int count, uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  if (count == 0) return 1;
  if (dstCap < kzo_fsd_max_encoded_len(count)) return 0;
  if (count < FSD_MIN_LENGTH) return 0;
  if (dataType) {
    const int dt = *dataType;
    if (dt != KZO_DT_UNDEFINED && dt != KZO_DT_MULTIMEDIA && dt != KZO_DT_BIN) return 0;
  }
  {                                                             /* :93-108 only a few container types are examined */
    const int32_t magic = kzo_magic_type(src);
    if (!(magic == (int32_t)KZO_MAGIC_BMP || magic == (int32_t)KZO_MAGIC_RIFF || magic == KZO_MAGIC_PBM ||
          magic == KZO_MAGIC_PGM || magic == KZO_MAGIC_PPM || magic == 0)) return 0;
  }
  static const int DIST[7] = { 0, 1, 2, 3, 4, 8, 16 };
  const int count10 = count / 10, count5 = 2 * count10;
  int histo[7][256];
  memset(histo, 0, sizeof(histo));
  const int start[3] = { 0 * count5, 2 * count5, 4 * count5 };
  for (int i = count10; i < count5; i++)                        /* :118-148 three sampled windows */
    for (int w = 0; w < 3; w++) {
      const uint8_t* p = src + start[w] + i;
      const uint8_t b = p[0];
      histo[0][b]++;
      for (int k = 1; k < 7; k++) histo[k][(uint8_t)(b ^ p[-DIST[k]])]++;
    }
  int ent[7], minIdx = 0;
  for (int i = 0; i < 7; i++) { ent[i] = kzo_entropy1024(3 * count10, histo[i]); if (ent[i] < ent[minIdx]) minIdx = i; }
  if (ent[minIdx] >= ent[0]) {                                  /* :160-165 */
    if (dataType) *dataType = kzo_detect_simple_type(3 * count10, histo[0]);
    return 0;
  }
  if (dataType) *dataType = KZO_DT_MULTIMEDIA;
  const int dist = DIST[minIdx];
  int largeDeltas = 0;
  for (int i = 2 * count5; i < 3 * count5; i++) {               /* :173-179 */
    const int delta = (int)src[i] - (int)src[i - dist];
    if (delta < -127 || delta > 127) largeDeltas++;
  }
  const uint8_t mode = (largeDeltas > (count5 >> 5)) ? FSD_XOR : FSD_DELTA;
  int srcIdx = 0, dstIdx = 0;
  const int srcEnd = count, dstEnd = kzo_fsd_max_encoded_len(count);
  dst[0] = mode; dst[1] = (uint8_t)dist; dstIdx = 2;
  for (int i = 0; i < dist; i++) dst[dstIdx++] = src[srcIdx++];
  if (mode == FSD_DELTA) {
    while (srcIdx < srcEnd && dstIdx < dstEnd - 1) {
      const int delta = (int)src[srcIdx] - (int)src[srcIdx - dist];
      if (delta < -127 || delta > 127) {
        dst[dstIdx++] = FSD_ESCAPE;
        dst[dstIdx++] = (uint8_t)(src[srcIdx] ^ src[srcIdx - dist]);
        srcIdx++;
        continue;
      }
      dst[dstIdx++] = (uint8_t)((delta >> 31) ^ (delta << 1));  /* zigzag */
      srcIdx++;
    }
  } else {
    while (srcIdx < srcEnd) { dst[dstIdx++] = (uint8_t)(src[srcIdx] ^ src[srcIdx - dist]); srcIdx++; }
  }
  if (srcIdx != srcEnd) return 0;
  int h[256];                                                   /* :222-233 does the coded form look better? */
  memset(h, 0, sizeof(h));
  for (int i = 0; i < count10; i++) { h[dst[1 * count5 + i]]++; h[dst[3 * count5 + i]]++; }
  if (kzo_entropy1024(count5, h) >= ent[0]) return 0;
  *produced = dstIdx;
  return 1;                                                     /* allowed to expand */
}

int kzo_fsd_inverse(const uint8_t* src, int count, uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  if (count == 0) return 1;
  /* The reference reads the two header bytes (and the first `dist` bytes) without looking at the length; past the
     block it would see stale buffer bytes or throw: restated as failure. */
  if (count < 2) return 0;
  int srcIdx = 0, dstIdx = 0;
  const int srcEnd = count, dstEnd = dstCap;
  const uint8_t mode = src[0];
  const int dist = src[1];
  srcIdx = 2;
  if (dist < 1 || (dist > 4 && dist != 8 && dist != 16)) return 0;
  if (srcIdx + dist > srcEnd || dist > dstEnd) return 0;
  for (int i = 0; i < dist; i++) dst[dstIdx++] = src[srcIdx++];
  if (mode == FSD_DELTA) {
    while (srcIdx < srcEnd && dstIdx < dstEnd) {
      if (src[srcIdx] == FSD_ESCAPE) {
        srcIdx++;
        if (srcIdx == srcEnd) break;
        dst[dstIdx] = (uint8_t)(src[srcIdx] ^ dst[dstIdx - dist]);
        srcIdx++; dstIdx++;
        continue;
      }
      const int delta = (src[srcIdx] >> 1) ^ -(src[srcIdx] & 1);
      dst[dstIdx] = (uint8_t)(dst[dstIdx - dist] + delta);
      srcIdx++; dstIdx++;
    }
  } else if (mode == FSD_XOR) {
    while (srcIdx < srcEnd) {
      if (dstIdx >= dstEnd) return 0;                           /* Java: write past the array end throws */
      dst[dstIdx] = (uint8_t)(src[srcIdx] ^ dst[dstIdx - dist]);
      srcIdx++; dstIdx++;
    }
  } else return 0;
  *produced = dstIdx;
  return srcIdx == srcEnd;
}
