/*
 * kzo_alias.c -- ORACLE (test infrastructure only; see kzo.h).  AliasCodec = transforms PACK (18) and DNA (19):
 *   K/transform/AliasCodec.java:76-279 (forward), :289-470 (inverse), :472-475 (getMaxEncodedLength),
 *   :477-520 (Alias ordering: frequency descending, then pair value descending);
 *   K/Global.java:341-420 (computeHistogramOrder1: pair (previous byte, byte), previous of the first byte = 0).
 * *dataType is the block's "dataType" context entry (NULL = codec built without a context).
 */
#include "kzo.h"
#include <stdlib.h>
#include <string.h>

#define ALIAS_MIN_BLOCK 1024

int kzo_alias_max_encoded_len(int n) { return n + 1024; }

static int cmp_alias_desc(const void* a, const void* b) {               /* composite (freq << 16 | pair) descending */
  const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return (x < y) - (x > y);
}

int kzo_alias_forward(int onlyDNA, int* dataType, const uint8_t* src, int count, uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  if (count == 0) return 1;
  if (count < ALIAS_MIN_BLOCK) return 0;
  if (dstCap < kzo_alias_max_encoded_len(count)) return 0;
  int dt = KZO_DT_UNDEFINED;
  if (dataType) {                                                       /* :101-114 */
    dt = *dataType;
    if (dt == KZO_DT_MULTIMEDIA || dt == KZO_DT_UTF8) return 0;
    if (dt == KZO_DT_EXE || dt == KZO_DT_BIN) return 0;
    if (onlyDNA && dt != KZO_DT_UNDEFINED && dt != KZO_DT_DNA) return 0;
  }
  int freqs0[256], absent[256], n0 = 0;
  memset(freqs0, 0, sizeof(freqs0));
  for (int i = 0; i < count; i++) freqs0[src[i]]++;
  for (int i = 0; i < 256; i++) if (freqs0[i] == 0) absent[n0++] = i;
  if (n0 < 16) return 0;
  if (dt == KZO_DT_UNDEFINED) {                                         /* :133-141 */
    dt = kzo_detect_simple_type(count, freqs0);
    if (dataType && dt != KZO_DT_UNDEFINED) *dataType = dt;
    if (dt != KZO_DT_DNA && onlyDNA) return 0;
  }
  int srcIdx = 0, dstIdx = 0;
  if (n0 >= 240) {                                                      /* small alphabet: pack bits :143-199 */
    dst[dstIdx++] = (uint8_t)n0;
    if (n0 == 255) {
      dst[dstIdx++] = src[0];
      dst[dstIdx++] = (uint8_t)count; dst[dstIdx++] = (uint8_t)(count >> 8); dst[dstIdx++] = (uint8_t)(count >> 16); dst[dstIdx++] = (uint8_t)(count >> 24);
      srcIdx += count;
    } else {
      int map8[256];
      memset(map8, 0, sizeof(map8));
      for (int i = 0, j = 0; i < 256; i++) if (freqs0[i] != 0) { dst[dstIdx++] = (uint8_t)i; map8[i] = j++; }
      if (n0 >= 252) {                                                  /* 4 symbols or less: 2 bits each */
        dst[dstIdx++] = (uint8_t)(count & 3);
        if ((count & 3) > 2) dst[dstIdx++] = src[srcIdx++];
        if ((count & 3) > 1) dst[dstIdx++] = src[srcIdx++];
        if ((count & 3) > 0) dst[dstIdx++] = src[srcIdx++];
        while (srcIdx < count) {
          dst[dstIdx++] = (uint8_t)((map8[src[srcIdx]] << 6) | (map8[src[srcIdx + 1]] << 4) | (map8[src[srcIdx + 2]] << 2) | map8[src[srcIdx + 3]]);
          srcIdx += 4;
        }
      } else {                                                          /* 16 symbols or less: 4 bits each */
        dst[dstIdx++] = (uint8_t)(count & 1);
        if ((count & 1) != 0) dst[dstIdx++] = src[srcIdx++];
        while (srcIdx < count) { dst[dstIdx++] = (uint8_t)((map8[src[srcIdx]] << 4) | map8[src[srcIdx + 1]]); srcIdx += 2; }
      }
    }
  } else {                                                              /* digram aliases :200-268 */
    uint32_t* freqs1 = (uint32_t*)calloc(65536, sizeof(uint32_t));
    uint64_t* keys = (uint64_t*)malloc(65536 * sizeof(uint64_t));
    { int prv = 0; for (int i = 0; i < count; i++) { freqs1[(prv << 8) | src[i]]++; prv = src[i]; } }
    int n1 = 0;
    for (int i = 0; i < 65536; i++) if (freqs1[i]) keys[n1++] = ((uint64_t)freqs1[i] << 16) | (uint64_t)i;
    free(freqs1);
    if (n1 < n0) { n0 = n1; if (n0 < 16) { free(keys); return 0; } }
    qsort(keys, (size_t)n1, sizeof(uint64_t), cmp_alias_desc);          /* TreeSet order (:492-503) */
    uint16_t* map16 = (uint16_t*)malloc(65536 * sizeof(uint16_t));
    for (int i = 0; i < 65536; i++) map16[i] = (uint16_t)((i >> 8) | 0x100);
    int64_t savings = 0;
    dst[0] = (uint8_t)n0; dst[1] = 0; dstIdx = 2;
    for (int i = 0; i < n0; i++) {
      const int idx = (int)(keys[i] & 0xFFFF);
      savings += (int64_t)(keys[i] >> 16);
      map16[idx] = (uint16_t)(absent[i] | 0x200);
      dst[dstIdx] = (uint8_t)(idx >> 8); dst[dstIdx + 1] = (uint8_t)idx; dst[dstIdx + 2] = (uint8_t)absent[i];
      dstIdx += 3;
    }
    free(keys);
    if (savings < count / 20) { free(map16); return 0; }
    const int srcEnd = count - 1;
    while (srcIdx < srcEnd) {
      const int alias = map16[(src[srcIdx] << 8) | src[srcIdx + 1]];
      dst[dstIdx++] = (uint8_t)alias;
      srcIdx += alias >> 8;
    }
    if (srcIdx != srcEnd + 1) { dst[1] = 1; dst[dstIdx++] = src[srcIdx++]; }
    free(map16);
  }
  *produced = dstIdx;
  return dstIdx < count;
}

/* Reads past the block and writes past the output array are what the Java code would die of (or serve stale bytes
 * for): restated as failure. */
int kzo_alias_inverse(const uint8_t* src, int count, uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  if (count == 0) return 1;
  int srcIdx = 0, dstIdx = 0;
  int n = src[srcIdx++];
  if (n < 16) return 0;
  if (n >= 240) {
    n = 256 - n;
    if (n == 1) {
      if (count < 6) return 0;
      const uint8_t val = src[srcIdx++];
      const int32_t oSize = (int32_t)((uint32_t)src[srcIdx] | ((uint32_t)src[srcIdx + 1] << 8) | ((uint32_t)src[srcIdx + 2] << 16) | ((uint32_t)src[srcIdx + 3] << 24));
      if (oSize < 0 || oSize > dstCap) return 0;
      memset(dst, val, (size_t)oSize);
      dstIdx = oSize;
    } else {
      uint8_t idx2symb[16];
      memset(idx2symb, 0, sizeof(idx2symb));
      if (1 + n + 1 > count) return 0;
      for (int i = 0; i < n; i++) idx2symb[i] = src[srcIdx++];
      const int adjust = src[srcIdx++];
      if (adjust >= 4) return 0;
      if (n <= 4) {
        if (srcIdx + adjust > count) return 0;
        if ((int64_t)adjust + 4LL * (count - srcIdx - adjust) > dstCap) return 0;
        for (int k = 0; k < adjust; k++) dst[dstIdx++] = src[srcIdx++];
        while (srcIdx < count) {
          const int v = src[srcIdx++];
          dst[dstIdx] = idx2symb[(v >> 6) & 3]; dst[dstIdx + 1] = idx2symb[(v >> 4) & 3];
          dst[dstIdx + 2] = idx2symb[(v >> 2) & 3]; dst[dstIdx + 3] = idx2symb[v & 3];
          dstIdx += 4;
        }
      } else {
        const int raw = adjust != 0 ? 1 : 0;                            /* :381-382: one byte whatever the value */
        if (srcIdx + raw > count) return 0;
        if ((int64_t)raw + 2LL * (count - srcIdx - raw) > dstCap) return 0;
        if (raw) dst[dstIdx++] = src[srcIdx++];
        while (srcIdx < count) {
          const int v = src[srcIdx++];
          dst[dstIdx] = idx2symb[v >> 4]; dst[dstIdx + 1] = idx2symb[v & 15];
          dstIdx += 2;
        }
      }
    }
  } else {
    if (2 + 3 * n > count) return 0;
    const int adjust = src[srcIdx++];
    const int srcEnd = count - adjust;
    uint32_t map16[256];
    for (int i = 0; i < 256; i++) map16[i] = 0x10000u | (uint32_t)i;
    for (int i = 0; i < n; i++) { map16[src[srcIdx + 2]] = 0x20000u | src[srcIdx] | ((uint32_t)src[srcIdx + 1] << 8); srcIdx += 3; }
    while (srcIdx < srcEnd) {                                           /* :405-431, one loop: same bytes, same verdict */
      const uint32_t val = map16[src[srcIdx++]];
      const int inc = (int)(val >> 16);
      if (dstIdx + inc > dstCap) return 0;
      dst[dstIdx] = (uint8_t)val;
      if (inc == 2) dst[dstIdx + 1] = (uint8_t)(val >> 8);
      dstIdx += inc;
    }
    if (adjust != 0) {
      if (dstIdx >= dstCap) return 0;
      if (srcIdx >= count || srcIdx < 0) return 0;
      dst[dstIdx++] = src[srcIdx++];
    }
  }
  *produced = dstIdx;
  return 1;
}
