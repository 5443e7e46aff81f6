/* ORACLE (test infrastructure only; parity unpinned -- see kzo.h).
 * LZ / LZX (LZXCodec) restated from K/transform/LZCodec.java:
 *   :299-597 forward, :904-911 hash, :271-287 findMatch, :211-231 emitLength, :233-252 readLength,
 *   :626-756 inverseV6, :961-964 getMaxEncodedLength.
 * dataType: 0 = UNDEFINED; 1 = DNA (minMatch 6); 2 = SMALL_ALPHABET (declines) -- Global.DataType as
 * seen through ctx["dataType"] (:342-353).
 */
#include "kzo.h"
#include <stdlib.h>
#include <string.h>

#define HASH_SEED 0x1E35A7BDLL
#define MAX_DISTANCE1 ((1 << 16) - 2)
#define MAX_DISTANCE2 ((1 << 24) - 2)
#define MAX_MATCH (65535 + 254 + 4)
#define MIN_BLOCK_LENGTH 24

static inline uint64_t le64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline int lz_hash(const uint8_t* p, int extra) {
  return (int)(((le64(p) << 24) * (uint64_t)HASH_SEED) >> (extra ? (64 - 19) : (64 - 16)));
}
static inline int different_ints(const uint8_t* a, int i, int j) { return memcmp(a + i, a + j, 4) != 0; }
static int find_match(const uint8_t* src, int srcIdx, int ref, int maxMatch) {
  int bestLen = 0;
  while (bestLen + 8 <= maxMatch) {
    const uint64_t diff = le64(src + srcIdx + bestLen) ^ le64(src + ref + bestLen);
    if (diff != 0) { bestLen += (__builtin_ctzll(diff) >> 3); break; }
    bestLen += 8;
  }
  return bestLen;
}
static int emit_length(uint8_t* block, int idx, int length) {
  if (length < 254) { block[idx] = (uint8_t)length; return idx + 1; }
  if (length < 65536 + 254) { length -= 254; block[idx] = 254; block[idx + 1] = (uint8_t)(length >> 8); block[idx + 2] = (uint8_t)length; return idx + 3; }
  length -= 255;
  block[idx] = 255; block[idx + 1] = (uint8_t)(length >> 16); block[idx + 2] = (uint8_t)(length >> 8); block[idx + 3] = (uint8_t)length;
  return idx + 4;
}

int kzo_lz_forward(int extra, int dataType, const uint8_t* src, int count, uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  if (count == 0) return 1;
  if (dstCap < ((count <= 1024) ? count + 16 : count + (count / 64)) + 2) return 0;
  if (count < MIN_BLOCK_LENGTH) return 0;
  int mm = 4;
  if (dataType == 1) mm = 6; else if (dataType == 2) return 0;
  const int hsize = extra ? (1 << 19) : (1 << 16);
  int32_t* hashes = (int32_t*)calloc((size_t)hsize, sizeof(int32_t));
  /* the reference grows mBuf / mLenBuf on demand but never tkBuf, which holds max(count / 5, 256) tokens (LZCodec.java:324-333): one
     token more is an ArrayIndexOutOfBoundsException that nothing catches before EncodingTask.call, i.e. the block and with it
     the whole write fails with ERR_PROCESS_BLOCK (CompressedOutputStream.java:1041-1044).  Returned as -1. */
  const size_t bufSize = (size_t)count + 1024;
  uint8_t* mBuf = (uint8_t*)malloc(bufSize);
  uint8_t* mLenBuf = (uint8_t*)malloc(bufSize);
  uint8_t* tkBuf = (uint8_t*)malloc(bufSize);
  const int srcEnd = count - 16 - 2;
  const int maxDist = (srcEnd < 4 * MAX_DISTANCE1) ? MAX_DISTANCE1 : MAX_DISTANCE2;
  dst[12] = (maxDist == MAX_DISTANCE1) ? 0 : 1;
  dst[12] |= (uint8_t)(((mm - 2) & 0x07) << 1);
  const int minMatch = mm;
  int srcIdx = 0, anchor = 0, dstIdx = 13, mIdx = 0, mLenIdx = 0, tkIdx = 0;
  const int tkCap = (count / 5 > 256) ? count / 5 : 256;
  int tkOver = 0;
  int repd[2] = { count, count };
  int repIdx = 0, srcInc = 0, ok = 1;

  while (srcIdx < srcEnd) {
    int bestLen = 0;
    const int h0 = lz_hash(src + srcIdx, extra);
    const int ref0 = hashes[h0];
    hashes[h0] = srcIdx;
    const int srcIdx1 = srcIdx + 1;
    int ref = srcIdx1 - repd[repIdx];
    const int minRef = (srcIdx - maxDist) > 0 ? (srcIdx - maxDist) : 0;
    if ((ref > minRef) && !different_ints(src, ref, srcIdx1)) {
      const int mx = (srcEnd - srcIdx1) < MAX_MATCH ? (srcEnd - srcIdx1) : MAX_MATCH;
      bestLen = find_match(src, srcIdx1, ref, mx);
    } else {
      ref = srcIdx1 - repd[repIdx ^ 1];
      if ((ref > minRef) && !different_ints(src, ref, srcIdx1)) {
        const int mx = (srcEnd - srcIdx1) < MAX_MATCH ? (srcEnd - srcIdx1) : MAX_MATCH;
        bestLen = find_match(src, srcIdx1, ref, mx);
      }
    }
    if (bestLen < minMatch) {
      ref = ref0;
      if ((ref > minRef) && !different_ints(src, ref, srcIdx)) {
        const int mx = (srcEnd - srcIdx) < MAX_MATCH ? (srcEnd - srcIdx) : MAX_MATCH;
        bestLen = find_match(src, srcIdx, ref, mx);
      }
      if (bestLen < minMatch) { srcIdx = srcIdx1 + (srcInc >> 6); srcInc++; repIdx = 0; continue; }
      if ((ref != srcIdx - repd[0]) && (ref != srcIdx - repd[1])) {
        const int h1 = lz_hash(src + srcIdx1, extra);
        const int ref1 = hashes[h1];
        hashes[h1] = srcIdx1;
        if ((ref1 > minRef + 1) && !different_ints(src, ref1 + bestLen - 3, srcIdx1 + bestLen - 3)) {
          const int mx = (srcEnd - srcIdx1) < MAX_MATCH ? (srcEnd - srcIdx1) : MAX_MATCH;
          const int bestLen1 = find_match(src, srcIdx1, ref1, mx);
          if (bestLen1 >= bestLen) { ref = ref1; bestLen = bestLen1; srcIdx = srcIdx1; }
        }
        if (extra) {
          const int srcIdx2 = srcIdx1 + 1;
          const int h2 = lz_hash(src + srcIdx2, extra);
          const int ref2 = hashes[h2];
          hashes[h2] = srcIdx2;
          if ((ref2 > minRef + 2) && !different_ints(src, ref2 + bestLen - 3, srcIdx2 + bestLen - 3)) {
            const int mx = (srcEnd - srcIdx2) < MAX_MATCH ? (srcEnd - srcIdx2) : MAX_MATCH;
            const int bestLen2 = find_match(src, srcIdx2, ref2, mx);
            if (bestLen2 >= bestLen) { ref = ref2; bestLen = bestLen2; srcIdx = srcIdx2; }
          }
        }
      }
      while ((srcIdx > anchor) && (ref > minRef) && (src[srcIdx - 1] == src[ref - 1])) { bestLen++; ref--; srcIdx--; }
      if (bestLen > MAX_MATCH) { ref += (bestLen - MAX_MATCH); srcIdx += (bestLen - MAX_MATCH); bestLen = MAX_MATCH; }
    } else {
      if ((bestLen >= MAX_MATCH) || (src[srcIdx] != src[ref - 1])) {
        srcIdx++;
        hashes[lz_hash(src + srcIdx, extra)] = srcIdx;
      } else { bestLen++; ref--; }
    }
    srcInc = 0;
    const int dist = srcIdx - ref;
    int token, mLenTh;
    if (dist == repd[0]) { token = 0x00; mLenTh = 3; }
    else if (dist == repd[1]) { token = 0x04; mLenTh = 3; }
    else {
      mBuf[mIdx] = (uint8_t)(dist >> 16);
      const int inc1 = dist >= 65536 ? 1 : 0; mIdx += inc1;
      mBuf[mIdx] = (uint8_t)(dist >> 8);
      const int inc2 = dist >= 256 ? 1 : 0; mIdx += inc2;
      mBuf[mIdx++] = (uint8_t)dist;
      token = (inc1 + inc2 + 1) << 3;
      mLenTh = 7;
    }
    const int mLen = bestLen - minMatch;
    if (mLen >= mLenTh) { token += mLenTh; mLenIdx = emit_length(mLenBuf, mLenIdx, mLen - mLenTh); }
    else token += mLen;
    repd[1] = repd[0]; repd[0] = dist; repIdx = 1;
    const int litLen = srcIdx - anchor;
    if (tkIdx >= tkCap) { tkOver = 1; break; }
    if (litLen == 0) tkBuf[tkIdx++] = (uint8_t)token;
    else {
      if (litLen >= 7) {
        if (litLen >= (1 << 24)) { ok = 0; break; }
        tkBuf[tkIdx++] = (uint8_t)((7 << 5) | token);
        dstIdx = emit_length(dst, dstIdx, litLen - 7);
      } else tkBuf[tkIdx++] = (uint8_t)((litLen << 5) | token);
      memcpy(dst + dstIdx, src + anchor, (size_t)litLen);
      dstIdx += litLen;
    }
    anchor = srcIdx + bestLen;
    while (srcIdx + 4 < anchor) {
      srcIdx += 4;
      hashes[lz_hash(src + srcIdx - 3, extra)] = srcIdx - 3;
      hashes[lz_hash(src + srcIdx - 2, extra)] = srcIdx - 2;
      hashes[lz_hash(src + srcIdx - 1, extra)] = srcIdx - 1;
      hashes[lz_hash(src + srcIdx, extra)] = srcIdx;
    }
    while (++srcIdx < anchor) hashes[lz_hash(src + srcIdx, extra)] = srcIdx;
  }
  int res = 0;
  if (ok) {
    const int litLen = count - anchor;
    if (dstIdx + litLen + tkIdx + mIdx + mLenIdx >= count) ok = 0;           /* :571-572 */
    else if (tkIdx >= tkCap) tkOver = 1;
    else {
      if (litLen >= 7) { tkBuf[tkIdx++] = (uint8_t)(7 << 5); dstIdx = emit_length(dst, dstIdx, litLen - 7); }
      else tkBuf[tkIdx++] = (uint8_t)(litLen << 5);
      memcpy(dst + dstIdx, src + anchor, (size_t)litLen);
      dstIdx += litLen;
      const uint32_t a = (uint32_t)dstIdx, b = (uint32_t)tkIdx, c = (uint32_t)mIdx;
      memcpy(dst, &a, 4); memcpy(dst + 4, &b, 4); memcpy(dst + 8, &c, 4);      /* little endian (:585-587) */
      memcpy(dst + dstIdx, tkBuf, (size_t)tkIdx); dstIdx += tkIdx;
      memcpy(dst + dstIdx, mBuf, (size_t)mIdx); dstIdx += mIdx;
      memcpy(dst + dstIdx, mLenBuf, (size_t)mLenIdx); dstIdx += mLenIdx;
      *produced = dstIdx;
      res = dstIdx <= count - (count / 100);
    }
  }
  free(hashes); free(mBuf); free(mLenBuf); free(tkBuf);
  if (tkOver) { *produced = 0; return -1; }
  return ok ? res : 0;
}

/* readLength (LZCodec.java).  Every read of the inverse is bounded by the block length `count`: past it the Java code
 * either throws (end of the array) or picks up stale bytes of a reused buffer; both are restated as failure (*bad). */
static int read_length(const uint8_t* a, int* idx, int count, int* bad) {
  if (*idx + 4 > count) {
    int need = 1;
    if (*idx < count) need = (a[*idx] < 254) ? 1 : (a[*idx] == 254 ? 3 : 4);
    if (*idx + need > count) { *bad = 1; return 0; }
  }
  int res = a[(*idx)++];
  if (res < 254) return res;
  if (res == 254) { res += (a[(*idx)++] << 8); res += a[(*idx)++]; return res; }
  res += (a[*idx] << 16); res += (a[*idx + 1] << 8); res += a[*idx + 2];
  *idx += 3;
  return res;
}

int kzo_lz_inverse(int extra, const uint8_t* src, int count, uint8_t* dst, int dstCap, int* produced) {
  (void)extra;
  *produced = 0;
  if (count == 0) return 1;
  if (count < 13) return 0;
  const int dstEnd = dstCap;
  int32_t tkLen, mIdxLen, mLenLen;
  memcpy(&tkLen, src, 4); memcpy(&mIdxLen, src + 4, 4); memcpy(&mLenLen, src + 8, 4);
  if ((tkLen < 0) || (mIdxLen < 0) || (mLenLen < 0)) return 0;
  if ((tkLen < 13) || (tkLen > count) || (mIdxLen > count - tkLen) || (mLenLen > count - tkLen - mIdxLen)) return 0;
  int tkIdx = tkLen, mIdx = tkIdx + mIdxLen, mLenIdx = mIdx + mLenLen;
  const int srcEnd = tkIdx - 13, litEnd = tkIdx;
  const int maxDist = ((src[12] & 1) == 0) ? MAX_DISTANCE1 : MAX_DISTANCE2;
  const int minMatch = ((src[12] >> 1) & 0x07) + 2;
  int srcIdx = 13, dstIdx = 0, repd0 = count, repd1 = count, bad = 0;
  for (;;) {
    if (tkIdx >= count) return 0;                                            /* Java: AIOOBE on corrupt data */
    const int token = src[tkIdx++];
    if (token >= 32) {
      const int litLen = (token >= 0xE0) ? 7 + read_length(src, &srcIdx, count, &bad) : token >> 5;
      if (bad) return 0;
      if ((litLen > dstEnd - dstIdx) || (litLen > litEnd - srcIdx)) return 0;
      memcpy(dst + dstIdx, src + srcIdx, (size_t)litLen);
      srcIdx += litLen; dstIdx += litLen;
      if (srcIdx >= srcEnd) break;
    }
    int mLen, dist;
    const int f = token & 0x18;
    if (f == 0) {
      mLen = token & 0x03;
      mLen += (mLen == 3) ? minMatch + read_length(src, &mLenIdx, count, &bad) : minMatch;
      dist = ((token & 0x04) == 0) ? repd0 : repd1;
    } else {
      mLen = token & 0x07;
      mLen += (mLen == 7) ? minMatch + read_length(src, &mLenIdx, count, &bad) : minMatch;
      if (mIdx + ((f == 0x18) ? 3 : (f == 0x10) ? 2 : 1) > count) return 0;
      dist = src[mIdx++];
      if (f == 0x18) { dist = (dist << 8) | src[mIdx++]; dist = (dist << 8) | src[mIdx++]; }
      else if (f == 0x10) dist = (dist << 8) | src[mIdx++];
    }
    if (bad) return 0;
    repd1 = repd0; repd0 = dist;
    const int mEnd = dstIdx + mLen;
    const int ref = dstIdx - dist;
    if ((ref < 0) || (dist > maxDist) || (mEnd > dstEnd)) return 0;
    if (dist == 0) return 0;   /* the reference copies the region onto itself here, exposing stale buffer bytes: rejected */
    for (int i = 0; i < mLen; i++) dst[dstIdx + i] = dst[ref + i];
    dstIdx = mEnd;
  }
  *produced = dstIdx;
  return srcIdx == srcEnd + 13;
}
