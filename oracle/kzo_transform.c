/* ORACLE (test infrastructure only; parity unpinned -- see kzo.h).
 * Byte transforms restated from the reference:
 *   ZRLT  K/transform/ZRLT.java:54-136 (forward), :146-233 (inverse)
 *   SBRT  K/transform/SBRT.java:87-151 (forward), :154-214 (inverse)   modes MTF=1 RANK=2 TIMESTAMP=3
 *   BWT   K/transform/BWT.java:148-191, :245-374; output convention K/transform/DivSufSort.java:204-227,
 *         primary indexes :230-327; header K/transform/BWTBlockCodec.java:71-213
 */
#include "kzo.h"
#include <stdlib.h>
#include <string.h>

static int ilog2(uint32_t x) { return 31 - __builtin_clz(x); }   /* K/Global.java:207-212 */

/* ---------------- ZRLT ---------------- */
int kzo_zrlt_forward(const uint8_t* src, int count, uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  if (count == 0) return 1;
  if (dstCap < count) return 0;                       /* :68 getMaxEncodedLength == srcLen */
  int srcIdx = 0, dstIdx = 0;
  const int srcEnd = count, dstEnd = count;           /* :77 do not expand */
  int res = 1;
  while (srcIdx < srcEnd) {
    if (src[srcIdx] == 0) {
      int runLength = 1;
      while ((srcIdx + runLength < srcEnd) && (src[srcIdx + runLength] == 0)) runLength++;
      srcIdx += runLength;
      runLength++;
      int log2 = ilog2((uint32_t)runLength);
      if (dstIdx >= dstEnd - log2) { res = 0; break; }   /* :94 */
      while (log2 > 0) { log2--; dst[dstIdx++] = (uint8_t)((runLength >> log2) & 1); }
      continue;
    }
    int val = src[srcIdx];
    if (val >= 0xFE) {
      if (dstIdx >= dstEnd - 1) { res = 0; break; }      /* :111 */
      dst[dstIdx] = 0xFF; dst[dstIdx + 1] = (uint8_t)(val - 0xFE); dstIdx += 2;
    } else {
      if (dstIdx >= dstEnd) { res = 0; break; }          /* :120 */
      dst[dstIdx++] = (uint8_t)(val + 1);
    }
    srcIdx++;
  }
  *produced = dstIdx;
  return res && (srcIdx == srcEnd);
}

int kzo_zrlt_inverse(const uint8_t* src, int count, uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  if (count == 0) return 1;
  int srcIdx = 0, dstIdx = 0;
  const int srcEnd = count, dstEnd = dstCap;          /* :162 dstEnd = output.length */
  /* runLength is a Java int: a run of more than 30 digits wraps (ZRLT.java:172-176), and so do the sums it is tested
     with.  A write past dstEnd is the ArrayIndexOutOfBounds the Java loop would die of: failure. */
  int32_t runLength = 0;
#define WRAP_ADD(a, b) ((int32_t)((uint32_t)(a) + (uint32_t)(b)))
  for (;;) {
    int val = src[srcIdx];
    if (val <= 1) {
      runLength = 1;
      int ended = 0;
      do {
        runLength = WRAP_ADD(runLength, WRAP_ADD(runLength, val));
        srcIdx++;
        if (srcIdx >= srcEnd) { ended = 1; break; }
        val = src[srcIdx];
      } while (val <= 1);
      if (ended) break;                               /* break mainLoop */
      runLength = WRAP_ADD(runLength, -1);
      if (runLength > 0) {
        if (WRAP_ADD(dstIdx, runLength) >= dstEnd) break;
        while (runLength > 0) { runLength--; if (dstIdx >= dstEnd) return 0; dst[dstIdx++] = 0; }
      }
    }
    if (val == 0xFF) {
      srcIdx++;
      if (srcIdx >= srcEnd) break;
      dst[dstIdx] = (uint8_t)(0xFE + src[srcIdx]);
    } else {
      dst[dstIdx] = (uint8_t)(val - 1);
    }
    srcIdx++; dstIdx++;
    if ((srcIdx >= srcEnd) || (dstIdx >= dstEnd)) break;
  }
  if (runLength > 0) {                                /* :217-228 trailing zeros */
    runLength--;
    if (WRAP_ADD(dstIdx, runLength) > dstEnd) return 0;
    while (runLength > 0) { runLength--; if (dstIdx >= dstEnd) return 0; dst[dstIdx++] = 0; }
  }
#undef WRAP_ADD
  *produced = dstIdx;
  return srcIdx == srcEnd;
}

/* ---------------- SBRT ---------------- */
int kzo_sbrt_forward(int mode, const uint8_t* src, int count, uint8_t* dst) {
  int32_t p[256], q[256], s2r[256], r2s[256];
  const int32_t m1 = (mode == 3) ? 0 : -1, m2 = (mode == 1) ? 0 : -1;
  const int s = (mode == 2) ? 1 : 0;
  for (int i = 0; i < 256; i++) { p[i] = 0; q[i] = 0; s2r[i] = i; r2s[i] = i; }
  for (int i = 0; i < count; i++) {
    int c = src[i];
    int r = s2r[c];
    dst[i] = (uint8_t)r;
    int32_t qc = ((i & m1) + (p[c] & m2)) >> s;
    p[c] = i; q[c] = qc;
    while ((r > 0) && (q[r2s[r - 1]] <= qc)) { r2s[r] = r2s[r - 1]; s2r[r2s[r]] = r; r--; }
    r2s[r] = c; s2r[c] = r;
  }
  return 1;
}

int kzo_sbrt_inverse(int mode, const uint8_t* src, int count, uint8_t* dst) {
  int32_t p[256], q[256], r2s[256];
  const int32_t m1 = (mode == 3) ? 0 : -1, m2 = (mode == 1) ? 0 : -1;
  const int s = (mode == 2) ? 1 : 0;
  for (int i = 0; i < 256; i++) { p[i] = 0; q[i] = 0; r2s[i] = i; }
  for (int i = 0; i < count; i++) {
    int r = src[i];
    int c = r2s[r];
    dst[i] = (uint8_t)c;
    int32_t qc = ((i & m1) + (p[c] & m2)) >> s;
    p[c] = i; q[c] = qc;
    while ((r > 0) && (q[r2s[r - 1]] <= qc)) { r2s[r] = r2s[r - 1]; r--; }
    r2s[r] = c;
  }
  return 1;
}

/* ---------------- BWT ---------------- */
static int bwt_chunks(int n) { return n < 256 ? 1 : 8; }   /* K/transform/BWT.java:561-563 */

int kzo_bwt_forward_raw(const uint8_t* src, int n, uint8_t* dst, int32_t primary[8]) {
  for (int k = 0; k < 8; k++) primary[k] = 0;
  if (n <= 0) return 1;
  if (n == 1) { dst[0] = src[0]; return 1; }            /* K/transform/BWT.java:174-177 */
  int32_t* sa = (int32_t*)malloc((size_t)n * sizeof(int32_t));
  if (!sa) return 0;
  kzo_suffix_array(src, sa, n);
  const int chunks = bwt_chunks(n);
  const int st = n / chunks;
  const int step = (st * chunks != n) ? st + 1 : st;    /* DivSufSort.java:233-234 */
  int pIdx = -1;
  /* DivSufSort.java:217-224: out[0]=in[n-1]; rows before suffix 0 shift by one */
  dst[0] = src[n - 1];
  for (int i = 0; i < n; i++) {
    int s = sa[i];
    if ((s % step) == 0 && (s / step) < 8) primary[s / step] = i + 1;   /* :254-255,302-303,316-317,325 */
    if (s == 0) { pIdx = i; continue; }
    if (pIdx < 0) dst[1 + i] = src[s - 1]; else dst[i] = src[s - 1];
  }
  primary[0] = pIdx + 1;
  free(sa);
  return 1;
}

/* K/transform/BWT.java:245-374 (inverseMergeTPSI): packed link array data[j] = (next << 8) | byte built by a
 * counting sort, then 8 interleaved walkers (one per primary index) for memory-level parallelism, exactly
 * the reference's loop structure.  Blocks > 2^24 (the reference switches to biPSIv2 above 8 MiB, same
 * output) use an unpacked 64-bit walk. */
int kzo_bwt_inverse_raw(const uint8_t* src, int n, uint8_t* dst, const int32_t primary[8]) {
  if (n <= 0) return 1;
  if (n == 1) { dst[0] = src[0]; return 1; }
  int pIdx = primary[0];
  if ((pIdx <= 0) || (pIdx > n)) return 0;               /* :261 */
  uint32_t b[256]; memset(b, 0, sizeof(b));
  for (int i = 0; i < n; i++) b[src[i]]++;
  for (int i = 0, sum = 0; i < 256; i++) { int t = (int)b[i]; b[i] = (uint32_t)sum; sum += t; }
  int ok = 1;
  if (n < (1 << 24)) {
    uint32_t* data = (uint32_t*)malloc(((size_t)n + 64) * sizeof(uint32_t));
    if (!data) return 0;
    { int v = src[0]; data[b[v]] = 0xFF00u | (uint32_t)v; b[v]++; }                               /* :273-275 */
    for (int i = 1; i < pIdx; i++) { int v = src[i]; data[b[v]] = ((uint32_t)(i - 1) << 8) | (uint32_t)v; b[v]++; }
    for (int i = pIdx; i < n; i++) { int v = src[i]; data[b[v]] = ((uint32_t)i << 8) | (uint32_t)v; b[v]++; }
    if (bwt_chunks(n) != 8) {
      uint32_t t = (uint32_t)(pIdx - 1);
      for (int i = 0; i < n; i++) { if (t >= (uint32_t)n) { ok = 0; break; } uint32_t ptr = data[t]; dst[i] = (uint8_t)ptr; t = ptr >> 8; }
    } else {
      const int ckSize = ((n & 7) == 0) ? (n >> 3) : (n >> 3) + 1;
      uint32_t t[8];
      for (int k = 0; k < 8; k++) {
        int64_t tk = (int64_t)primary[k] - 1;
        if (tk < 0 || tk >= n) { ok = 0; break; }          /* :305-311 */
        t[k] = (uint32_t)tk;
      }
      if (ok) {
        const int end = n - ckSize * 7;                    /* :313-314 */
        int i = 0;
        for (; i < end && ok; i++) {
          for (int k = 0; k < 8; k++) {
            if (t[k] >= (uint32_t)n) { ok = 0; break; }    /* corrupt stream (Java would throw) */
            uint32_t ptr = data[t[k]]; dst[i + k * ckSize] = (uint8_t)ptr; t[k] = ptr >> 8;
          }
        }
        for (; i < ckSize && ok; i++) {
          for (int k = 0; k < 7; k++) {
            if (t[k] >= (uint32_t)n) { ok = 0; break; }
            uint32_t ptr = data[t[k]]; dst[i + k * ckSize] = (uint8_t)ptr; t[k] = ptr >> 8;
          }
        }
      }
    }
    free(data);
    return ok;
  }
  uint32_t* nxt = (uint32_t*)malloc((size_t)n * sizeof(uint32_t));
  uint8_t* fch = (uint8_t*)malloc((size_t)n);
  if (!nxt || !fch) { free(nxt); free(fch); return 0; }
  { int v = src[0]; fch[b[v]] = (uint8_t)v; nxt[b[v]] = 0xFF; b[v]++; }
  for (int i = 1; i < pIdx; i++) { int v = src[i]; fch[b[v]] = (uint8_t)v; nxt[b[v]] = (uint32_t)(i - 1); b[v]++; }
  for (int i = pIdx; i < n; i++) { int v = src[i]; fch[b[v]] = (uint8_t)v; nxt[b[v]] = (uint32_t)i; b[v]++; }
  {
    const int ckSize = ((n & 7) == 0) ? (n >> 3) : (n >> 3) + 1;
    for (int k = 0; k < 8 && ok; k++) {
      int64_t t = (int64_t)primary[k] - 1;
      if (t < 0 || t >= n) { ok = 0; break; }
      int start = k * ckSize;
      int end = start + ckSize < n ? start + ckSize : n;
      for (int i = start; i < end; i++) { if (t >= n) { ok = 0; break; } dst[i] = fch[t]; t = nxt[t]; }
    }
  }
  free(nxt); free(fch);
  return ok;
}

int kzo_bwt_forward(const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  if (n == 0) return 1;
  if (dstCap < n + 33) return 0;                          /* BWTBlockCodec.java:84-86 */
  int logBlockSize = ilog2((uint32_t)n);
  if ((n & (n - 1)) != 0) logBlockSize++;
  const int pIndexSize = (logBlockSize + 7) >> 3;
  if ((pIndexSize <= 0) || (pIndexSize >= 5)) return 0;
  const int chunks = bwt_chunks(n);
  const int logNbChunks = ilog2((uint32_t)chunks);
  const int hdr = 1 + chunks * pIndexSize;
  int32_t primary[8];
  if (!kzo_bwt_forward_raw(src, n, dst + hdr, primary)) return 0;
  dst[0] = (uint8_t)((logNbChunks << 2) | (pIndexSize - 1));
  int idx = 1;
  for (int i = 0; i < chunks; i++) {
    int32_t pi = primary[i] - 1;
    for (int shift = (pIndexSize - 1) << 3; shift >= 0; shift -= 8) dst[idx++] = (uint8_t)(pi >> shift);
  }
  *produced = hdr + n;
  return 1;
}

int kzo_bwt_inverse(const uint8_t* src, int blockSize, uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  if (blockSize == 0) return 1;
  int pos = 0;
  uint8_t mode = src[pos++];
  const int logNbChunks = (mode >> 2) & 7;
  const int pIndexSize = (mode & 3) + 1;
  const int chunks = 1 << logNbChunks;
  const int headerSize = 1 + chunks * pIndexSize;
  if (blockSize < headerSize) return 0;
  if (chunks != bwt_chunks(blockSize - headerSize)) return 0;
  int32_t primary[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < chunks; i++) {
    int64_t pi = 0;
    for (int shift = (pIndexSize - 1) << 3; shift >= 0; shift -= 8) pi = (pi << 8) | src[pos++];
    if (pi >= 0x7FFFFFFFLL) return 0;
    primary[i] = (int32_t)pi + 1;
  }
  int n = blockSize - headerSize;
  if (n > dstCap) return 0;
  if (n == 0) return 1;
  if (!kzo_bwt_inverse_raw(src + pos, n, dst, primary)) return 0;
  *produced = n;
  return 1;
}
