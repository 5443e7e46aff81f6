/* ORACLE (test infrastructure only; parity unpinned -- see kzo.h).
 * FPAQ  K/entropy/FPAQEncoder.java:128-173 (encode), :182-199 (encodeBit), :208-213 (flush), :232-238 (dispose)
 *       K/entropy/FPAQDecoder.java:161-242 (decode), :290-314 (decodeBitV2), :322-335 (read)
 * SRT   K/transform/SRT.java:66-168 (forward), :171-257 (inverse), :259-302 (preprocess), :304-346 (headers)
 */
#include "kzo.h"
#include <stdlib.h>
#include <string.h>

#define FPAQ_TOP 0x00FFFFFFFFFFFFFFULL
#define MASK_24_56 0x00FFFFFFFF000000ULL
#define MASK_0_24 0x0000000000FFFFFFULL
#define MASK_0_32 0x00000000FFFFFFFFULL
#define MASK_0_56 0x00FFFFFFFFFFFFFFULL
#define FPAQ_CHUNK (4 * 1024 * 1024)
#define PSCALE 65536

typedef struct { uint64_t low, high; int32_t probs[4][256]; uint8_t* out; size_t idx; } fpaq_enc;

static inline void fpaq_encode_bit(fpaq_enc* e, int32_t* p, int bit, int pIdx) {
  const uint64_t split = (((e->high - e->low) >> 8) * (uint64_t)(uint32_t)p[pIdx]) >> 8;
  if (bit == 0) { e->low += (split + 1); p[pIdx] -= (p[pIdx] >> 6); }
  else { e->high = e->low + split; p[pIdx] -= ((p[pIdx] - PSCALE + 64) >> 6); }
  while (((e->low ^ e->high) & MASK_24_56) == 0) {                 /* flush :208-213 */
    const uint32_t w = (uint32_t)(e->high >> 24);
    e->out[e->idx] = (uint8_t)(w >> 24); e->out[e->idx + 1] = (uint8_t)(w >> 16);
    e->out[e->idx + 2] = (uint8_t)(w >> 8); e->out[e->idx + 3] = (uint8_t)w;
    e->idx += 4;
    e->low <<= 32;
    e->high = (e->high << 32) | MASK_0_32;
  }
}

int kzo_fpaq_encode(kzo_obs* bs, const uint8_t* block, int count) {
  if (count < 0) return -1;
  if (count == 0) { kzo_obs_write(bs, 0ULL | MASK_0_24, 56); return 0; }   /* dispose() still runs (CompressedOutputStream.java:916) */
  fpaq_enc e;
  e.low = 0; e.high = FPAQ_TOP;
  for (int i = 0; i < 4; i++) for (int j = 0; j < 256; j++) e.probs[i][j] = PSCALE >> 1;
  e.out = (uint8_t*)malloc((size_t)FPAQ_CHUNK + (FPAQ_CHUNK >> 3) + 64);
  int startChunk = 0;
  while (startChunk < count) {
    const int chunkSize = (count - startChunk) < FPAQ_CHUNK ? (count - startChunk) : FPAQ_CHUNK;
    e.idx = 0;
    int32_t* p = e.probs[0];
    for (int i = startChunk; i < startChunk + chunkSize; i++) {
      const int val = block[i];
      const int bits = val + 256;
      fpaq_encode_bit(&e, p, val & 0x80, 1);
      fpaq_encode_bit(&e, p, val & 0x40, bits >> 7);
      fpaq_encode_bit(&e, p, val & 0x20, bits >> 6);
      fpaq_encode_bit(&e, p, val & 0x10, bits >> 5);
      fpaq_encode_bit(&e, p, val & 0x08, bits >> 4);
      fpaq_encode_bit(&e, p, val & 0x04, bits >> 3);
      fpaq_encode_bit(&e, p, val & 0x02, bits >> 2);
      fpaq_encode_bit(&e, p, val & 0x01, bits >> 1);
      p = e.probs[val >> 6];
    }
    kzo_write_varint(bs, (uint32_t)e.idx);
    kzo_obs_write_bytes(bs, e.out, (uint64_t)e.idx * 8);
    startChunk += chunkSize;
    if (startChunk < count) kzo_obs_write(bs, e.low | MASK_0_24, 56);
  }
  kzo_obs_write(bs, e.low | MASK_0_24, 56);                          /* dispose :232-238 */
  free(e.out);
  return count;
}

int kzo_fpaq_decode(kzo_ibs* bs, uint8_t* block, int count) {
  if (count < 0) return -1;
  if (count == 0) return 0;
  uint64_t low = 0, high = FPAQ_TOP, current = 0;
  int32_t probs[4][256];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 256; j++) probs[i][j] = PSCALE >> 1;
  uint8_t* buf = NULL; size_t bufCap = 0;
  int startChunk = 0, ret = count;
  while (startChunk < count) {
    const int szBytes = (int)kzo_read_varint(bs);
    if (szBytes < 0 || szBytes >= 2 * count) { ret = 0; break; }     /* :176-177 */
    size_t bufSize = (size_t)szBytes + ((size_t)szBytes >> 2); if (bufSize < 1024) bufSize = 1024;
    if (bufCap < bufSize) { free(buf); buf = (uint8_t*)malloc(bufSize); bufCap = bufSize; }
    current = kzo_ibs_read(bs, 56);
    memset(buf + szBytes, 0, bufSize - (size_t)szBytes);
    kzo_ibs_read_bytes(bs, buf, (uint64_t)szBytes * 8);
    if (bs->error) { ret = -1; break; }
    const int bufLimit = szBytes;
    int idx = 0;
    const int chunkSize = (count - startChunk) < FPAQ_CHUNK ? (count - startChunk) : FPAQ_CHUNK;
    int32_t* p = probs[0];
    int bad = 0;
    for (int i = startChunk; i < startChunk + chunkSize; i++) {
      int ctx = 1;
      for (int k = 0; k < 8; k++) {                                   /* decodeBitV2 :290-314 */
        const uint64_t split = ((((high - low) >> 8) * (uint64_t)(uint32_t)p[ctx]) >> 8) + low;
        if (split >= current) { high = split; p[ctx] -= ((p[ctx] - PSCALE + 64) >> 6); ctx = (ctx << 1) + 1; }
        else { low = split + 1; p[ctx] -= (p[ctx] >> 6); ctx = ctx << 1; }
        while (((low ^ high) & MASK_24_56) == 0) {                   /* read :322-335 */
          low = (low << 32) & MASK_0_56;
          high = ((high << 32) | MASK_0_32) & MASK_0_56;
          if (idx + 4 > bufLimit) { current = (current << 32) & MASK_0_56; idx = bufLimit + 1; continue; }
          const uint64_t val = ((uint64_t)buf[idx] << 24) | ((uint64_t)buf[idx + 1] << 16) | ((uint64_t)buf[idx + 2] << 8) | (uint64_t)buf[idx + 3];
          current = ((current << 32) | val) & MASK_0_56;
          idx += 4;
        }
      }
      block[i] = (uint8_t)ctx;
      if (idx > szBytes) { bad = 1; break; }
      p = probs[(ctx & 0xFF) >> 6];
    }
    if (bad || idx > szBytes) { ret = 0; break; }
    startChunk += chunkSize;
  }
  free(buf);
  return ret;
}

/* ---------------- SRT ---------------- */
static int srt_preprocess(const int* freqs, uint8_t* symbols) {        /* :259-302 (shell sort; total order) */
  int nb = 0;
  for (int i = 0; i < 256; i++) if (freqs[i] > 0) symbols[nb++] = (uint8_t)i;
  int h = 4;
  while (h < nb) h = h * 3 + 1;
  for (;;) {
    h /= 3;
    for (int i = h; i < nb; i++) {
      const int t = symbols[i];
      int b = i - h;
      while ((b >= 0) && ((freqs[symbols[b]] < freqs[t]) || ((freqs[t] == freqs[symbols[b]]) && (t < symbols[b])))) {
        symbols[b + h] = symbols[b];
        b -= h;
      }
      symbols[b + h] = (uint8_t)t;
    }
    if (h == 1) break;
  }
  return nb;
}

int kzo_srt_forward(const uint8_t* src, int count, uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  if (count == 0) return 1;
  if (dstCap < count + 1024) return 0;                                  /* :81-82 */
  int freqs[256], r2s[256], s2r[256], buckets[256];
  uint8_t symbols[256];
  memset(freqs, 0, sizeof(freqs));
  for (int i = 0, b = 0; i < count;) {                                  /* :95-111 */
    const int c = src[i];
    if (freqs[c] == 0) { r2s[b] = c; s2r[c] = b; b++; }
    int j = i + 1;
    while ((j < count) && (src[j] == c)) j++;
    freqs[c] += (j - i);
    i = j;
  }
  const int nbSymbols = srt_preprocess(freqs, symbols);
  for (int i = 0, bucketPos = 0; i < nbSymbols; i++) { const int c = symbols[i]; buckets[c] = bucketPos; bucketPos += freqs[c]; }
  int h = 0;                                                            /* encodeHeader :304-319 */
  for (int i = 0; i < 256; i++) {
    uint32_t f = (uint32_t)freqs[i];
    while (f >= 128) { dst[h++] = (uint8_t)(0x80 | f); f >>= 7; }
    dst[h++] = (uint8_t)f;
  }
  uint8_t* d = dst + h;
  for (int i = 0; i < count;) {                                         /* :131-163 */
    const int c = src[i];
    int r = s2r[c] & 0xFF;
    int p = buckets[c];
    d[p] = (uint8_t)r;
    p++;
    if (r != 0) {
      do { r2s[r] = r2s[r - 1]; s2r[r2s[r]] = r; r--; } while (r != 0);
      r2s[0] = c; s2r[c] = 0;
    }
    i++;
    while ((i < count) && (src[i] == c)) { d[p] = 0; p++; i++; }
    buckets[c] = p;
  }
  *produced = h + count;
  return 1;
}

int kzo_srt_inverse(const uint8_t* src, int length, uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  if (length == 0) return 1;
  int freqs[256], buckets[256], bucketEnds[256], r2s[256];
  uint8_t symbols[256];
  int h = 0;                                                            /* decodeHeader :321-346 */
  for (int i = 0; i < 256; i++) {
    if (h >= length) return 0;
    int val = src[h++];
    int res = val & 0x7F, shift = 7;
    while (val >= 128) {
      if (h >= length) return 0;
      val = src[h++];
      res = (int)((uint32_t)res | ((uint32_t)(val & 0x7F) << shift));   /* wraps like the Java int */
      if (shift > 21) break;
      shift += 7;
    }
    freqs[i] = res;
  }
  const int count = length - h;
  if (count > dstCap) return 0;
  const uint8_t* s = src + h;
  int nbSymbols = srt_preprocess(freqs, symbols);
  /* fields of a fresh SRT instance (one per block, TransformFactory.newFunction): all zero */
  memset(r2s, 0, sizeof(r2s)); memset(buckets, 0, sizeof(buckets)); memset(bucketEnds, 0, sizeof(bucketEnds));
  for (int i = 0, bucketPos = 0; i < nbSymbols; i++) {                  /* :204-215 */
    const int c = symbols[i];
    /* the reference tests (srcIdx+bucketPos >= input.length) with srcIdx already past the header */
    { const int at = (int)((uint32_t)h + (uint32_t)bucketPos); if ((at < 0) || (at >= length)) return 0; }
    r2s[s[bucketPos]] = c;
    buckets[c] = bucketPos + 1;
    bucketPos = (int)((uint32_t)bucketPos + (uint32_t)freqs[c]);       /* Java int wrap */
    bucketEnds[c] = bucketPos;
  }
  int c = r2s[0];
  for (int i = 0; i < count; i++) {                                     /* :222-250 */
    dst[i] = (uint8_t)c;
    if (buckets[c] < bucketEnds[c]) {
      if (buckets[c] >= count) return 0;                                /* Java would throw on a corrupt stream */
      const int r = s[buckets[c]];
      buckets[c]++;
      if (r == 0) continue;
      for (int k = 0; k < r; k++) r2s[k] = r2s[k + 1];
      r2s[r] = c;
      c = r2s[0];
    } else {
      if (nbSymbols == 1) continue;
      nbSymbols--;
      for (int k = 0; k < nbSymbols; k++) r2s[k] = r2s[k + 1];
      c = r2s[0];
    }
  }
  *produced = count;
  return 1;
}
