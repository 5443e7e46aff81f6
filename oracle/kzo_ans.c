/* ORACLE (test infrastructure only; parity unpinned -- see kzo.h).
 * ANS0 (order-0 range ANS) restated from
 *   K/entropy/ANSRangeEncoder.java:263-305 (encode), :171-200 (updateFrequencies),
 *     :211-252 (encodeHeader), :315-328 (encodeSymbol), :337-407 (encodeChunk), :473-496 (Symbol.reset)
 *   K/entropy/ANSRangeDecoder.java:189-236 (decode), :357-440 (decodeChunkV2), :452-544 (decodeHeader)
 * plus the raw (NONE) codec K/entropy/NullEntropyEncoder.java:66-81.
 */
#include "kzo.h"
#include <stdlib.h>
#include <string.h>

#define ANS_TOP (1 << 15)
#define ANS_CHUNK 16384
#define ANS_LOG_RANGE 12
#define ANS_MAX_CHUNK (1 << 27)

typedef struct { int32_t xMax, bias, cmplFreq, invShift; uint64_t invFreq; } enc_sym;

/* K/entropy/ANSRangeEncoder.java:473-496 */
static void enc_sym_reset(enc_sym* s, int cumFreq, int freq, int lr) {
  if (freq >= (1 << lr)) freq = (1 << lr) - 1;
  s->xMax = (int32_t)(((uint32_t)(ANS_TOP >> lr) << 16) * (uint32_t)freq);
  s->cmplFreq = (1 << lr) - freq;
  if (freq < 2) {
    s->invFreq = 0xFFFFFFFFULL; s->invShift = 32; s->bias = cumFreq + (1 << lr) - 1;
  } else {
    int shift = 0;
    while (freq > (1 << shift)) shift++;
    s->invFreq = (((1ULL << (shift + 31)) + (uint64_t)freq - 1) / (uint64_t)freq) & 0xFFFFFFFFULL;
    s->invShift = 32 + shift - 1;
    s->bias = cumFreq;
  }
}

/* K/entropy/ANSRangeEncoder.java:211-252 */
static void ans_encode_header(kzo_obs* bs, int alphabetSize, const int* alphabet, const int* freqs, int lr) {
  int encoded = kzo_encode_alphabet(bs, alphabet, alphabetSize);
  if (encoded <= 1) return;
  int chkSize = (alphabetSize >= 64) ? 8 : 6;
  int llr = 3;
  while ((1 << llr) <= lr) llr++;
  for (int i = 1; i < alphabetSize; i += chkSize) {
    int max = freqs[alphabet[i]] - 1, logMax = 0;
    int endj = (i + chkSize < alphabetSize) ? i + chkSize : alphabetSize;
    for (int j = i + 1; j < endj; j++)
      if (freqs[alphabet[j]] - 1 > max) max = freqs[alphabet[j]] - 1;
    while ((1 << logMax) <= max) logMax++;
    kzo_obs_write(bs, (uint64_t)logMax, llr);
    if (logMax == 0) continue;
    for (int j = i; j < endj; j++) kzo_obs_write(bs, (uint64_t)(freqs[alphabet[j]] - 1), logMax);
  }
}

int kzo_ans0_encode(kzo_obs* bs, const uint8_t* block, int count) {
  if (count < 0) return -1;
  if (count <= 32) { kzo_obs_write_bytes(bs, block, (uint64_t)count * 8); return count; }  /* :267-270 */
  const int lr = ANS_LOG_RANGE;
  int sizeChunk = ANS_CHUNK;
  int bufLen = sizeChunk + (sizeChunk >> 3);                  /* :281 */
  if (bufLen > 2 * count) bufLen = 2 * count;
  if (bufLen < 65536) bufLen = 65536;
  uint8_t* buffer = (uint8_t*)malloc((size_t)bufLen);
  enc_sym symb[256];
  int freqs[257], alphabet[256];
  int startChunk = 0;
  while (startChunk < count) {
    int endChunk = startChunk + sizeChunk < count ? startChunk + sizeChunk : count;
    /* rebuildStatistics :419-449 + updateFrequencies :171-200 */
    memset(freqs, 0, sizeof(freqs));
    for (int i = startChunk; i < endChunk; i++) freqs[block[i]]++;
    freqs[256] = endChunk - startChunk;
    kzo_obs_write(bs, (uint64_t)(lr - 8), 3);
    int alphabetSize = kzo_normalize_freqs(freqs, alphabet, freqs[256], 1 << lr);
    if (alphabetSize > 0) {
      int sum = 0;
      for (int i = 0, c = 0; (i < 256) && (c < alphabetSize); i++) {
        if (freqs[i] == 0) continue;
        enc_sym_reset(&symb[i], sum, freqs[i], lr);
        sum += freqs[i]; c++;
      }
    }
    ans_encode_header(bs, alphabetSize, alphabet, freqs, lr);
    if (alphabetSize <= 1) { startChunk = endChunk; continue; }   /* :295-298 */
    /* encodeChunk :337-407 */
    int32_t st[4] = { ANS_TOP, ANS_TOP, ANS_TOP, ANS_TOP };
    int n = bufLen - 1;
    int start = startChunk, end = endChunk;
    int end4 = start + ((end - start) & -4);
    for (int i = end - 1; i >= end4; i--) buffer[n--] = block[i];
    int idx = n;
    for (int i = end4 - 1; i > start; i -= 4) {
      for (int k = 0; k < 4; k++) {                           /* st0..st3 <- block[i], [i-1], [i-2], [i-3] */
        const enc_sym* sym = &symb[block[i - k]];
        int32_t s = st[k];
        int x = (s >= sym->xMax) ? 1 : 0;                     /* encodeSymbol :315-328 */
        buffer[idx] = (uint8_t)s; idx -= x;
        buffer[idx] = (uint8_t)(s >> 8); idx -= x;
        s >>= (-x & 16);
        int32_t q = (int32_t)(((int64_t)s * (int64_t)sym->invFreq) >> sym->invShift);
        st[k] = (int32_t)((uint32_t)s + (uint32_t)sym->bias + (uint32_t)q * (uint32_t)sym->cmplFreq);
      }
    }
    n = idx + 1;
    kzo_write_varint(bs, (uint32_t)(bufLen - n));
    for (int k = 0; k < 4; k++) kzo_obs_write(bs, (uint32_t)st[k], 32);
    if (bufLen != n) kzo_obs_write_bytes(bs, buffer + n, (uint64_t)(bufLen - n) * 8);
    startChunk = endChunk;
  }
  free(buffer);
  return count;
}

int kzo_ans0_decode(kzo_ibs* bs, uint8_t* block, int count) {
  if (count < 0) return -1;
  if (count <= 32) { kzo_ibs_read_bytes(bs, block, (uint64_t)count * 8); return bs->error ? -1 : count; }
  int sizeChunk = ANS_CHUNK;
  int freqs[256], alphabet[256], dfreq[256], dcum[256];
  uint8_t* f2s = (uint8_t*)malloc(1 << 15);
  int bufLen = 0;                                               /* this.buffer starts empty and only grows (:371-374) */
  uint8_t* buffer = NULL;
  int startChunk = 0, ret = count;
  /* When decodeChunkV2 returns false the reference breaks and still returns count (:229-231): the bytes it never wrote
     keep whatever the caller's array held.  Callers here pass a zeroed array, which is what the HIP path produces too. */
  while (startChunk < count) {
    int endChunk = startChunk + sizeChunk < count ? startChunk + sizeChunk : count;
    /* decodeHeader :452-544 */
    int logRange = 8 + (int)kzo_ibs_read(bs, 3);
    int scale = 1 << logRange;
    int alphabetSize = kzo_decode_alphabet(bs, alphabet);
    if (bs->error) { ret = -1; break; }
    if (alphabetSize == 0) { ret = startChunk; break; }        /* :214-215 */
    {
      int llr = 3;
      while ((1 << llr) <= logRange) llr++;
      memset(freqs, 0, sizeof(freqs));
      int chkSize = (alphabetSize >= 64) ? 8 : 6, sum = 0, bad = 0;
      for (int i = 1; i < alphabetSize && !bad; i += chkSize) {
        int logMax = (int)kzo_ibs_read(bs, llr);
        if ((1 << logMax) > scale) { bad = 1; break; }
        int endj = (i + chkSize < alphabetSize) ? i + chkSize : alphabetSize;
        for (int j = i; j < endj; j++) {
          int freq = (logMax == 0) ? 1 : (int)(1 + kzo_ibs_read(bs, logMax));
          if ((freq <= 0) || (freq >= scale)) { bad = 1; break; }
          freqs[alphabet[j]] = freq; sum += freq;
        }
      }
      if (bad || scale <= sum || bs->error) { ret = -1; break; }
      freqs[alphabet[0]] = scale - sum;
      sum = 0;
      for (int i = 0; i < 256; i++) {
        if (freqs[i] == 0) continue;
        for (int j = freqs[i] - 1; j >= 0; j--) f2s[sum + j] = (uint8_t)i;
        dcum[i] = sum;
        dfreq[i] = (freqs[i] >= scale) ? scale - 1 : freqs[i];  /* Symbol.reset :576-579 */
        sum += freqs[i];
      }
    }
    if (alphabetSize == 1) {                                    /* :217-220 */
      for (int i = startChunk; i < endChunk; i++) block[i] = (uint8_t)alphabet[0];
      startChunk = endChunk; continue;
    }
    /* decodeChunkV2 :357-440 */
    uint32_t sz = kzo_read_varint(bs);
    if (sz >= ANS_MAX_CHUNK) break;
    int32_t st0 = (int32_t)kzo_ibs_read(bs, 32), st1 = (int32_t)kzo_ibs_read(bs, 32);
    int32_t st2 = (int32_t)kzo_ibs_read(bs, 32), st3 = (int32_t)kzo_ibs_read(bs, 32);
    int start = startChunk, end = endChunk;
    int minBuf = 2 * (end - start) > 256 ? 2 * (end - start) : 256;
    if (bufLen < minBuf) { free(buffer); bufLen = minBuf; buffer = (uint8_t*)malloc((size_t)bufLen); }
    if ((uint32_t)bufLen < sz) { ret = -1; break; }              /* readBits past the array end throws */
    memset(buffer, 0, (size_t)bufLen);
    kzo_ibs_read_bytes(bs, buffer, (uint64_t)sz * 8);
    if (bs->error) { ret = -1; break; }
    int n = 0;
    const int mask = scale - 1;
    int end4 = start + ((end - start) & -4);
#define DEC(stv, outpos) do { int cur = f2s[(stv) & mask]; block[outpos] = (uint8_t)cur;              \
      (stv) = (int32_t)((uint32_t)dfreq[cur] * ((uint32_t)(stv) >> logRange) + ((uint32_t)(stv) & mask) - (uint32_t)dcum[cur]); \
      if ((stv) < ANS_TOP) { (stv) = (int32_t)(((uint32_t)(stv) << 8) | buffer[n]);                     \
        (stv) = (int32_t)(((uint32_t)(stv) << 8) | buffer[n + 1]); n += 2; } } while (0)
    for (int i = start; i < end4; i += 4) {
      if (n + 8 > bufLen) break;
      DEC(st3, i); DEC(st2, i + 1); DEC(st1, i + 2); DEC(st0, i + 3);
    }
#undef DEC
    for (int i = end4; i < end; i++) block[i] = buffer[n++];
    if ((uint32_t)n != sz) break;                               /* :439 -> decode() breaks, returns count */
    startChunk = endChunk;
  }
  free(buffer); free(f2s);
  return ret;
}

/* K/entropy/NullEntropyEncoder.java:66-81 / NullEntropyDecoder */
int kzo_null_encode(kzo_obs* bs, const uint8_t* block, int count) {
  kzo_obs_write_bytes(bs, block, (uint64_t)count * 8);
  return count;
}
int kzo_null_decode(kzo_ibs* bs, uint8_t* block, int count) {
  kzo_ibs_read_bytes(bs, block, (uint64_t)count * 8);
  return bs->error ? -1 : count;
}
