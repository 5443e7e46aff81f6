/* ORACLE (test infrastructure only; parity unpinned -- see kzo.h).
 * Canonical, length-limited (12 bit) Huffman codec restated from
 *   K/entropy/HuffmanEncoder.java:103-178 (updateFrequencies), :191-273 (limitCodeLengths),
 *     :285-308 (computeCodeLengths), :317-376 (Moffat-Katajainen phases), :380-416 (encode), :419-493 (encodeChunk)
 *   K/entropy/HuffmanCommon.java:71-111 (generateCanonicalCodes)
 *   K/entropy/ExpGolombEncoder.java:123-131 / ExpGolombDecoder.java:41-63 (signed Exp-Golomb of length deltas)
 *   K/entropy/HuffmanDecoder.java:115-154 (readLengths), :353-390 (decodeV6), :404-587 (decodeChunk)
 */
#include "kzo.h"
#include <stdlib.h>
#include <string.h>

#define HUF_CHUNK 16384
#define HUF_MAXLEN 12

/* signed Exp-Golomb: 0 -> '1'; else a=|v|, k=floor(log2(a+1)), m=a+1-2^k: k zeros, '1', m (k bits), sign.
 * Equals CACHE_VALUES[1][v & 0xFF] for |v| <= 126 (checked against the table entries for 1,-1,2,-2,3). */
static void eg_put(kzo_obs* bs, int v) {
  if (v == 0) { kzo_obs_write(bs, 1, 1); return; }
  int a = v < 0 ? -v : v;
  int k = 31 - __builtin_clz((unsigned)(a + 1));
  int m = a + 1 - (1 << k);
  uint32_t bits = (1u << (k + 1)) | ((uint32_t)m << 1) | (v < 0 ? 1u : 0u);
  kzo_obs_write(bs, bits, 2 * k + 2);
}
/* ExpGolombDecoder.java:41-58 (signed), as Java computes it on ANY input: the zeros are counted until a 1 arrives (the end of the
 * stream throws), readBits takes 1..64 bits (more: IllegalArgumentException), `1 << log2` is an int shift (count mod 32, sign
 * extended) and the result is cast to byte.  A damaged header can therefore decode to a small delta after 13 or 40 zeros. */
static int eg_get(kzo_ibs* bs) {
  if (kzo_ibs_read(bs, 1) == 1) return 0;
  int log2 = 1;
  while (kzo_ibs_read(bs, 1) == 0) { log2++; if (bs->error) return 0; }
  if (bs->error) return 0;
  if (log2 + 1 > 64) { bs->error = 1; return 0; }       /* DefaultInputBitStream.java:98-99 */
  uint64_t res = kzo_ibs_read(bs, log2 + 1);
  uint64_t sgn = res & 1;
  res = (res >> 1) + (uint64_t)(int64_t)(int32_t)(1u << (log2 & 31)) - 1;
  return (int)(int8_t)((res - sgn) ^ (0 - sgn));
}

static int cmp_int(const void* a, const void* b) { int x = *(const int*)a, y = *(const int*)b; return (x > y) - (x < y); }

static void phase1(int* data, int n) {                 /* HuffmanEncoder.java:317-340 */
  for (int s = 0, r = 0, t = 0; t < n - 1; t++) {
    int sum = 0;
    for (int i = 0; i < 2; i++) {
      if ((s >= n) || ((r < t) && (data[r] < data[s]))) { sum += data[r]; data[r] = t; r++; continue; }
      sum += data[s];
      if (s > t) data[s] = 0;
      s++;
    }
    data[t] = sum;
  }
}
static int phase2(int* data, int n) {                  /* :342-376 */
  if (n < 2) return 0;
  int levelTop = n - 2, depth = 1, i = n, totalNodesAtLevel = 2;
  while (i > 0) {
    int k = levelTop;
    while ((k > 0) && (data[k - 1] >= levelTop)) k--;
    const int internalNodesAtLevel = levelTop - k;
    const int leavesAtLevel = totalNodesAtLevel - internalNodesAtLevel;
    for (int j = 0; j < leavesAtLevel; j++) data[--i] = depth;
    totalNodesAtLevel = internalNodesAtLevel << 1;
    levelTop = k;
    depth++;
  }
  return depth - 1;
}
static int compute_code_lengths(int16_t* sizes, int* ranks, int count) {    /* :285-308 */
  qsort(ranks, (size_t)count, sizeof(int), cmp_int);
  int freqs[256];
  for (int i = 0; i < count; i++) { freqs[i] = (int)((uint32_t)ranks[i] >> 8); ranks[i] &= 0xFF; if (freqs[i] == 0) return 0; }
  phase1(freqs, count);
  const int maxCodeLen = phase2(freqs, count);
  for (int i = 0; i < count; i++) sizes[ranks[i]] = (int16_t)freqs[i];
  return maxCodeLen;
}

/* EntropyUtils.normalizeFrequencies on a compacted array of `alen` entries (HuffmanEncoder.java:263) */
static int normalize_n(int* freqs, int* alphabet, int alen, int totalFreq, int scale) {
  if (alen == 0 || totalFreq == 0) return 0;
  int alphabetSize = 0;
  if (totalFreq == scale) { for (int i = 0; i < alen; i++) if (freqs[i] != 0) alphabet[alphabetSize++] = i; return alphabetSize; }
  int sumScaledFreq = 0, sumFreq = 0, idxMax = 0;
  for (int i = 0; i < alen; i++) {
    alphabet[i] = 0;
    int f = freqs[i];
    if (f == 0) continue;
    int64_t sf = (int64_t)freqs[i] * scale;
    int scaledFreq = (sf <= totalFreq) ? 1 : (int)((sf + ((int64_t)totalFreq >> 1)) / (int64_t)totalFreq);
    alphabet[alphabetSize++] = i;
    sumScaledFreq += scaledFreq; freqs[i] = scaledFreq; sumFreq += f;
    if (scaledFreq > freqs[idxMax]) idxMax = i;
    if (sumFreq >= totalFreq) break;
  }
  if (alphabetSize == 0) return 0;
  if (alphabetSize == 1) { freqs[alphabet[0]] = scale; return 1; }
  if (sumScaledFreq == scale) return alphabetSize;
  int delta = sumScaledFreq - scale;
  int errThr = freqs[idxMax] >> 4;
  if ((delta < 0 ? -delta : delta) <= errThr) { freqs[idxMax] -= delta; return alphabetSize; }
  if (delta < 0) { delta += errThr; freqs[idxMax] += errThr; } else { delta -= errThr; freqs[idxMax] -= errThr; }
  int inc = (delta > 0) ? -1 : 1;
  delta = delta < 0 ? -delta : delta;
  int round = 0;
  while ((++round < 6) && (delta > 0)) {
    int adjustments = 0;
    for (int i = 0; i < alphabetSize; i++) {
      int idx = alphabet[i];
      if (freqs[idx] <= 2) continue;
      freqs[idx] += inc; adjustments++; delta--;
      if (delta == 0) break;
    }
    if (adjustments == 0) break;
  }
  int v = freqs[idxMax] - delta;
  freqs[idxMax] = v > 1 ? v : 1;
  return alphabetSize;
}

static int limit_code_lengths(const int* alphabet, int* freqs, int16_t* sizes, int* ranks, int count) {   /* :191-273 */
  int n = 0, debt = 0;
  while (n < count && sizes[ranks[n]] >= HUF_MAXLEN) { debt += (sizes[ranks[n]] - HUF_MAXLEN); sizes[ranks[n]] = HUF_MAXLEN; n++; }
  int ll[6][256], head[6] = {0, 0, 0, 0, 0, 0}, tail[6] = {0, 0, 0, 0, 0, 0};   /* FIFO lists (LinkedList add/removeFirst) */
  while (n < count) {
    const int idx = HUF_MAXLEN - 1 - sizes[ranks[n]];
    if ((idx >= 6) || (debt < (1 << idx))) break;
    ll[idx][tail[idx]++] = ranks[n];
    n++;
  }
  int idx = 5;
  while ((debt > 0) && (idx >= 0)) {
    if ((head[idx] == tail[idx]) || (debt < (1 << idx))) { idx--; continue; }
    const int r = ll[idx][head[idx]++];
    sizes[r]++;
    debt -= (1 << idx);
  }
  idx = 0;
  while ((debt > 0) && (idx < 6)) {
    if (head[idx] == tail[idx]) { idx++; continue; }
    const int r = ll[idx][head[idx]++];
    sizes[r]++;
    debt -= (1 << idx);
  }
  if (debt > 0) {                                       /* :250-270 slow fallback */
    int f[256], symbols[256], totalFreq = 0;
    for (int i = 0; i < count; i++) { f[i] = freqs[alphabet[i]]; totalFreq += f[i]; }
    normalize_n(f, symbols, count, totalFreq, HUF_CHUNK >> 3);
    for (int i = 0; i < count; i++) { freqs[alphabet[i]] = f[i]; ranks[i] = (int)(((uint32_t)f[i] << 8) | (uint32_t)alphabet[i]); }
    return compute_code_lengths(sizes, ranks, count);
  }
  return HUF_MAXLEN;
}

static int canonical_codes(const int16_t* sizes, uint32_t* codes, int* symbols, int count) {   /* HuffmanCommon.java:71-111 */
  if (count > 1) {
    uint8_t buf[(14 << 8) + 256]; memset(buf, 0, sizeof(buf));
    for (int i = 0; i < count; i++) {
      const int s = symbols[i];
      if (((s & 0xFF) != s) || (sizes[s] > HUF_MAXLEN) || sizes[s] < 1) return -1;
      buf[((sizes[s] - 1) << 8) | s] = 1;
    }
    int n = 0;
    for (int i = 0; i < (int)sizeof(buf); i++) { if (!buf[i]) continue; symbols[n++] = i & 0xFF; if (n == count) break; }
  }
  uint32_t code = 0;
  int curLen = sizes[symbols[0]];
  for (int i = 0; i < count; i++) {
    const int s = symbols[i];
    code <<= (sizes[s] - curLen);
    curLen = sizes[s];
    codes[s] = code;
    code++;
  }
  return count;
}

int kzo_huffman_encode(kzo_obs* bs, const uint8_t* block, int count) {
  if (count < 0) return -1;
  if (count == 0) return 0;
  int startChunk = 0;
  uint8_t* frag = (uint8_t*)malloc(4 * (HUF_CHUNK / 4) * 2 + 64);
  while (startChunk < count) {
    const int sizeChunk = (count - startChunk) < HUF_CHUNK ? (count - startChunk) : HUF_CHUNK;
    const uint8_t* blk = block + startChunk;
    if (sizeChunk < 32) { kzo_obs_write_bytes(bs, blk, (uint64_t)sizeChunk * 8); startChunk += sizeChunk; continue; }   /* :400-402 */
    int freqs[256]; memset(freqs, 0, sizeof(freqs));
    for (int i = 0; i < sizeChunk; i++) freqs[blk[i]]++;
    /* updateFrequencies :103-178 */
    int alphabet[256], n = 0;
    uint32_t codes[256]; int16_t sizes[256];
    memset(codes, 0, sizeof(codes)); memset(sizes, 0, sizeof(sizes));
    for (int i = 0; i < 256; i++) if (freqs[i] > 0) alphabet[n++] = i;
    kzo_encode_alphabet(bs, alphabet, n);
    if (n == 1) { codes[alphabet[0]] = 1u << 24; sizes[alphabet[0]] = 1; }
    else {
      int ranks[256];
      for (int i = 0; i < n; i++) ranks[i] = (int)(((uint32_t)freqs[alphabet[i]] << 8) | (uint32_t)alphabet[i]);
      int maxCodeLen = compute_code_lengths(sizes, ranks, n);
      if (maxCodeLen == 0) { free(frag); return -1; }
      if (maxCodeLen > HUF_MAXLEN) { maxCodeLen = limit_code_lengths(alphabet, freqs, sizes, ranks, n); if (maxCodeLen == 0) { free(frag); return -1; } }
      if (maxCodeLen > HUF_MAXLEN) { for (int i = 0; i < n; i++) { codes[alphabet[i]] = (uint32_t)i; sizes[alphabet[i]] = 8; } }   /* :146-155 */
      else canonical_codes(sizes, codes, ranks, n);
    }
    int prevSize = 2;
    for (int i = 0; i < n; i++) {                        /* :163-174 */
      const int s = alphabet[i];
      codes[s] |= ((uint32_t)sizes[s] << 24);
      eg_put(bs, (int)(int8_t)(sizes[s] - prevSize));
      prevSize = sizes[s];
    }
    if (n > 1) {                                         /* encodeChunk :419-493 */
      const int szFrag = sizeChunk / 4;
      uint64_t nbBits[4];
      kzo_obs f[4];
      for (int j = 0; j < 4; j++) {
        kzo_obs_wrap(&f[j], frag + (size_t)j * (HUF_CHUNK / 2 + 16), HUF_CHUNK / 2 + 16);
        const uint8_t* p = blk + j * szFrag;
        for (int i = 0; i < szFrag; i++) { const uint32_t c = codes[p[i]]; kzo_obs_write(&f[j], c & 0xFFFFFF, (int)(c >> 24)); }
        nbBits[j] = f[j].nbits;
      }
      for (int j = 0; j < 4; j++) kzo_write_varint(bs, (uint32_t)nbBits[j]);
      for (int j = 0; j < 4; j++) kzo_obs_write_bytes(bs, f[j].buf, nbBits[j]);
      for (int i = 4 * szFrag; i < sizeChunk; i++) kzo_obs_write(bs, blk[i], 8);
    }
    startChunk += sizeChunk;
  }
  free(frag);
  return count;
}

int kzo_huffman_decode(kzo_ibs* bs, uint8_t* block, int count) {
  if (count < 0) return -1;
  if (count == 0) return 0;
  int startChunk = 0;
  uint16_t* table = (uint16_t*)malloc(sizeof(uint16_t) << HUF_MAXLEN);
  int ret = count;
  while (startChunk < count) {
    const int sizeChunk = (count - startChunk) < HUF_CHUNK ? (count - startChunk) : HUF_CHUNK;
    uint8_t* blk = block + startChunk;
    if (sizeChunk < 32) { kzo_ibs_read_bytes(bs, blk, (uint64_t)sizeChunk * 8); if (bs->error) { ret = -1; break; } startChunk += sizeChunk; continue; }
    int alphabet[256];
    const int n = kzo_decode_alphabet(bs, alphabet);    /* readLengths :115-154 */
    if (n <= 0 || bs->error) { ret = startChunk; break; }
    int16_t sizes[256]; uint32_t codes[256];
    memset(sizes, 0, sizeof(sizes)); memset(codes, 0, sizeof(codes));
    int curSize = 2, bad = 0;
    for (int i = 0; i < n; i++) {
      curSize += eg_get(bs);
      if ((curSize <= 0) || (curSize > HUF_MAXLEN)) { bad = 1; break; }
      sizes[alphabet[i]] = (int16_t)curSize;
    }
    if (bad || bs->error) { ret = -1; break; }
    int syms[256]; memcpy(syms, alphabet, sizeof(int) * (size_t)n);
    if (canonical_codes(sizes, codes, syms, n) < 0) { ret = -1; break; }
    if (n == 1) { memset(blk, alphabet[0], (size_t)sizeChunk); startChunk += sizeChunk; continue; }   /* :374-377 */
    /* buildDecodingTables :162-191: prefix-indexed table of (len<<8)|sym, default 7 */
    for (int i = 0; i < (1 << HUF_MAXLEN); i++) table[i] = 7;
    for (int i = 0; i < n; i++) {
      const int s = alphabet[i];
      const int len = sizes[s];
      int idx = (int)(codes[s] << (HUF_MAXLEN - len));
      const int end = idx + (1 << (HUF_MAXLEN - len));
      if (end > (1 << HUF_MAXLEN)) { bad = 1; break; }            /* over-subscribed lengths: Java throws on table[idx] */
      while (idx < end) table[idx++] = (uint16_t)((len << 8) | s);
    }
    if (bad) { ret = -1; break; }
    uint32_t szBits[4];
    for (int j = 0; j < 4; j++) szBits[j] = kzo_read_varint(bs);
    const int szFrag = sizeChunk / 4;
    uint64_t pos = bs->pos;
    for (int j = 0; j < 4 && !bad; j++) {
      uint64_t p = pos, endp = pos + szBits[j];
      if (endp > bs->nbits) { bad = 1; break; }
      for (int i = 0; i < szFrag; i++) {
        /* peek HUF_MAXLEN bits, zero padded past the fragment end (the decoder's buffer is zero filled) */
        uint32_t v = 0;
        for (int b = 0; b < HUF_MAXLEN; b++) { uint64_t q = p + (uint64_t)b; uint32_t bit = (q < endp) ? ((bs->buf[q >> 3] >> (7 - (q & 7))) & 1u) : 0u; v = (v << 1) | bit; }
        const uint16_t t = table[v];
        blk[j * szFrag + i] = (uint8_t)t;
        p += (uint64_t)(t >> 8);
      }
      /* decodeChunk's return value (HuffmanDecoder.java: `((idx - base) << 3) - (bs + MAX_SYMBOL_SIZE_V4) == szBits` for each of the four
       * fragments): the bits a fragment's symbols took must be exactly its stated size, else decodeV6 returns startChunk - blkptr */
      if (p != endp) { bad = 1; break; }
      pos = endp;
    }
    if (bad) { ret = startChunk; break; }
    bs->pos = pos;
    for (int i = 4 * szFrag; i < sizeChunk; i++) blk[i] = (uint8_t)kzo_ibs_read(bs, 8);
    if (bs->error) { ret = -1; break; }
    startChunk += sizeChunk;
  }
  free(table);
  return ret;
}
