/* ORACLE (test infrastructure only; parity unpinned -- see kzo.h).
 * UTF transform restated from K/transform/UTFCodec.java: forward :68-218, inverse :221-305, validate :317-440,
 * pack :443-474, unpackV1 :514-548 (bitstream version >= 4), the (freq, sym) order of SymbolComparator :560-565.
 * Variable names follow the Java.  Reads outside the coded block fail (Java: stale bytes or an exception).
 */
#include "kzo.h"
#include <stdlib.h>
#include <string.h>

#define MIN_BLOCK_SIZE 1024
static const int SIZES[16] = {1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 2, 2, 3, 4};

static int len_seq(int b) {                                      /* LEN_SEQ :32-41 */
  if (b < 0x80) return 1;
  if (b < 0xC2) return 0;
  if (b < 0xE0) return 2;
  if (b < 0xF0) return 3;
  if (b < 0xF5) return 4;
  return 0;
}

static int pack(const uint8_t* in, int idx, int* out) {          /* :443-474 */
  int s = SIZES[(in[idx] >> 4) & 0x0F];
  switch (s) {
    case 1: *out = in[idx]; break;
    case 2: *out = (1 << 19) | (in[idx] << 8) | in[idx + 1]; break;
    case 3: *out = (2 << 19) | ((in[idx] & 0x0F) << 12) | ((in[idx + 1] & 0x3F) << 6) | (in[idx + 2] & 0x3F); break;
    case 4: *out = (4 << 19) | ((in[idx] & 0x07) << 18) | ((in[idx + 1] & 0x3F) << 12) | ((in[idx + 2] & 0x3F) << 6) | (in[idx + 3] & 0x3F); break;
    default: *out = 0; s = 0; break;
  }
  return s;
}

static int unpackV1(int in, uint32_t* value) {                   /* :514-548 */
  switch ((uint32_t)in >> 19) {
    case 0: *value = (uint32_t)in; return 1;
    case 1: *value = (uint32_t)(((in & 0xFF) << 8) | ((in >> 8) & 0xFF)); return 2;
    case 2: *value = (uint32_t)((((in >> 12) & 0x0F) | 0xE0) | ((((in >> 6) & 0x3F) | 0x80) << 8) | (((in & 0x3F) | 0x80) << 16)); return 3;
    case 4: case 5: case 6: case 7:
      *value = (uint32_t)(((in >> 18) & 0x07) | 0xF0) | ((uint32_t)(((in >> 12) & 0x3F) | 0x80) << 8) |
               ((uint32_t)(((in >> 6) & 0x3F) | 0x80) << 16) | ((uint32_t)((in & 0x3F) | 0x80) << 24);
      return 4;
    default: return 0;
  }
}

static int validate(const uint8_t* block, int start, int count) {   /* :317-440 */
  int freqs0[256];
  int (*freqs1)[256] = (int (*)[256])calloc(256, sizeof(int[256]));
  memset(freqs0, 0, sizeof(freqs0));
  int prv = 0;
  const int end = start + count;
  const int end4 = start + (count & -4);
  for (int i = start; i < end4; i += 4) {
    const int cur0 = block[i], cur1 = block[i + 1], cur2 = block[i + 2], cur3 = block[i + 3];
    freqs0[cur0]++; freqs0[cur1]++; freqs0[cur2]++; freqs0[cur3]++;
    freqs1[prv][cur0]++; freqs1[cur0][cur1]++; freqs1[cur1][cur2]++; freqs1[cur2][cur3]++;
    prv = cur3;
    if ((i & 0x0FFF) == start) {                                 /* early check as written (only ever true for start < 4096) */
      int sum = freqs0[0xC0] + freqs0[0xC1];
      for (int j = 0xF5; j <= 0xFF; j++) sum += freqs0[j];
      if (sum != 0) { free(freqs1); return 0; }
    }
  }
  if (end4 != end) {
    for (int i = end4; i < end; i++) { const int cur = block[i]; freqs0[cur]++; freqs1[prv][cur]++; prv = cur; }
    int sum = freqs0[0xC0] + freqs0[0xC1];
    for (int i = 0xF5; i <= 0xFF; i++) sum += freqs0[i];
    if (sum != 0) { free(freqs1); return 0; }
  }
  int sum1 = 0, sum2 = 0;
  for (int i = 0; i < 256; i++) {
    if ((i < 0xA0) || (i > 0xBF)) sum1 += freqs1[0xE0][i];
    if ((i < 0x80) || (i > 0x9F)) sum1 += freqs1[0xED][i];
    if ((i < 0x90) || (i > 0xBF)) sum1 += freqs1[0xF0][i];
    if ((i < 0x80) || (i > 0x8F)) sum1 += freqs1[0xF4][i];
    if ((i < 0x80) || (i > 0xBF)) {
      for (int j = 0xC2; j <= 0xDF; j++) sum1 += freqs1[j][i];
      for (int j = 0xE1; j <= 0xEC; j++) sum1 += freqs1[j][i];
      sum1 += freqs1[0xF1][i]; sum1 += freqs1[0xF2][i]; sum1 += freqs1[0xF3][i];
      sum1 += freqs1[0xEE][i]; sum1 += freqs1[0xEF][i];
    } else {
      sum2 += freqs0[i];
    }
    if (sum1 != 0) { free(freqs1); return 0; }
  }
  free(freqs1);
  return sum2 >= (count / 8);
}

typedef struct { int sym; int freq; } SymbolData;
static int cmp_symbol(const void* a, const void* b) {            /* SymbolComparator :560-565: a total order */
  const SymbolData* l = (const SymbolData*)a; const SymbolData* r = (const SymbolData*)b;
  const int res = l->freq - r->freq;
  return (res != 0) ? res : l->sym - r->sym;
}

int kzo_utf_forward(int* dataType, const uint8_t* src, int count, uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  if (count == 0) return 1;
  if (count < MIN_BLOCK_SIZE) return 0;
  if (dstCap < count + 8192) return 0;                           /* getMaxEncodedLength :308-310 */
  int srcIdx = 0;
  int mustValidate = 1;
  if (dataType) {
    const int dt = *dataType;
    if ((dt != KZO_DT_UNDEFINED) && (dt != KZO_DT_UTF8)) return 0;
    mustValidate = dt != KZO_DT_UTF8;
  }
  const int srcEnd = count - 4;
  int start = 0;
  if ((src[0] == 0xEF) && (src[1] == 0xBB) && (src[2] == 0xBF)) start = 3;
  else { while ((start < 4) && (len_seq(src[start]) == 0)) start++; }
  if (mustValidate && !validate(src, srcIdx + start, srcEnd - start)) return 0;
  if (dataType) *dataType = KZO_DT_UTF8;
  int* aliasMap = (int*)calloc((size_t)1 << 22, sizeof(int));
  SymbolData* symb = (SymbolData*)calloc(32768, sizeof(SymbolData));
  int n = 0, res = 1, val = 0;
  for (int i = srcIdx + start; i < srcEnd;) {
    const int s = pack(src, i, &val);
    res = s != 0;
    res &= ((s != 3) || ((src[i + 2] >= 0x80) && (src[i + 2] <= 0xBF)));
    const int val2 = (src[i + 2] << 8) | src[i + 3];
    res &= ((s != 4) || ((val2 & 0xC0C0) == 0x8080));
    if (aliasMap[val] == 0) {
      if (n < 32768) symb[n].sym = val;
      n++;
      res &= (n < 32768);
    }
    if (!res) break;
    aliasMap[val]++;
    i += s;
  }
  const int maxTarget = count - (count / 10);
  if (!res || (n == 0) || ((3 * n + 6) >= maxTarget)) { free(aliasMap); free(symb); return 0; }
  for (int i = 0; i < n; i++) symb[i].freq = aliasMap[symb[i].sym];
  qsort(symb, (size_t)n, sizeof(SymbolData), cmp_symbol);        /* ranks by increasing (freq, sym) */
  int dstIdx = 2;
  dst[dstIdx++] = (uint8_t)(n >> 8);
  dst[dstIdx++] = (uint8_t)n;
  int estimate = dstIdx + 6;
  for (int i = 0; i < n; i++) {
    const SymbolData* r = &symb[n - 1 - i];
    const int s = r->sym;
    dst[dstIdx] = (uint8_t)(s >> 16); dst[dstIdx + 1] = (uint8_t)(s >> 8); dst[dstIdx + 2] = (uint8_t)s;
    dstIdx += 3;
    estimate += ((i < 128) ? r->freq : 2 * r->freq);
    aliasMap[s] = (i < 128) ? i : (0x10080 | ((i << 1) & 0xFF00) | (i & 0x7F));
  }
  if (estimate >= maxTarget) { free(aliasMap); free(symb); return 0; }
  /* The map (3 n bytes) is not part of `estimate`: map + aliases can exceed count + 8192 bytes (getMaxEncodedLength, the size the
     buffers are given, Sequence.java:82-90) for small blocks with thousands of distinct code points.  The reference then either
     runs over its array (ArrayIndexOutOfBoundsException: the block fails) or, when the array was sized by an earlier, larger
     block, finishes and declines at the end (dstIdx >= maxTarget, :214).  Restated as the second outcome -- decline -- decided
     before anything is written: the output is at least map + start + aliases + 1 tail byte.  (The HIP path does the same;
     INTEGRATION.md 4.) */
  if ((long long)dstIdx + start + (estimate - 10) + 1 >= (long long)maxTarget) { free(aliasMap); free(symb); return 0; }
  for (int i = 0; i < start; i++) dst[dstIdx++] = src[srcIdx + i];
  srcIdx += start;
  while (srcIdx < srcEnd) {
    srcIdx += pack(src, srcIdx, &val);
    const int alias = aliasMap[val];
    dst[dstIdx++] = (uint8_t)alias;
    dst[dstIdx] = (uint8_t)((uint32_t)alias >> 8);
    dstIdx += ((uint32_t)alias >> 16);
  }
  dst[0] = (uint8_t)start;
  dst[1] = (uint8_t)(srcIdx - srcEnd);
  while (srcIdx < srcEnd + 4) dst[dstIdx++] = src[srcIdx++];
  free(aliasMap); free(symb);
  *produced = dstIdx;
  return dstIdx < maxTarget;
}

int kzo_utf_inverse(const uint8_t* src, int count, uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  if (count == 0) return 1;
  if (count < 4) return 0;
  int srcIdx = 0, dstIdx = 0;
  const int start = src[0] & 0x03;
  const int adjust = src[1] & 0x03;
  const int n = (src[2] << 8) + src[3];
  const int srcEnd = count - 4 + adjust;
  const int dstEnd = dstCap - 4;                                 /* output.length - 4 */
  if ((n == 0) || (n >= 32768) || (3 * n >= count)) return 0;
  typedef struct { uint32_t value; int length; } UTFSymbol;
  UTFSymbol* m = (UTFSymbol*)calloc(32768, sizeof(UTFSymbol));
  srcIdx += 4;
  for (int i = 0; i < n; i++) {
    if (srcIdx + 3 > count) { free(m); return 0; }
    const int s = (src[srcIdx] << 16) | (src[srcIdx + 1] << 8) | src[srcIdx + 2];
    const int sl = unpackV1(s, &m[i].value);
    if (sl == 0) { free(m); return 0; }
    m[i].length = sl;
    srcIdx += 3;
  }
  if (dstEnd < 0) { free(m); return 0; }
  for (int i = 0; i < start; i++) { if (srcIdx >= count) { free(m); return 0; } dst[dstIdx++] = src[srcIdx++]; }
  int res = 1;
  while ((srcIdx < srcEnd) && (dstIdx < dstEnd)) {
    int alias = src[srcIdx++];
    if (alias >= 128) { if (srcIdx >= count) { res = 0; break; } alias = (src[srcIdx++] << 7) + (alias & 0x7F); }
    if (alias >= n) { res = 0; break; }                          /* m[alias] == null -> NullPointerException */
    const UTFSymbol* s = &m[alias];
    dst[dstIdx] = (uint8_t)s->value; dst[dstIdx + 1] = (uint8_t)(s->value >> 8);            /* LittleEndian.writeInt32 */
    dst[dstIdx + 2] = (uint8_t)(s->value >> 16); dst[dstIdx + 3] = (uint8_t)(s->value >> 24);
    dstIdx += s->length;
  }
  if (res) {
    if ((srcIdx < srcEnd) || (dstIdx >= dstEnd - count + srcEnd)) res = 0;
    else for (int i = srcEnd; i < count; i++) { if (srcIdx >= count) { res = 0; break; } dst[dstIdx++] = src[srcIdx++]; }
  }
  free(m);
  *produced = dstIdx;
  return res;
}
