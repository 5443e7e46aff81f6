/* ORACLE (test infrastructure only; parity unpinned -- see kzo.h).
 * TEXT transform restated from K/transform/TextCodec.java:
 *   static dictionary :88-179 (createDictionary :215-244), computeStats :269-384, detectType :387-466,
 *   TextCodec.forward/inverse (variant bit 0x10 of the first byte) :482-531,
 *   TextCodec1 (escape tokens 0x0F / 0x0E + varint index; chosen for FPAQ, CM, TPAQ: TransformFactory.java:275-286)
 *     reset :578-613, forward :616-787, expandDictionary :790-803, emitSymbols :806-847, emitWordIndex :850-863, inverse :866-1028
 *   TextCodec2 (index bytes with the high bit set; chosen for NONE, ANS0, HUFFMAN, RANGE)
 *     reset :1092-1119, forward :1122-1285, emitSymbols :1304-1367, emitWordIndex :1370-1394, inverse :1397-1603
 * The structure (DictEntry objects, dictMap of references, dictList) and the variable names follow the Java line by line.
 * Java `byte` is signed: isText() and the hash multiply see negative values for bytes >= 0x80 (sb() below).
 * Reads outside [0, count) of the coded block, which in Java return stale buffer bytes or throw, fail here.
 */
#include "kzo.h"
#include "kzo_text_dict.h"
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define THRESHOLD1 128
#define THRESHOLD2 (THRESHOLD1 * THRESHOLD1)
#define THRESHOLD3 64
#define THRESHOLD4 (THRESHOLD3 * 128)
#define MAX_DICT_SIZE (1 << 19)
#define MAX_WORD_LENGTH 31
#define MIN_BLOCK_SIZE 1024
#define MAX_BLOCK_SIZE (1 << 30)
#define LF 0x0A
#define CR 0x0D
#define ESCAPE_TOKEN1 0x0F
#define ESCAPE_TOKEN2 0x0E
#define HASH1 0x7FEB352D
#define HASH2 ((int32_t)0x846CA68B)
#define MASK_FLIP_CASE 0x80
#define MASK_NOT_TEXT 0x80
#define MASK_CRLF 0x40
#define MASK_XML_HTML 0x20
#define MASK_TEXT_CODEC 0x10
#define MASK_DT 0x0F
#define MASK_LENGTH 0x0007FFFF

typedef struct { int32_t hash; int32_t pos; int32_t data; const uint8_t* buf; } DictEntry;

static int ilog2(uint32_t x) { return 31 - __builtin_clz(x); }
static int32_t sb(uint8_t b) { return (int32_t)(int8_t)b; }                       /* Java byte -> int */
static int32_t mul32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static int isLowerCase(int32_t v) { return (v >= 'a') && (v <= 'z'); }
static int isUpperCase(int32_t v) { return (v >= 'A') && (v <= 'Z'); }
static int isText(uint8_t val) { return isLowerCase((int32_t)(int8_t)(val | 0x20)); }   /* :247-249 */

static int DELIMITER_CHARS[256];
static uint8_t DICT_WORDS[KZO_DICT_EN_1024_LEN];
static DictEntry STATIC_DICTIONARY[1024];
static int STATIC_DICT_WORDS;
static const uint8_t ESC2_BUF[1] = {ESCAPE_TOKEN2}, ESC1_BUF[1] = {ESCAPE_TOKEN1};
static pthread_once_t text_once = PTHREAD_ONCE_INIT;

static void text_init(void) {
  for (int i = 0; i < 256; i++) {                                                   /* :57-85 */
    int r = 0;
    if ((i >= ' ') && (i <= '/')) r = 1;
    else if ((i >= ':') && (i <= '?')) r = 1;
    else switch (i) { case '\n': case '\t': case '\r': case '_': case '|': case '{': case '}': case '[': case ']': r = 1; break; default: r = 0; }
    DELIMITER_CHARS[i] = r;
  }
  memcpy(DICT_WORDS, KZO_DICT_EN_1024, KZO_DICT_EN_1024_LEN);
  /* createDictionary(words, dict, 1024, 0) :215-244 */
  uint8_t* words = DICT_WORDS;
  int anchor = 0, nbWords = 0;
  int32_t h = HASH1;
  for (int i = 0; (i < KZO_DICT_EN_1024_LEN) && (nbWords < 1024); i++) {
    if (!isText(words[i])) continue;
    if (isUpperCase(sb(words[i]))) {
      if (i > anchor) {
        DictEntry e = {h, anchor, ((i - anchor) << 24) | nbWords, words};
        STATIC_DICTIONARY[nbWords] = e;
        nbWords++;
        anchor = i;
        h = HASH1;
      }
      words[i] ^= 0x20;
    }
    h = mul32(h, HASH1) ^ mul32(sb(words[i]), HASH2);
  }
  if (nbWords < 1024) {
    DictEntry e = {h, anchor, ((KZO_DICT_EN_1024_LEN - anchor) << 24) | nbWords, words};
    STATIC_DICTIONARY[nbWords] = e;
    nbWords++;
  }
  STATIC_DICT_WORDS = nbWords;
}

static int sameWords(const uint8_t* buf1, int idx1, const uint8_t* buf2, int idx2, int length) {   /* :469-479 */
  while (length > 0) { length--; if (buf1[idx1 + length] != buf2[idx2 + length]) return 0; }
  return 1;
}

/* ---- detectType :387-466 ---- */
static int detectType(const int* freqs0, int (*freqs)[256], int count) {
  const int dt = kzo_detect_simple_type(count, freqs0);
  if (dt != KZO_DT_UNDEFINED) return MASK_NOT_TEXT | dt;
  int sum = freqs0[0xC0] + freqs0[0xC1];
  for (int i = 0xF5; i <= 0xFF; i++) sum += freqs0[i];
  if (sum != 0) return MASK_NOT_TEXT;
  int sum1 = 0, sum2 = 0;
  for (int i = 0; i < 256; i++) {
    if ((i < 0xA0) || (i > 0xBF)) sum1 += freqs[0xE0][i];
    if ((i < 0x80) || (i > 0x9F)) sum1 += freqs[0xED][i];
    if ((i < 0x90) || (i > 0xBF)) sum1 += freqs[0xF0][i];
    if ((i < 0x80) || (i > 0x8F)) sum1 += freqs[0xF4][i];
    if ((i < 0x80) || (i > 0xBF)) {
      for (int j = 0xC2; j <= 0xDF; j++) sum1 += freqs[j][i];
      for (int j = 0xE1; j <= 0xEC; j++) sum1 += freqs[j][i];
      sum1 += freqs[0xF1][i]; sum1 += freqs[0xF2][i]; sum1 += freqs[0xF3][i];
      sum1 += freqs[0xEE][i]; sum1 += freqs[0xEF][i];
    } else {
      sum2 += freqs0[i];
    }
    if (sum1 != 0) return MASK_NOT_TEXT;
  }
  return sum2 >= (count / 8) ? (MASK_NOT_TEXT | KZO_DT_UTF8) : MASK_NOT_TEXT;
}

/* ---- computeStats :269-384 ---- */
static int computeStats(const uint8_t* block, int start, int end, int* freqs0, int strict) {
  if (!strict && (kzo_magic_type(block + start) != 0 /* Magic.NO_MAGIC */)) return MASK_NOT_TEXT;
  int (*freqs)[256] = (int (*)[256])calloc(256, sizeof(int[256]));
  int prv = 0;
  const int count = end - start;
  for (int i = start; i < end; i++) {                 /* the unrolled loop and its tail count the same pairs */
    const int cur = block[i];
    freqs0[cur]++;
    freqs[prv][cur]++;
    prv = cur;
  }
  int nbTextChars = freqs0[CR] + freqs0[LF];
  int nbASCII = 0;
  for (int i = 0; i < 128; i++) { if (isText((uint8_t)i)) nbTextChars += freqs0[i]; nbASCII += freqs0[i]; }
  const int nbBinChars = count - nbASCII;
  int notText = nbBinChars > (count >> 2);
  if (!notText) {
    notText = nbTextChars < (count / 4);
    if (strict) notText |= ((freqs0[0] >= (count / 100)) || ((nbASCII / 95) < (count / 100)));
    else notText |= (freqs0[32] < (count / 50));
  }
  int res = 0;
  if (notText) { res |= detectType(freqs0, freqs, count); free(freqs); return res; }
  if (nbBinChars <= count - count / 10) {
    const int f1 = freqs0['<'], f2 = freqs0['>'];
    const int f3 = freqs['&']['a'] + freqs['&']['g'] + freqs['&']['l'] + freqs['&']['q'];
    int minFreq = (count - nbBinChars) >> 9; if (minFreq < 2) minFreq = 2;
    if ((f1 >= minFreq) && (f2 >= minFreq) && (f3 > 0)) {
      if (f1 < f2) { if (f1 >= f2 - f2 / 100) res |= MASK_XML_HTML; }
      else if (f2 < f1) { if (f2 >= f1 - f1 / 100) res |= MASK_XML_HTML; }
      else res |= MASK_XML_HTML;
    }
  }
  if ((freqs0[CR] != 0) && (freqs0[CR] == freqs0[LF])) {
    res |= MASK_CRLF;
    for (int i = 0; i < 256; i++) {
      if ((i != LF) && (freqs[CR][i]) != 0) { res &= ~MASK_CRLF; break; }
      if ((i != CR) && (freqs[i][LF]) != 0) { res &= ~MASK_CRLF; break; }
    }
  }
  free(freqs);
  return res;
}

/* ---- codec state (fields of TextCodec1 / TextCodec2) ---- */
typedef struct {
  DictEntry** dictMap; DictEntry** dictList; DictEntry* pool; int poolCap;
  int staticDictSize, logHashSize, hashMask, isCRLF, dictSize, variant;
} TC;

static void tc_init(TC* t, int variant, int blockSize) {
  int log = 13;                                                                     /* :561-575 / :1068-1081 */
  if (variant == 1) { if (blockSize >= 8) { log = ilog2((uint32_t)(blockSize / 8)); if (log > 26) log = 26; if (log < 13) log = 13; } }
  else { if (blockSize >= 32) { log = ilog2((uint32_t)(blockSize / 32)); if (log > 24) log = 24; if (log < 13) log = 13; } }
  memset(t, 0, sizeof(*t));
  t->variant = variant;
  t->logHashSize = log;
  t->dictSize = 1 << 13;
  t->hashMask = (1 << log) - 1;
  t->staticDictSize = (variant == 1) ? STATIC_DICT_WORDS + 2 : STATIC_DICT_WORDS;
}
static void tc_free(TC* t) { free(t->dictMap); free(t->dictList); free(t->pool); }

/* the DictEntry objects a codec instance allocates itself (indexes >= staticDictSize, and TextCodec1's two escape entries) */
static DictEntry* tc_new_entry(TC* t, int idx, const uint8_t* buf, int pos, int32_t hash, int length) {
  DictEntry* e = &t->pool[idx];
  e->buf = buf; e->pos = pos; e->hash = hash; e->data = (length << 24) | idx;
  return e;
}

static void tc_reset(TC* t, int count) {                                            /* :578-613 / :1092-1119 */
  int log = 13;
  if (count >= 1024) { log = ilog2((uint32_t)(count / 128)); if (log > 18) log = 18; if (log < 13) log = 13; }
  t->dictSize = 1 << log;
  t->dictMap = (DictEntry**)calloc((size_t)1 << t->logHashSize, sizeof(DictEntry*));
  t->dictList = (DictEntry**)calloc(MAX_DICT_SIZE, sizeof(DictEntry*));
  t->pool = (DictEntry*)calloc(MAX_DICT_SIZE, sizeof(DictEntry));
  for (int i = 0; i < 1024 && i < t->dictSize && i < STATIC_DICT_WORDS; i++) t->dictList[i] = &STATIC_DICTIONARY[i];
  if (t->variant == 1) {
    t->dictList[STATIC_DICT_WORDS] = tc_new_entry(t, STATIC_DICT_WORDS, ESC2_BUF, 0, 0, 1);
    t->dictList[STATIC_DICT_WORDS + 1] = tc_new_entry(t, STATIC_DICT_WORDS + 1, ESC1_BUF, 0, 0, 1);
  }
  for (int i = 0; i < t->staticDictSize; i++) { DictEntry* e = t->dictList[i]; t->dictMap[e->hash & t->hashMask] = e; }
  for (int i = t->staticDictSize; i < t->dictSize; i++) t->dictList[i] = tc_new_entry(t, i, NULL, -1, 0, 0);
}

static int tc_expand(TC* t) {                                                       /* expandDictionary */
  if (t->dictSize >= MAX_DICT_SIZE) return 0;
  for (int i = t->dictSize; i < t->dictSize * 2; i++) t->dictList[i] = tc_new_entry(t, i, NULL, -1, 0, 0);
  t->dictSize <<= 1;
  return 1;
}

/* ---- TextCodec1 ---- */
static int emitWordIndex1(uint8_t* dst, int dstIdx, int val) {                      /* :850-863 */
  if (val >= THRESHOLD1) {
    if (val >= THRESHOLD2) dst[dstIdx++] = (uint8_t)(0xE0 | (val >> 14));
    dst[dstIdx] = (uint8_t)(0x80 | (val >> 7));
    dst[dstIdx + 1] = (uint8_t)(0x7F & val);
    return dstIdx + 2;
  }
  dst[dstIdx] = (uint8_t)val;
  return dstIdx + 1;
}
static int emitSymbols1(TC* t, const uint8_t* src, int srcIdx, uint8_t* dst, int dstIdx, int srcEnd, int dstEnd) {   /* :806-847 */
  for (int i = srcIdx; i < srcEnd; i++) {
    if (dstIdx >= dstEnd) return dstEnd + 1;
    const uint8_t cur = src[i];
    switch (cur) {
      case ESCAPE_TOKEN1: case ESCAPE_TOKEN2: {
        dst[dstIdx++] = ESCAPE_TOKEN1;
        const int idx = (cur == ESCAPE_TOKEN1) ? t->staticDictSize - 1 : t->staticDictSize - 2;
        int lenIdx = 2;
        if (idx >= THRESHOLD2) lenIdx = 3; else if (idx < THRESHOLD1) lenIdx = 1;
        if (dstIdx + lenIdx >= dstEnd) return dstEnd + 1;
        dstIdx = emitWordIndex1(dst, dstIdx, idx);
        break;
      }
      case CR: if (!t->isCRLF) dst[dstIdx++] = cur; break;
      default: dst[dstIdx++] = cur;
    }
  }
  return dstIdx;
}

/* ---- TextCodec2 ---- */
static int emitWordIndex2(uint8_t* dst, int dstIdx, int wIdx) {                     /* :1370-1394 */
  wIdx++;
  if (wIdx >= THRESHOLD3) {
    if (wIdx >= THRESHOLD4) {
      dst[dstIdx + 0] = (uint8_t)(0xF0 | (wIdx >> 16)); dst[dstIdx + 1] = (uint8_t)(wIdx >> 8); dst[dstIdx + 2] = (uint8_t)wIdx;
      return dstIdx + 3;
    }
    dst[dstIdx] = (uint8_t)(0xC0 | (wIdx >> 8)); dst[dstIdx + 1] = (uint8_t)wIdx;
    return dstIdx + 2;
  }
  dst[dstIdx] = (uint8_t)(0x80 | wIdx);
  return dstIdx + 1;
}
static int emitSymbols2(TC* t, const uint8_t* src, int srcIdx, uint8_t* dst, int dstIdx, int srcEnd, int dstEnd) {   /* :1304-1367 */
  if (dstIdx + 2 * (srcEnd - srcIdx) < dstEnd) {
    for (int i = srcIdx; i < srcEnd; i++) {
      const uint8_t cur = src[i];
      switch (cur) {
        case ESCAPE_TOKEN1: dst[dstIdx++] = ESCAPE_TOKEN1; dst[dstIdx++] = ESCAPE_TOKEN1; break;
        case CR: if (!t->isCRLF) dst[dstIdx++] = cur; break;
        default: dst[dstIdx] = ESCAPE_TOKEN1; dstIdx += (cur >> 7); dst[dstIdx++] = cur;          /* cur >>> 31 of the signed byte */
      }
    }
  } else {
    for (int i = srcIdx; i < srcEnd; i++) {
      const uint8_t cur = src[i];
      switch (cur) {
        case ESCAPE_TOKEN1:
          if (dstIdx >= dstEnd - 1) return dstEnd + 1;
          dst[dstIdx++] = ESCAPE_TOKEN1; dst[dstIdx++] = ESCAPE_TOKEN1;
          break;
        case CR:
          if (!t->isCRLF) { if (dstIdx >= dstEnd) return dstEnd + 1; dst[dstIdx++] = cur; }
          break;
        default:
          if ((cur & 0x80) != 0) { if (dstIdx >= dstEnd) return dstEnd + 1; dst[dstIdx++] = ESCAPE_TOKEN1; }
          if (dstIdx >= dstEnd) return dstEnd + 1;
          dst[dstIdx++] = cur;
      }
    }
  }
  return dstIdx;
}

/* forward of both variants: they differ in strict stats, the end margin (4 / 3), the reference token and emitSymbols */
static int tc_forward(TC* t, int* dataType, const uint8_t* src, int count, uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  if (dstCap < count) return 0;                                                     /* getMaxEncodedLength == srcLength */
  int srcIdx = 0, dstIdx = 0;
  const int srcEnd = count;
  if (dataType) {                                                                    /* :636-645 */
    const int dt = *dataType;
    if ((dt != KZO_DT_UNDEFINED) && (dt != KZO_DT_TEXT) && (dt != KZO_DT_BIN)) return 0;
  }
  int freqs0[256];
  memset(freqs0, 0, sizeof(freqs0));
  const int mode = computeStats(src, srcIdx, srcEnd, freqs0, t->variant == 1);
  if ((mode & MASK_NOT_TEXT) != 0) {                                                 /* :650-665: DataType.values()[t] */
    if (dataType) { const int dt = mode & MASK_DT; if (dt <= KZO_DT_UTF8) *dataType = dt; }
    return 0;
  }
  if (dataType) *dataType = KZO_DT_TEXT;
  tc_reset(t, count);
  const int dstEnd = count;
  const int dstEndM = dstEnd - (t->variant == 1 ? 4 : 3);
  int emitAnchor = 0;
  int words = t->staticDictSize;
  t->isCRLF = (mode & MASK_CRLF) != 0;
  dst[dstIdx++] = (uint8_t)mode;
  int res = 1;
  while ((srcIdx < srcEnd) && (src[srcIdx] == ' ')) { dst[dstIdx++] = ' '; srcIdx++; emitAnchor++; }
  if (srcIdx >= srcEnd) return 0;                     /* a block of spaces: src[srcIdx] below is past the block (cannot pass computeStats anyway) */
  int delimAnchor = isText(src[srcIdx]) ? srcIdx - 1 : srcIdx;
  while (srcIdx < srcEnd) {
    const uint8_t cur = src[srcIdx];
    if (isText(cur)) { srcIdx++; continue; }
    if ((srcIdx > delimAnchor + 2) && DELIMITER_CHARS[cur]) {
      const int length = srcIdx - delimAnchor - 1;
      if (length <= MAX_WORD_LENGTH) {
        const int32_t val = sb(src[delimAnchor + 1]);
        int32_t h1 = mul32(HASH1, HASH1) ^ mul32(val, HASH2);
        int32_t h2 = mul32(HASH1, HASH1) ^ mul32(val ^ 0x20, HASH2);
        for (int i = delimAnchor + 2; i < srcIdx; i++) {
          const int32_t h = mul32(sb(src[i]), HASH2);
          h1 = mul32(h1, HASH1) ^ h;
          h2 = mul32(h2, HASH1) ^ h;
        }
        DictEntry* e = NULL;
        DictEntry* e1 = t->dictMap[h1 & t->hashMask];
        if ((e1 != NULL) && (e1->hash == h1) && (((uint32_t)e1->data >> 24) == (uint32_t)length)) e = e1;
        else {
          DictEntry* e2 = t->dictMap[h2 & t->hashMask];
          if ((e2 != NULL) && (e2->hash == h2) && (((uint32_t)e2->data >> 24) == (uint32_t)length)) e = e2;
        }
        if (e != NULL) { if (!sameWords(src, delimAnchor + 2, e->buf, e->pos + 1, length - 1)) e = NULL; }
        if (e == NULL) {
          if (((length > 3) || ((length == 3) && (words < THRESHOLD2))) && (e1 == NULL)) {
            e = t->dictList[words];
            if ((e->data & MASK_LENGTH) >= t->staticDictSize) {
              t->dictMap[e->hash & t->hashMask] = NULL;
              e->buf = src; e->pos = delimAnchor + 1; e->hash = h1; e->data = (length << 24) | words;
            }
            t->dictMap[h1 & t->hashMask] = e;
            words++;
            if (words >= t->dictSize) { if (!tc_expand(t)) words = t->staticDictSize; }
          }
        } else {
          if ((emitAnchor != delimAnchor) || (src[delimAnchor] != ' ')) {
            dstIdx = (t->variant == 1) ? emitSymbols1(t, src, emitAnchor, dst, dstIdx, delimAnchor + 1, dstEnd)
                                       : emitSymbols2(t, src, emitAnchor, dst, dstIdx, delimAnchor + 1, dstEnd);
          }
          if (dstIdx >= dstEndM) { res = 0; break; }
          if (t->variant == 1) {
            dst[dstIdx++] = (e == e1) ? ESCAPE_TOKEN1 : ESCAPE_TOKEN2;
            dstIdx = emitWordIndex1(dst, dstIdx, e->data & MASK_LENGTH);
          } else {
            dst[dstIdx] = MASK_FLIP_CASE;
            dstIdx += (e == e1 ? 0 : 1);
            dstIdx = emitWordIndex2(dst, dstIdx, e->data & MASK_LENGTH);
          }
          emitAnchor = delimAnchor + 1 + (int)((uint32_t)e->data >> 24);
        }
      }
    }
    delimAnchor = srcIdx;
    srcIdx++;
  }
  if (res) {
    const int dIdx = (t->variant == 1) ? emitSymbols1(t, src, emitAnchor, dst, dstIdx, srcEnd, dstEnd)
                                       : emitSymbols2(t, src, emitAnchor, dst, dstIdx, srcEnd, dstEnd);
    if (dIdx > dstEnd) res = 0; else dstIdx = dIdx;
    res &= (srcIdx == srcEnd);
  }
  *produced = dstIdx;
  return res;
}

/* inverse of both variants (:866-1028, :1397-1603); dstEnd = dst.length = the capacity handed in */
static int tc_inverse(TC* t, const uint8_t* src, int count, uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  tc_reset(t, dstCap);                                /* reset(output.length) */
  int srcIdx = 0, dstIdx = 0;
  const int isCRLF = (src[srcIdx++] & MASK_CRLF) != 0;
  t->isCRLF = isCRLF;
  const int srcEnd = count, dstEnd = dstCap;
  if (srcIdx >= srcEnd) { return srcIdx == srcEnd; }  /* a 1-byte block: the loop does not run */
  int delimAnchor = isText(src[srcIdx]) ? srcIdx - 1 : srcIdx;
  int words = t->staticDictSize;
  int wordRun = 0, res = 1;
#define NEED(k) if (srcIdx + (k) > srcEnd) { res = 0; break; }
  while ((srcIdx < srcEnd) && (dstIdx < dstEnd)) {
    uint8_t cur = src[srcIdx];
    if (isText(cur)) { dst[dstIdx] = cur; srcIdx++; dstIdx++; continue; }
    if ((srcIdx > delimAnchor + 3) && DELIMITER_CHARS[cur]) {
      const int length = srcIdx - delimAnchor - 1;
      if (length <= MAX_WORD_LENGTH) {
        int32_t h1 = HASH1;
        for (int i = delimAnchor + 1; i < srcIdx; i++) h1 = mul32(h1, HASH1) ^ mul32(sb(src[i]), HASH2);
        DictEntry* e = NULL;
        DictEntry* e1 = t->dictMap[h1 & t->hashMask];
        if ((e1 != NULL) && (e1->hash == h1) && (((uint32_t)e1->data >> 24) == (uint32_t)length)) {
          if (sameWords(src, delimAnchor + 2, e1->buf, e1->pos + 1, length - 1)) e = e1;
        }
        if (e == NULL) {
          if (((length > 3) || (words < THRESHOLD2)) && (e1 == NULL)) {
            e = t->dictList[words];
            if ((e->data & MASK_LENGTH) >= t->staticDictSize) {
              t->dictMap[e->hash & t->hashMask] = NULL;
              e->buf = src; e->pos = delimAnchor + 1; e->hash = h1; e->data = (length << 24) | words;
            }
            t->dictMap[h1 & t->hashMask] = e;
            words++;
            if (words >= t->dictSize) { if (!tc_expand(t)) words = t->staticDictSize; }
          }
        }
      }
    }
    srcIdx++;
    int isRef, flip = 0, idx = 0;
    if (t->variant == 1) isRef = (cur == ESCAPE_TOKEN1) || (cur == ESCAPE_TOKEN2);
    else isRef = (cur & 0x80) != 0;
    if (isRef) {
      if (t->variant == 1) {                                                        /* :945-961 */
        NEED(1)
        idx = src[srcIdx++];
        if (idx >= 128) {
          idx &= 0x7F;
          NEED(1)
          int idx2 = sb(src[srcIdx++]);
          if ((idx2 & 0x80) != 0) { idx = ((idx & 0x1F) << 7) | (idx2 & 0x7F); NEED(1) idx2 = src[srcIdx++] & 0x7F; }
          idx = (idx << 7) | idx2;
          if (idx >= t->dictSize) { res = 0; break; }
        }
        flip = (cur == ESCAPE_TOKEN2) ? 0x20 : 0;
      } else {                                                                      /* :1503-1537 (bsVersion >= 6) */
        if (cur == MASK_FLIP_CASE) { flip = 0x20; NEED(1) cur = src[srcIdx++]; }
        idx = cur & 0x7F;
        if (idx >= 64) {
          if (idx >= 112) { NEED(2) idx = ((idx & 0x0F) << 16) | (src[srcIdx] << 8) | src[srcIdx + 1]; srcIdx += 2; }
          else { NEED(1) idx = ((idx & 0x1F) << 8) | src[srcIdx]; srcIdx++; }
          if (idx > t->dictSize) { res = 0; break; }
        } else if (idx == 0) { res = 0; break; }
        idx--;
      }
      if (idx < 0 || idx >= t->dictSize || t->dictList[idx] == NULL) { res = 0; break; }   /* Java: array bound / null entry -> exception */
      const DictEntry* e = t->dictList[idx];
      const int length = (int)(((uint32_t)e->data >> 24) & 0xFF);
      const uint8_t* buf = e->buf;
      if (wordRun && (length > 1)) { if (dstIdx >= dstEnd) { res = 0; break; } dst[dstIdx++] = ' '; }
      if ((e->pos < 0) || (dstIdx + length >= dstEnd)) { res = 0; break; }
      dst[dstIdx++] = (uint8_t)(buf[e->pos] ^ flip);
      if (length > 1) {
        for (int n = e->pos + 1, l = e->pos + length; n < l; n++, dstIdx++) dst[dstIdx] = buf[n];
        wordRun = 1;
        delimAnchor = srcIdx;
      } else {
        wordRun = 0;
        delimAnchor = srcIdx - 1;
      }
    } else {
      if ((t->variant == 2) && (cur == ESCAPE_TOKEN1)) {
        NEED(1)
        dst[dstIdx++] = src[srcIdx++];
      } else {
        if (isCRLF && (cur == LF)) { dst[dstIdx++] = CR; if (dstIdx >= dstEnd) { res = 0; break; } }
        dst[dstIdx++] = cur;
      }
      wordRun = 0;
      delimAnchor = srcIdx - 1;
    }
  }
#undef NEED
  *produced = dstIdx;
  return res && (srcIdx == srcEnd);
}

/* ---- TextCodec.forward / inverse :482-531.  codecType: 1 or 2 (the context's "textcodec"); blockSize: the context's "blockSize" ---- */
int kzo_text_forward(int codecType, int blockSize, int* dataType, const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  if (n == 0) return 1;
  if ((n < MIN_BLOCK_SIZE) || (n > MAX_BLOCK_SIZE)) return 0;
  pthread_once(&text_once, text_init);
  TC t;
  tc_init(&t, codecType == 1 ? 1 : 2, blockSize);
  const int res = tc_forward(&t, dataType, src, n, dst, dstCap, produced);
  tc_free(&t);
  if (res) { if (codecType == 1) dst[0] &= (uint8_t)~MASK_TEXT_CODEC; else dst[0] |= MASK_TEXT_CODEC; }   /* bsVersion 7 */
  return res;
}

int kzo_text_inverse(int blockSize, const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  if (n == 0) return 1;
  if (n > MAX_BLOCK_SIZE) return 0;
  pthread_once(&text_once, text_init);
  const int encodingType = ((src[0] & MASK_TEXT_CODEC) == 0) ? 1 : 2;
  TC t;
  tc_init(&t, encodingType, blockSize);
  const int res = tc_inverse(&t, src, n, dst, dstCap, produced);
  tc_free(&t);
  return res;
}

int kzo_text_static_dict_words(void) { pthread_once(&text_once, text_init); return STATIC_DICT_WORDS; }
