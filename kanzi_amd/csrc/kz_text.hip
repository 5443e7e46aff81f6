// kz_text.hip -- the two CPU pre-transforms in front of the GPU chains of the reference's levels 3, 5 and 6, host C++:
//   TEXT = K/transform/TextCodec.java (dictionary word replacement; TextCodec1 for FPAQ/CM/TPAQ streams, TextCodec2 for
//          NONE/ANS0/HUFFMAN/RANGE: TransformFactory.java:275-286; the variant is bit 0x10 of the first coded byte, :496-528)
//   UTF  = K/transform/UTFCodec.java (UTF-8 code points -> one- or two-byte aliases ranked by frequency)
// Both are branchy sequential dictionary coders whose output depends on the exact order of dictionary updates; they run per
// block on host threads in front of (forward) or behind (inverse) the batched GPU stages (kz_api.hip).
//
// Data structures differ from the reference's object graph: the dictionary is a table of plain records indexed by word
// number, the hash map holds word numbers (-1 = empty) instead of references, and one arena per thread is reused from
// block to block.  What must not differ is the sequence of decisions, so every rule below cites its line.
#include "kz_internal.h"
#include "kz_magic.h"
#include "kz_text_dict.h"
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <algorithm>
#include <memory>
#include <mutex>
#include <vector>

typedef uint8_t u8;
typedef uint32_t u32;

namespace {

constexpr int kThreshold1 = 128, kThreshold2 = 128 * 128, kThreshold3 = 64, kThreshold4 = 64 * 128;   // TextCodec.java:32-35
constexpr int kMaxDict = 1 << 19, kMaxWord = 31, kMinBlock = 1024, kMaxBlock = 1 << 30;             // :36-39
constexpr u8 kLF = 0x0A, kCR = 0x0D, kEsc1 = 0x0F, kEsc2 = 0x0E;                                     // :41-44
constexpr u32 kHash1 = 0x7FEB352Du, kHash2 = 0x846CA68Bu;                                            // :45-46
constexpr int kNotText = 0x80, kCRLF = 0x40, kXml = 0x20, kCodecBit = 0x10, kDtMask = 0x0F;        // :48-52
constexpr u32 kIdxMask = 0x0007FFFF;                                                                 // :53

inline bool is_text(u8 c) { const u8 l = c | 0x20; return l >= 'a' && l <= 'z'; }                    // :247-249 (bytes >= 0x80 never are)

// ---- 32 bytes at a time (AVX2; the scalar loops below them are the reference form and the fallback) ----
// The reference walks a block byte by byte (:694-790); three quarters of the bytes of a text are letters inside words, at which
// nothing happens.  The host CPUs are what the level-exact chains wait for, so the walks skip them with one compare per
// 32 bytes: a bit mask of the letters, and only the non-letters are visited.
#if defined(__x86_64__)
#define KZ_TEXT_AVX2 1
__attribute__((target("avx2"))) inline u32 letters32(const u8* p) {                                  // bit k: p[k] is a letter
  const __m256i v = _mm256_loadu_si256((const __m256i*)p);
  const __m256i t = _mm256_sub_epi8(_mm256_or_si256(v, _mm256_set1_epi8(0x20)), _mm256_set1_epi8('a'));
  return (u32)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_max_epu8(t, _mm256_set1_epi8(25)), _mm256_set1_epi8(25)));
}
__attribute__((target("avx2"))) inline u32 eq32(const u8* p, char c) {
  return (u32)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)p), _mm256_set1_epi8(c)));
}
__attribute__((target("avx2"))) inline u32 high32(const u8* p) { return (u32)_mm256_movemask_epi8(_mm256_loadu_si256((const __m256i*)p)); }   // bit k: p[k] >= 0x80
inline bool have_avx2() { static const bool v = __builtin_cpu_supports("avx2"); return v; }
#else
#define KZ_TEXT_AVX2 0
inline bool have_avx2() { return false; }
#endif
inline u32 hash_step(u32 h, u8 c) { return h * kHash1 ^ (u32)(int32_t)(int8_t)c * kHash2; }          // :234: Java byte is signed

struct Word { u32 hash; int32_t pos; u32 lenIdx; const u8* text; };       // lenIdx = length << 24 | word number (DictEntry.data)

struct StaticDict {
  bool delim[256];
  u8 text[KZ_DICT_EN_1024_LEN];
  Word words[1024];
  int count = 0;
  StaticDict() {
    for (int c = 0; c < 256; c++)                                                                    // :57-85
      delim[c] = (c >= ' ' && c <= '/') || (c >= ':' && c <= '?') || c == '\n' || c == '\t' || c == '\r' || c == '_' || c == '|' ||
                 c == '{' || c == '}' || c == '[' || c == ']';
    memcpy(text, KZ_DICT_EN_1024, KZ_DICT_EN_1024_LEN);
    // a word starts at every capital; the stored text is lower case (createDictionary :215-244)
    int anchor = 0;
    u32 h = kHash1;
    for (int i = 0; i < KZ_DICT_EN_1024_LEN && count < 1024; i++) {
      if (text[i] >= 'A' && text[i] <= 'Z') {
        if (i > anchor) { words[count] = Word{h, anchor, ((u32)(i - anchor) << 24) | (u32)count, text}; count++; anchor = i; h = kHash1; }
        text[i] ^= 0x20;
      }
      h = hash_step(h, text[i]);
    }
    if (count < 1024) { words[count] = Word{h, anchor, ((u32)(KZ_DICT_EN_1024_LEN - anchor) << 24) | (u32)count, text}; count++; }
  }
};
const StaticDict& static_dict() { static const StaticDict d; return d; }

// ---- block statistics: is this text, and if not, what is it (computeStats :269-384, detectType :387-466) ----
struct PairCounts {
  std::unique_ptr<int32_t[]> c{new int32_t[65536]};
  int32_t* row(int a) { return c.get() + a * 256; }
};

// Global.detectSimpleType (K/Global.java:556-605) over an order-0 histogram
int detect_simple_type(int count, const int32_t* f) {
  if (count == 0) return KZ_DT_UNDEFINED;
  int64_t dna = 0, num = 0, b64 = 0;
  int distinct = 0;
  for (int c = 0; c < 256; c++) {
    const int32_t v = f[c];
    if (v == 0) continue;
    distinct++;
    const bool digit = c >= '0' && c <= '9';
    if (c == 'a' || c == 'c' || c == 'g' || c == 'n' || c == 't' || c == 'u' || c == 'A' || c == 'C' || c == 'G' || c == 'N' || c == 'T' || c == 'U') dna += v;
    if (digit || c == '+' || c == '-' || c == '*' || c == '/' || c == '=' || c == ',' || c == '.' || c == ':' || c == ';' || c == ' ') num += v;
    if (digit || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '+' || c == '/') b64 += v;
  }
  if (f['='] == 1) b64++;                                        // trailing padding
  if (dna > count - count / 12) return KZ_DT_DNA;
  if (num == count) return KZ_DT_NUMERIC;
  if (b64 == count) return KZ_DT_BASE64;
  if (distinct == 256) return KZ_DT_BIN;
  if (distinct <= 4) return KZ_DT_SMALL_ALPHABET;
  return KZ_DT_UNDEFINED;
}

// the UTF-8 plausibility rules shared by TextCodec.detectType and UTFCodec.validate (Unicode table 3.7): false when a lead
// byte is followed by a byte outside its allowed range anywhere in the pair histogram
bool utf8_pairs_ok(const int32_t* f0, PairCounts& pc, int64_t* continuation) {
  int64_t bad = f0[0xC0] + f0[0xC1];
  for (int c = 0xF5; c <= 0xFF; c++) bad += f0[c];
  if (bad) return false;
  int64_t cont = 0;
  for (int i = 0; i < 256; i++) {
    int64_t s = 0;
    if (i < 0xA0 || i > 0xBF) s += pc.row(0xE0)[i];
    if (i < 0x80 || i > 0x9F) s += pc.row(0xED)[i];
    if (i < 0x90 || i > 0xBF) s += pc.row(0xF0)[i];
    if (i < 0x80 || i > 0x8F) s += pc.row(0xF4)[i];
    if (i < 0x80 || i > 0xBF) {
      for (int j = 0xC2; j <= 0xDF; j++) s += pc.row(j)[i];
      for (int j = 0xE1; j <= 0xEC; j++) s += pc.row(j)[i];
      s += pc.row(0xF1)[i] + pc.row(0xF2)[i] + pc.row(0xF3)[i] + pc.row(0xEE)[i] + pc.row(0xEF)[i];
    } else cont += f0[i];
    if (s) return false;
  }
  *continuation = cont;
  return true;
}

// pairs (previous byte, byte) the text path asks about; the first byte's previous byte is 0 (as in the reference's pair table)
#define KZ_PAIR(A, B) { amp += ((A) == '&') & (((B) == 'a') | ((B) == 'g') | ((B) == 'l') | ((B) == 'q')); \
                        crx += ((A) == kCR) & ((B) != kLF); xlf += ((B) == kLF) & ((A) != kCR); }
#if KZ_TEXT_AVX2
__attribute__((target("avx2"))) int pair_counts_avx2(const u8* p, int n, int64_t& amp, int64_t& crx, int64_t& xlf) {
  int i = 1;
  const __m256i kAmp = _mm256_set1_epi8('&'), kA = _mm256_set1_epi8('a'), kG = _mm256_set1_epi8('g'), kL = _mm256_set1_epi8('l'),
                kQ = _mm256_set1_epi8('q'), vCR = _mm256_set1_epi8((char)kCR), vLF = _mm256_set1_epi8((char)kLF);
  for (; i + 32 <= n; i += 32) {
    const __m256i prev = _mm256_loadu_si256((const __m256i*)(p + i - 1)), cur = _mm256_loadu_si256((const __m256i*)(p + i));
    const __m256i aglq = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(cur, kA), _mm256_cmpeq_epi8(cur, kG)),
                                         _mm256_or_si256(_mm256_cmpeq_epi8(cur, kL), _mm256_cmpeq_epi8(cur, kQ)));
    const __m256i pcr = _mm256_cmpeq_epi8(prev, vCR), clf = _mm256_cmpeq_epi8(cur, vLF);
    amp += __builtin_popcount((u32)_mm256_movemask_epi8(_mm256_and_si256(_mm256_cmpeq_epi8(prev, kAmp), aglq)));
    crx += __builtin_popcount((u32)_mm256_movemask_epi8(_mm256_andnot_si256(clf, pcr)));
    xlf += __builtin_popcount((u32)_mm256_movemask_epi8(_mm256_andnot_si256(pcr, clf)));
  }
  return i;
}
#endif
void pair_counts(const u8* p, int n, int64_t* pamp, int64_t* pcrx, int64_t* pxlf) {
  int64_t amp = 0, crx = 0, xlf = 0;
  if (n > 0) KZ_PAIR(0, p[0])
  int i = 1;
#if KZ_TEXT_AVX2
  if (have_avx2() && n >= 64) i = pair_counts_avx2(p, n, amp, crx, xlf);
#endif
  for (; i < n; i++) KZ_PAIR(p[i - 1], p[i])
  *pamp = amp; *pcrx = crx; *pxlf = xlf;
}
#undef KZ_PAIR

int text_block_mode(const u8* p, int count, bool strict) {
  if (!strict && mm_magic_type(p) != 0) return kNotText;                                             // :272-273
  // One pass: the order-0 histogram (four interleaved tables: no store-to-load chain on runs) and the handful of pair counts the
  // text path reads: "&" followed by a / g / l / q (:343-345), CR followed by something else than LF, LF behind something else than
  // CR (:358-372).  The reference fills a whole 256 x 256 pair table per block (computeStats :283-308); that table is built here
  // only for blocks that turn out not to be text, where detectType's UTF-8 rules need it.  Like the reference's, the pair
  // statistics start from a previous byte of 0.
  int32_t h4[4][256];
  memset(h4, 0, sizeof(h4));
  int64_t amp = 0, crOther = 0, otherLf = 0;
  {
    int i = 0;
    for (; i + 4 <= count; i += 4) { h4[0][p[i]]++; h4[1][p[i + 1]]++; h4[2][p[i + 2]]++; h4[3][p[i + 3]]++; }
    for (; i < count; i++) h4[0][p[i]]++;
  }
  pair_counts(p, count, &amp, &crOther, &otherLf);
  int32_t f0[256];
  for (int c = 0; c < 256; c++) f0[c] = h4[0][c] + h4[1][c] + h4[2][c] + h4[3][c];
  int64_t letters = f0[kCR] + f0[kLF], ascii = 0;
  for (int c = 0; c < 128; c++) { if (is_text((u8)c)) letters += f0[c]; ascii += f0[c]; }
  const int64_t bin = count - ascii;
  bool notText = bin > (count >> 2);
  if (!notText) {
    notText = letters < count / 4;
    if (strict) notText = notText || f0[0] >= count / 100 || ascii / 95 < count / 100;              // :323-324
    else notText = notText || f0[32] < count / 50;                                                   // :326
  }
  if (notText) {                                                                                     // detectType
    const int dt = detect_simple_type(count, f0);
    if (dt != KZ_DT_UNDEFINED) return kNotText | dt;
    PairCounts pc;                                                  // not text: now the pair table (previous byte of the first one: 0)
    memset(pc.c.get(), 0, 65536 * sizeof(int32_t));
    { int prev = 0; for (int i = 0; i < count; i++) { const int c = p[i]; pc.row(prev)[c]++; prev = c; } }
    int64_t cont = 0;
    if (!utf8_pairs_ok(f0, pc, &cont)) return kNotText;
    return cont >= count / 8 ? (kNotText | KZ_DT_UTF8) : kNotText;
  }
  int mode = 0;
  if (bin <= count - count / 10) {                                                                   // :336-356 looks like XML / HTML
    const int lt = f0['<'], gt = f0['>'];
    const int minFreq = std::max((int)((count - bin) >> 9), 2);
    if (lt >= minFreq && gt >= minFreq && amp > 0) {
      const int lo = std::min(lt, gt), hi = std::max(lt, gt);
      if (lo == hi || lo >= hi - hi / 100) mode |= kXml;
    }
  }
  if (f0[kCR] != 0 && f0[kCR] == f0[kLF]) {                                                          // :358-372 every CR is followed by LF and vice versa
    if (crOther == 0 && otherLf == 0) mode |= kCRLF;
  }
  return mode;
}

// ---- the word dictionary of one block ----
class Dictionary {
 public:
  Dictionary(int variant, int ctxBlockSize, int firstCount) : variant_(variant) {
    // hash map size from the context's "blockSize" (:561-575 / :1068-1081), list size from the block (:578-582)
    int log = 13;
    if (variant == 1) { if (ctxBlockSize >= 8) log = std::max(std::min(31 - __builtin_clz((u32)(ctxBlockSize / 8)), 26), 13); }
    else if (ctxBlockSize >= 32) log = std::max(std::min(31 - __builtin_clz((u32)(ctxBlockSize / 32)), 24), 13);
    mask_ = (1u << log) - 1;
    slots_.assign((size_t)1 << log, -1);
    const StaticDict& sd = static_dict();
    fixed_ = sd.count + (variant == 1 ? 2 : 0);
    int llog = 13;
    if (firstCount >= 1024) llog = std::max(std::min(31 - __builtin_clz((u32)(firstCount / 128)), 18), 13);
    size_ = 1 << llog;
    words_.resize(size_);
    for (int i = 0; i < sd.count; i++) words_[i] = sd.words[i];
    if (variant == 1) {                                          // the two escape bytes as one-letter words (:600-603)
      static const u8 e2[1] = {kEsc2}, e1[1] = {kEsc1};
      words_[sd.count] = Word{0, 0, (1u << 24) | (u32)sd.count, e2};
      words_[sd.count + 1] = Word{0, 0, (1u << 24) | (u32)(sd.count + 1), e1};
    }
    for (int i = 0; i < fixed_; i++) slots_[words_[i].hash & mask_] = i;
    for (int i = fixed_; i < size_; i++) words_[i] = Word{0, -1, (u32)i, nullptr};
    next_ = fixed_;
  }
  int fixed() const { return fixed_; }
  int size() const { return size_; }
  int next() const { return next_; }
  const Word& word(int i) const { return words_[i]; }
  int slot(u32 h) const { return slots_[h & mask_]; }
  // a word of `length` letters at text + pos with hash h, not found and its slot empty: it takes the next word number
  // (forward :725-747, inverse :906-928): the number's previous owner leaves the map, a full list doubles up to 2^19 and then
  // the numbering restarts behind the fixed part
  void learn(u32 h, const u8* text, int pos, int length) {
    Word& w = words_[next_];
    if ((int)(w.lenIdx & kIdxMask) >= fixed_) {
      slots_[w.hash & mask_] = -1;
      w = Word{h, pos, ((u32)length << 24) | (u32)next_, text};
    }
    slots_[h & mask_] = next_;
    next_++;
    if (next_ >= size_) {
      if (size_ >= kMaxDict) next_ = fixed_;
      else { words_.resize((size_t)size_ * 2); for (int i = size_; i < size_ * 2; i++) words_[i] = Word{0, -1, (u32)i, nullptr}; size_ *= 2; }
    }
  }

 private:
  int variant_, fixed_, size_, next_;
  u32 mask_;
  std::vector<int32_t> slots_;
  std::vector<Word> words_;
};

inline bool same_tail(const u8* a, const u8* b, int n) { return memcmp(a, b, (size_t)n) == 0; }    // sameWords :469-479

// ---- output of plain bytes between two word references ----
struct Emitter {
  int variant; bool crlf; int fixed;
  // returns the new write position, or end + 1 when the output ran out (:806-847 / :1304-1367)
  int plain(const u8* src, int from, int to, u8* dst, int at, int end) const {
    if (variant == 1) {
      for (int i = from; i < to; i++) {
        if (at >= end) return end + 1;
        const u8 c = src[i];
        if (c == kEsc1 || c == kEsc2) {                         // a literal escape byte becomes a reference to its one-letter word
          dst[at++] = kEsc1;
          const int idx = (c == kEsc1) ? fixed - 1 : fixed - 2;
          const int need = idx >= kThreshold2 ? 3 : (idx < kThreshold1 ? 1 : 2);
          if (at + need >= end) return end + 1;
          at = index1(dst, at, idx);
        } else if (c == kCR) { if (!crlf) dst[at++] = c; }
        else dst[at++] = c;
      }
      return at;
    }
    const bool roomy = at + 2 * (to - from) < end;              // the reference's unchecked fast loop writes the same bytes
    for (int i = from; i < to; i++) {
      const u8 c = src[i];
      if (c == kEsc1) {
        if (!roomy && at >= end - 1) return end + 1;
        dst[at++] = kEsc1; dst[at++] = kEsc1;
      } else if (c == kCR) {
        if (!crlf) { if (!roomy && at >= end) return end + 1; dst[at++] = c; }
      } else {
        if (c & 0x80) { if (!roomy && at >= end) return end + 1; dst[at++] = kEsc1; }
        if (!roomy && at >= end) return end + 1;
        dst[at++] = c;
      }
    }
    return at;
  }
  static int index1(u8* dst, int at, int v) {                                                        // :850-863
    if (v >= kThreshold1) {
      if (v >= kThreshold2) dst[at++] = (u8)(0xE0 | (v >> 14));
      dst[at] = (u8)(0x80 | (v >> 7)); dst[at + 1] = (u8)(v & 0x7F);
      return at + 2;
    }
    dst[at] = (u8)v;
    return at + 1;
  }
  static int index2(u8* dst, int at, int v) {                                                        // :1370-1394 (0x80 alone = case flip)
    v++;
    if (v >= kThreshold4) { dst[at] = (u8)(0xF0 | (v >> 16)); dst[at + 1] = (u8)(v >> 8); dst[at + 2] = (u8)v; return at + 3; }
    if (v >= kThreshold3) { dst[at] = (u8)(0xC0 | (v >> 8)); dst[at + 1] = (u8)v; return at + 2; }
    dst[at] = (u8)(0x80 | v);
    return at + 1;
  }
};

int text_forward(int variant, int ctxBlockSize, int* dataType, const u8* src, int n, u8* dst, int dstCap, int* produced) {
  *produced = 0;
  if (dstCap < n) return 0;                                                                          // getMaxEncodedLength = n
  if (dataType) {                                                                                    // :636-645
    const int dt = *dataType;
    if (dt != KZ_DT_UNDEFINED && dt != KZ_DT_TEXT && dt != KZ_DT_BIN) return 0;
  }
  const int mode = text_block_mode(src, n, variant == 1);
  if (mode & kNotText) { if (dataType) *dataType = mode & kDtMask; return 0; }                       // :650-665
  if (dataType) *dataType = KZ_DT_TEXT;
  Dictionary dict(variant, ctxBlockSize, n);
  const StaticDict& sd = static_dict();
  const Emitter em{variant, (mode & kCRLF) != 0, dict.fixed()};
  const int end = n, margin = end - (variant == 1 ? 4 : 3);
  int at = 0, i = 0, pending = 0;                  // pending: first source byte not yet written
  dst[at++] = (u8)mode;
  while (i < n && src[i] == ' ') { dst[at++] = ' '; i++; pending++; }                                // :688-692
  if (i >= n) return 0;
  int last = is_text(src[i]) ? i - 1 : i;          // position of the previous non-letter
  bool ok = true;
  // what happens at a non-letter at position i (:704-790); false: the output ran out
  auto visit = [&](const int i) -> bool {
    const u8 c = src[i];
    if (i > last + 2 && sd.delim[c]) {             // a run of at least two letters closed by a delimiter (:704)
      const int len = i - last - 1;
      if (len <= kMaxWord) {
        const u8* w = src + last + 1;
        u32 h1 = kHash1 * kHash1 ^ (u32)(int32_t)(int8_t)w[0] * kHash2;
        for (int k = 1; k < len; k++) h1 = hash_step(h1, w[k]);
        const int s1 = dict.slot(h1);
        int found = -1;
        bool flipped = false;
        if (s1 >= 0 && dict.word(s1).hash == h1 && (int)(dict.word(s1).lenIdx >> 24) == len) found = s1;
        else {                                     // the hash with the first letter's case flipped is only needed now (:712-718)
          u32 h2 = kHash1 * kHash1 ^ (u32)(int32_t)(int8_t)(w[0] ^ 0x20) * kHash2;
          for (int k = 1; k < len; k++) h2 = hash_step(h2, w[k]);
          const int s2 = dict.slot(h2);
          if (s2 >= 0 && dict.word(s2).hash == h2 && (int)(dict.word(s2).lenIdx >> 24) == len) { found = s2; flipped = (s2 != s1); }
        }
        if (found >= 0) {                          // hash collision check on everything but the first letter (:720-723)
          const Word& e = dict.word(found);
          if (!same_tail(w + 1, e.text + e.pos + 1, len - 1)) found = -1;
        }
        if (found < 0) {
          if ((len > 3 || (len == 3 && dict.next() < kThreshold2)) && s1 < 0) dict.learn(h1, src, last + 1, len);
        } else {
          // a single space between two references is implied (:752-754)
          if (pending != last || src[last] != ' ') at = em.plain(src, pending, last + 1, dst, at, end);
          if (at >= margin) { ok = false; return false; }
          const int number = (int)(dict.word(found).lenIdx & kIdxMask);
          if (variant == 1) { dst[at++] = flipped ? kEsc2 : kEsc1; at = Emitter::index1(dst, at, number); }
          else { if (flipped) dst[at++] = 0x80; at = Emitter::index2(dst, at, number); }
          pending = last + 1 + len;
        }
      }
    }
    last = i;
    return true;
  };
#if KZ_TEXT_AVX2
  if (have_avx2()) {                               // only the non-letters are visited: one mask per 32 bytes
    while (ok && i + 32 <= n) {
      u32 nl = ~letters32(src + i);
      while (nl) {
        const int j = __builtin_ctz(nl);
        nl &= nl - 1;
        if (!visit(i + j)) break;
      }
      i += 32;
    }
  }
#endif
  for (; ok && i < n; i++) {                        // the reference's walk (and the last bytes of the block)
    if (is_text(src[i])) continue;
    if (!visit(i)) break;
  }
  if (ok) {
    const int e = em.plain(src, pending, n, dst, at, end);
    if (e > end) ok = false; else at = e;
  }
  *produced = at;
  if (ok) { if (variant == 1) dst[0] &= (u8)~kCodecBit; else dst[0] |= kCodecBit; }                   // :496-501
  return ok ? 1 : 0;
}

int text_inverse(int ctxBlockSize, const u8* src, int n, u8* dst, int dstCap, int* produced) {
  *produced = 0;
  const int variant = (src[0] & kCodecBit) ? 2 : 1;                                                  // :525-528
  Dictionary dict(variant, ctxBlockSize, dstCap);                                                     // reset(output.length)
  const StaticDict& sd = static_dict();
  const bool crlf = (src[0] & kCRLF) != 0;
  int i = 1, at = 0;
  if (i >= n) return 1;
  int last = is_text(src[i]) ? i - 1 : i;
  bool afterWord = false, ok = true;
  const int end = dstCap;
#if KZ_TEXT_AVX2
  const bool avx2 = have_avx2();
#endif
  while (i < n && at < end) {
    u8 c = src[i];
    if (is_text(c)) {
#if KZ_TEXT_AVX2
      if (avx2 && i + 32 <= n && at + 32 <= end) {                  // a run of letters is copied in one piece (they are plain bytes: :880-882)
        const u32 m = letters32(src + i);
        const int run = (m == 0xFFFFFFFFu) ? 32 : __builtin_ctz(~m);
        memcpy(dst + at, src + i, 32);                             // (the bytes behind the run are overwritten by what follows)
        at += run; i += run;
        continue;
      }
#endif
      dst[at++] = c; i++; continue;
    }
    if (i > last + 3 && sd.delim[c]) {             // the decoder learns only words of at least three letters (:891)
      const int len = i - last - 1;
      if (len <= kMaxWord) {
        u32 h = kHash1;
        for (int k = last + 1; k < i; k++) h = hash_step(h, src[k]);
        const int s1 = dict.slot(h);
        bool known = false;
        if (s1 >= 0 && dict.word(s1).hash == h && (int)(dict.word(s1).lenIdx >> 24) == len) {
          const Word& e = dict.word(s1);
          known = same_tail(src + last + 2, e.text + e.pos + 1, len - 1);
        }
        if (!known && (len > 3 || dict.next() < kThreshold2) && s1 < 0) dict.learn(h, src, last + 1, len);
      }
    }
    i++;
    const bool ref = (variant == 1) ? (c == kEsc1 || c == kEsc2) : (c & 0x80) != 0;
    if (!ref) {
      if (variant == 2 && c == kEsc1) {            // escaped byte >= 0x80 or a literal 0x0F (:1577-1578)
        if (i >= n) { ok = false; break; }
        dst[at++] = src[i++];
      } else {
        if (crlf && c == kLF) { dst[at++] = kCR; if (at >= end) { ok = false; break; } }
        dst[at++] = c;
      }
      afterWord = false;
      last = i - 1;
      continue;
    }
    int number;
    u8 flip = 0;
    if (variant == 1) {                                                                              // :945-961
      if (i >= n) { ok = false; break; }
      number = src[i++];
      if (number >= 128) {
        number &= 0x7F;
        if (i >= n) { ok = false; break; }
        int b2 = (int8_t)src[i++];
        if (b2 & 0x80) {
          number = ((number & 0x1F) << 7) | (b2 & 0x7F);
          if (i >= n) { ok = false; break; }
          b2 = src[i++] & 0x7F;
        }
        number = (number << 7) | b2;
        if (number >= dict.size()) { ok = false; break; }
      }
      flip = (c == kEsc2) ? 0x20 : 0;
    } else {                                                                                         // :1503-1537
      if (c == 0x80) { flip = 0x20; if (i >= n) { ok = false; break; } c = src[i++]; }
      number = c & 0x7F;
      if (number >= 64) {
        if (number >= 112) { if (i + 2 > n) { ok = false; break; } number = ((number & 0x0F) << 16) | (src[i] << 8) | src[i + 1]; i += 2; }
        else { if (i >= n) { ok = false; break; } number = ((number & 0x1F) << 8) | src[i]; i++; }
        if (number > dict.size()) { ok = false; break; }
      } else if (number == 0) { ok = false; break; }
      number--;
    }
    if (number < 0 || number >= dict.size()) { ok = false; break; }
    const Word& e = dict.word(number);
    const int len = (int)(e.lenIdx >> 24) & 0xFF;
    if (afterWord && len > 1) { if (at >= end) { ok = false; break; } dst[at++] = ' '; }             // the implied space (:970-971)
    if (e.pos < 0 || at + len >= end) { ok = false; break; }                                         // :974-977
    dst[at++] = e.text[e.pos] ^ flip;
    if (len > 1) {
      memcpy(dst + at, e.text + e.pos + 1, (size_t)(len - 1));
      at += len - 1;
      afterWord = true;
      last = i;
    } else {                                       // one of TextCodec1's escape-byte words
      afterWord = false;
      last = i - 1;
    }
  }
  *produced = at;
  return (ok && i == n) ? 1 : 0;
}

// ---- UTF (UTFCodec.java) ----
inline int utf8_units(u8 lead) { static const int8_t t[16] = {1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 2, 2, 3, 4}; return t[lead >> 4]; }   // SIZES :30
inline int utf8_lead_ok(u8 b) { return b < 0x80 ? 1 : (b < 0xC2 ? 0 : (b < 0xE0 ? 2 : (b < 0xF0 ? 3 : (b < 0xF5 ? 4 : 0)))); }          // LEN_SEQ :32-41
// code point -> 22-bit key: 3 size bits + payload (:443-474); 0 units = not a lead byte
inline int utf8_key(const u8* p, u32* key) {
  const int s = utf8_units(p[0]);
  switch (s) {
    case 1: *key = p[0]; break;
    case 2: *key = (1u << 19) | ((u32)p[0] << 8) | p[1]; break;
    case 3: *key = (2u << 19) | ((u32)(p[0] & 0x0F) << 12) | ((u32)(p[1] & 0x3F) << 6) | (p[2] & 0x3F); break;
    case 4: *key = (4u << 19) | ((u32)(p[0] & 0x07) << 18) | ((u32)(p[1] & 0x3F) << 12) | ((u32)(p[2] & 0x3F) << 6) | (p[3] & 0x3F); break;
    default: *key = 0; break;
  }
  return s;
}
// key -> the UTF-8 bytes, little endian in a word, and their count (unpackV1 :514-548)
inline int utf8_bytes(u32 key, u32* le) {
  switch (key >> 19) {
    case 0: *le = key; return 1;
    case 1: *le = ((key & 0xFF) << 8) | ((key >> 8) & 0xFF); return 2;
    case 2: *le = (((key >> 12) & 0x0F) | 0xE0) | ((((key >> 6) & 0x3F) | 0x80) << 8) | (((key & 0x3F) | 0x80) << 16); return 3;
    case 4: case 5: case 6: case 7:
      *le = (((key >> 18) & 0x07) | 0xF0) | ((((key >> 12) & 0x3F) | 0x80) << 8) | ((((key >> 6) & 0x3F) | 0x80) << 16) | (((key & 0x3F) | 0x80) << 24);
      return 4;
    default: return 0;
  }
}

bool utf_validate(const u8* p, int start, int count) {                                               // :317-440
  PairCounts pc;
  memset(pc.c.get(), 0, 65536 * sizeof(int32_t));
  int32_t f0[256] = {0};
  auto forbidden = [&]() { int64_t s = f0[0xC0] + f0[0xC1]; for (int c = 0xF5; c <= 0xFF; c++) s += f0[c]; return s != 0; };
  int prev = 0;
  const int end = start + count, end4 = start + (count & -4);
  for (int i = start; i < end4; i += 4) {
    for (int k = 0; k < 4; k++) { const int c = p[i + k]; f0[c]++; pc.row(prev)[c]++; prev = c; }
    if ((i & 0x0FFF) == start && forbidden()) return false;      // the reference's early exit, as written
  }
  if (end4 != end) {
    for (int i = end4; i < end; i++) { const int c = p[i]; f0[c]++; pc.row(prev)[c]++; prev = c; }
    if (forbidden()) return false;                               // (only checked here when the length is not a multiple of 4)
  }
  // the pair rules; the single-byte rule is NOT re-checked for lengths that are multiples of 4 (as in the reference)
  int64_t cont = 0;
  for (int i = 0; i < 256; i++) {
    int64_t s = 0;
    if (i < 0xA0 || i > 0xBF) s += pc.row(0xE0)[i];
    if (i < 0x80 || i > 0x9F) s += pc.row(0xED)[i];
    if (i < 0x90 || i > 0xBF) s += pc.row(0xF0)[i];
    if (i < 0x80 || i > 0x8F) s += pc.row(0xF4)[i];
    if (i < 0x80 || i > 0xBF) {
      for (int j = 0xC2; j <= 0xDF; j++) s += pc.row(j)[i];
      for (int j = 0xE1; j <= 0xEC; j++) s += pc.row(j)[i];
      s += pc.row(0xF1)[i] + pc.row(0xF2)[i] + pc.row(0xF3)[i] + pc.row(0xEE)[i] + pc.row(0xEF)[i];
    } else cont += f0[i];
    if (s) return false;
  }
  return cont >= count / 8;
}

int utf_forward(int* dataType, const u8* src, int n, u8* dst, int dstCap, int* produced) {
  *produced = 0;
  if (n < kMinBlock) return 0;
  if (dstCap < n + 8192) return 0;
  bool mustValidate = true;
  if (dataType) {                                                                                    // :93-101
    if (*dataType != KZ_DT_UNDEFINED && *dataType != KZ_DT_UTF8) return 0;
    mustValidate = *dataType != KZ_DT_UTF8;
  }
  const int body = n - 4;                          // the last four bytes are copied (a code point may be cut by the block end)
  int start = 0;
  if (src[0] == 0xEF && src[1] == 0xBB && src[2] == 0xBF) start = 3;                                 // byte order mark
  else while (start < 4 && utf8_lead_ok(src[start]) == 0) start++;
  if (mustValidate && !utf_validate(src, start, body - start)) return 0;
  if (dataType) *dataType = KZ_DT_UTF8;
  // occurrences per key, later the alias per key.  The reference indexes an int[1 << 22] by key (UTFCodec.java:128); a block has
  // fewer than 32768 distinct keys, so a 65536-slot open-addressing map (512 KiB per host thread, cache resident, cleaned by slot)
  // gives the same answers
  struct Sym { int32_t key, freq; };
  struct KeyMap {
    // keys of 1-, 2- and 3-byte sequences are 16 payload bits under a 2-bit size tag: a direct table (3 x 65536 entries; a block
    // touches a few hundred of them); 4-byte sequences (21 payload bits, rare) go through a 65536-slot open-addressing map
    std::vector<uint32_t> direct, key, val; std::vector<uint32_t> used;
    KeyMap() : direct(3 * 65536, 0), key(65536, 0xFFFFFFFFu), val(65536, 0) {}
    uint32_t& at(uint32_t k) {                                   // the counter / alias of k, claimed (value 0) when new
      if (k < (4u << 19)) {
        const uint32_t i = ((k >> 19) << 16) | (k & 0xFFFFu);
        if (direct[i] == 0) used.push_back(i);                   // (may list an entry twice when it is reset to 0 by the caller: harmless)
        return direct[i];
      }
      uint32_t h = (k * 2654435761u) >> 16;
      while (key[h] != k) { if (key[h] == 0xFFFFFFFFu) { key[h] = k; val[h] = 0; used.push_back(0x80000000u | h); break; } h = (h + 1) & 0xFFFF; }
      return val[h];
    }
    void clear() {
      for (uint32_t u : used) { if (u & 0x80000000u) key[u & 0xFFFF] = 0xFFFFFFFFu; else direct[u] = 0; }
      used.clear();
    }
  };
  static thread_local KeyMap seenMap;
  struct Clean { KeyMap& m; ~Clean() { m.clear(); } } clean_{seenMap};
  std::vector<Sym> syms;
  bool ok = true;
  {
    // pass 1: occurrences per code point.  One-, two- and three-byte sequences index the direct table straight from their bytes
    // (index = size tag << 16 | the key's low 16 bits); the generic path is kept for four-byte sequences and for the error cases.
    uint32_t* const D = seenMap.direct.data();
    auto touch = [&](uint32_t idx, uint32_t key) {
      uint32_t& c = D[idx];
      if (__builtin_expect(c == 0, 0)) { seenMap.used.push_back(idx); syms.push_back(Sym{(int32_t)key, 0}); if (syms.size() >= 32768) ok = false; }
      c++;
    };
    int i = start;
    while (i < body && ok) {
      const u32 b0 = src[i];
      if (b0 < 0x80) { touch(b0, b0); i++; }
      else if (b0 >= 0xC0 && b0 < 0xE0) { const u32 lo = (b0 << 8) | src[i + 1]; touch((1u << 16) | lo, (1u << 19) | lo); i += 2; }
      else if (b0 >= 0xE0 && b0 < 0xF0) {
        const u32 b1 = src[i + 1], b2 = src[i + 2];
        if (b2 < 0x80 || b2 > 0xBF) { ok = false; break; }                                            // :140-145
        const u32 lo = ((b0 & 0x0F) << 12) | ((b1 & 0x3F) << 6) | (b2 & 0x3F);
        touch((2u << 16) | lo, (2u << 19) | lo); i += 3;
      } else {
        u32 key;
        const int s4 = utf8_key(src + i, &key);
        if (s4 != 4 || ((((u32)src[i + 2] << 8) | src[i + 3]) & 0xC0C0) != 0x8080) { ok = false; break; }
        uint32_t& cnt = seenMap.at(key);
        if (cnt == 0) { syms.push_back(Sym{(int32_t)key, 0}); if (syms.size() >= 32768) { ok = false; break; } }
        cnt++;
        i += 4;
      }
    }
  }
  const int nsym = (int)syms.size();
  const int maxTarget = n - n / 10;
  if (!ok || nsym == 0 || 3 * nsym + 6 >= maxTarget) return 0;
  for (Sym& s : syms) s.freq = (int32_t)seenMap.at((uint32_t)s.key);
  // most frequent first; equal counts: larger key first (the reference sorts ascending by (freq, key) and reads backwards)
  std::sort(syms.begin(), syms.end(), [](const Sym& a, const Sym& b) { return a.freq != b.freq ? a.freq > b.freq : a.key > b.key; });
  int at = 2;
  dst[at++] = (u8)(nsym >> 8); dst[at++] = (u8)nsym;
  int64_t estimate = at + 6;
  for (int r = 0; r < nsym; r++) {
    const int32_t key = syms[r].key;
    dst[at] = (u8)(key >> 16); dst[at + 1] = (u8)(key >> 8); dst[at + 2] = (u8)key;
    at += 3;
    estimate += (r < 128) ? syms[r].freq : 2 * (int64_t)syms[r].freq;
    seenMap.at((uint32_t)key) = (uint32_t)((r < 128) ? r : (0x10080 | ((r << 1) & 0xFF00) | (r & 0x7F)));                        // two-byte alias + its size in bits 16..
  }
  if (estimate >= maxTarget) return 0;
  // the map is not part of `estimate`: map + aliases can pass n + 8192 bytes (small blocks, thousands of distinct code points).  Such
  // an output is declined at the end in any case (:214): declined here, before a byte of it is written (the reference with buffers of
  // exactly getMaxEncodedLength bytes runs over its array instead: INTEGRATION.md 4)
  if ((int64_t)at + start + (estimate - 10) + 1 >= (int64_t)maxTarget) return 0;
  for (int i = 0; i < start; i++) dst[at++] = src[i];
  int i = start;
  {
    const uint32_t* const D = seenMap.direct.data();
    while (i < body) {                                              // pass 2: the alias of every code point (one or two bytes)
      const u32 b0 = src[i];
      u32 alias;
      if (b0 < 0x80) { alias = D[b0]; i++; }
      else if (b0 < 0xE0) { alias = D[(1u << 16) | (b0 << 8) | src[i + 1]]; i += 2; }
      else if (b0 < 0xF0) { alias = D[(2u << 16) | ((b0 & 0x0F) << 12) | ((src[i + 1] & 0x3Fu) << 6) | (src[i + 2] & 0x3Fu)]; i += 3; }
      else { u32 key; i += utf8_key(src + i, &key); alias = seenMap.at(key); }
      dst[at++] = (u8)alias;
      dst[at] = (u8)(alias >> 8);
      at += alias >> 16;
    }
  }
  dst[0] = (u8)start;
  dst[1] = (u8)(i - body);                         // how far the last code point reached into the four tail bytes
  while (i < n) dst[at++] = src[i++];
  *produced = at;
  return at < maxTarget ? 1 : 0;
}

int utf_inverse(const u8* src, int n, u8* dst, int dstCap, int* produced) {
  *produced = 0;
  if (n < 4) return 0;
  const int start = src[0] & 3, adjust = src[1] & 3;
  const int nsym = (src[2] << 8) | src[3];
  const int body = n - 4 + adjust, end = dstCap - 4;
  if (nsym == 0 || nsym >= 32768 || 3 * nsym >= n) return 0;
  struct Cp { u32 le; int len; };
  std::vector<Cp> map(nsym);
  int i = 4;
  for (int r = 0; r < nsym; r++, i += 3) {
    if (i + 3 > n) return 0;
    const u32 key = ((u32)src[i] << 16) | ((u32)src[i + 1] << 8) | src[i + 2];
    map[r].len = utf8_bytes(key, &map[r].le);
    if (map[r].len == 0) return 0;
  }
  if (end < 0) return 0;
  int at = 0;
  for (int k = 0; k < start; k++) { if (i >= n) return 0; dst[at++] = src[i++]; }
  while (i < body && at < end) {
    int alias = src[i++];
    if (alias >= 128) { if (i >= n) return 0; alias = (src[i++] << 7) + (alias & 0x7F); }
    if (alias >= nsym) return 0;
    const Cp& c = map[alias];
    dst[at] = (u8)c.le; dst[at + 1] = (u8)(c.le >> 8); dst[at + 2] = (u8)(c.le >> 16); dst[at + 3] = (u8)(c.le >> 24);
    at += c.len;
  }
  if (i < body || at >= end - n + body) { *produced = at; return 0; }
  for (int k = body; k < n; k++) {                 // the four tail bytes less `adjust` (:296-297); a two-byte alias that straddled `body` leaves too few
    if (i >= n) { *produced = at; return 0; }
    dst[at++] = src[i++];
  }
  *produced = at;
  return 1;
}

}  // namespace

// ---- the static dictionary as flat tables, for the device form of the TEXT inverse (kz_text_gpu.hip) ----
// words: sd.count static words, then TextCodec1's two escape words (:600-603); text: the lower-cased dictionary text + the two escape bytes
void kz_text_static_tables(std::vector<uint32_t>& hash, std::vector<int32_t>& pos, std::vector<uint32_t>& lenIdx, std::vector<uint8_t>& text,
                           std::vector<uint8_t>& delim, int* count) {
  const StaticDict& sd = static_dict();
  *count = sd.count;
  hash.assign((size_t)sd.count + 2, 0); pos.assign((size_t)sd.count + 2, 0); lenIdx.assign((size_t)sd.count + 2, 0);
  for (int i = 0; i < sd.count; i++) { hash[i] = sd.words[i].hash; pos[i] = sd.words[i].pos; lenIdx[i] = sd.words[i].lenIdx; }
  text.assign(sd.text, sd.text + KZ_DICT_EN_1024_LEN);
  text.push_back(kEsc2); text.push_back(kEsc1);
  hash[sd.count] = 0; pos[sd.count] = KZ_DICT_EN_1024_LEN; lenIdx[sd.count] = (1u << 24) | (u32)sd.count;
  hash[sd.count + 1] = 0; pos[sd.count + 1] = KZ_DICT_EN_1024_LEN + 1; lenIdx[sd.count + 1] = (1u << 24) | (u32)(sd.count + 1);
  delim.assign(256, 0);
  for (int c = 0; c < 256; c++) delim[c] = sd.delim[c] ? 1 : 0;
}

// ---- entry points used by kz_api.hip / kz_stream.hip ----
bool kz_is_host_transform(int type) { return type == KZ_T_TEXT || type == KZ_T_UTF; }

// the writer's tag from the block's first four bytes (CompressedOutputStream.java:795-804)
int kz_host_block_data_type(const uint8_t* p, int n, int init) {
  if (n < 4) return init;
  const int32_t m = mm_magic_type(p);
  if (mm_is_compressed(m)) return KZ_DT_BIN;
  if (mm_is_multimedia(m)) return KZ_DT_MULTIMEDIA;
  if (mm_is_executable(m)) return KZ_DT_EXE;
  return init;
}

// entropyType / blockSize: the context entries "entropy" and "blockSize" of the reference's map
int kz_host_transform_forward(int type, int entropyType, int blockSize, int* dataType, const uint8_t* src, int n,
                              uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  if (n == 0) return 1;
  if (type == KZ_T_TEXT) {
    if (n < kMinBlock || n > kMaxBlock) return 0;                                                     // TextCodec.forward :491-492
    const bool type2 = entropyType == KZ_E_NONE || entropyType == KZ_E_ANS0 || entropyType == KZ_E_HUFFMAN || entropyType == 4 /* RANGE */;
    return text_forward(type2 ? 2 : 1, blockSize, dataType, src, n, dst, dstCap, produced);
  }
  if (type == KZ_T_UTF) return utf_forward(dataType, src, n, dst, dstCap, produced);
  return 0;
}
int kz_host_transform_inverse(int type, int blockSize, const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  if (n == 0) return 1;
  if (type == KZ_T_TEXT) { if (n > kMaxBlock) return 0; return text_inverse(blockSize, src, n, dst, dstCap, produced); }
  if (type == KZ_T_UTF) return utf_inverse(src, n, dst, dstCap, produced);
  return 0;
}

// the same two stages without a context: nothing here touches a GPU (C-ABI, include/kanzi_hip.h)
extern "C" int32_t kz_host_stage_forward(uint32_t type, uint32_t entropyType, int32_t blockSize, int32_t* dataType,
                                         const uint8_t* src, int32_t n, uint8_t* dst, int32_t dstCap, int32_t* produced) {
  if (!src || !dst || !produced || n < 0) return -KZ_ERR_INVALID_PARAM;
  if (!kz_is_host_transform((int)type)) return -KZ_ERR_INVALID_CODEC;
  // TPAQX (9): the reference gives TEXT one more hash bit under it (TextCodec.java extraPerf); not modelled, so refused
  if (entropyType >= 9 || entropyType == 3) return -KZ_ERR_INVALID_CODEC;
  *produced = 0;
  if (dstCap < kz_transform_max_encoded_len(type, n)) return 0;
  int dt = dataType ? *dataType : KZ_DT_UNDEFINED;
  const int r = kz_host_transform_forward((int)type, (int)entropyType, blockSize, dataType ? &dt : nullptr, src, n, dst, dstCap, produced);
  if (dataType) *dataType = dt;
  if (!r) *produced = 0;
  return r ? 1 : 0;
}
extern "C" int32_t kz_host_stage_inverse(uint32_t type, int32_t blockSize, const uint8_t* src, int32_t n,
                                         uint8_t* dst, int32_t dstCap, int32_t* produced) {
  if (!src || !dst || !produced || n < 0) return -KZ_ERR_INVALID_PARAM;
  if (!kz_is_host_transform((int)type)) return -KZ_ERR_INVALID_CODEC;
  const int r = kz_host_transform_inverse((int)type, blockSize, src, n, dst, dstCap, produced);
  if (!r) *produced = 0;
  return r ? 1 : 0;
}
