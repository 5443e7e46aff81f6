// kz_text_fwd_gpu.hip -- TEXT forward (TextCodec2) on the device: K/transform/TextCodec.java:1124-1300 (TextCodec2.forward),
// :269-384 (computeStats, the text branch), :1304-1394 (emitSymbols / emitWordIndex) for blocks that sit in HBM.
//
// The host form (kz_text.hip: text_forward) stays the reference of this file and the fallback: a block the device does not take or
// does not finish cleanly (not text by the order-0 rules -- detectType then wants the whole pair table --, a Magic number in front, fewer
// than 1 024 bytes, a word list that would wrap at 2^19 words, an output within 8 bytes of the block's length) is left untouched and
// goes through the host stage, which gives the reference's verdict, bytes and "dataType" entry.  The device form only has to be
// exact on what it accepts.
//
// What the forward depends on, and what that allows:
//   * word boundaries, the two hashes of a word (as written / first letter's case flipped) and the cost of every plain byte depend on
//     the block's bytes alone;
//   * the dictionary changes only when a word is LEARNED: not found, eligible by length, and the hash slot of the word as written
//     EMPTY (TextCodec.java:725: `e1 == null`) -- a few thousand times per block against a million lookups; a lookup never writes;
//   * a found word replaces its letters by a 1 - 3 byte number (+ 0x80 when the case of the first letter was flipped), the bytes
//     between two found words go out as plain bytes, a single space between two found words is implied.
// So: k_tf_stats / k_tf_decide = computeStats' text branch (order-0 histogram and the four pair counts it reads);
//     k_tf_walk  = one wave per block walks rows of 64 bytes: every lane that closes a word hashes it and looks it up on its own
//                  (map entry = hash | length | number | position, 16 bytes, two loads in flight per lane); rows in which some word
//                  could be learned replay their words one by one from that word on, with the dictionary updated in between;
//                  a found word leaves (number + 1 | length << 24 | flip << 31) at its first letter in `tok`;
//     k_tf_emit<false> / k_tf_scan / k_tf_emit<true> = output bytes per 1 024 positions (and the letters of every found word against
//                  its dictionary word), offsets and the verdict, the bytes.
#include "kz_device.h"
#include "kz_internal.h"
#include "kz_magic.h"
#include "kz_datatype.h"
#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <vector>

typedef uint8_t u8;
typedef uint32_t u32;
typedef unsigned long long u64;

void kz_text_static_tables(std::vector<uint32_t>& hash, std::vector<int32_t>& pos, std::vector<uint32_t>& lenIdx, std::vector<uint8_t>& text,
                           std::vector<uint8_t>& delim, int* count);   // kz_text.hip

#define TF_T2 (128 * 128)        // TextCodec.java:32-35
#define TF_T3 64
#define TF_T4 (64 * 128)
#define TF_MAXDICT (1 << 19)
#define TF_MAXWORD 31
#define TF_MINBLOCK 1024
#define TF_MAXBLOCK (1 << 24)    // larger blocks: host stage (the token word keeps the word number in 24 bits, positions in 31)
#define TF_LF 0x0Au
#define TF_CR 0x0Du
#define TF_ESC1 0x0Fu
#define TF_HASH1 0x7FEB352Du
#define TF_HASH2 0x846CA68Bu
#define TF_CRLF 0x40
#define TF_XML 0x20
#define TF_CODEC2 0x10
#define TF_STATS 264             // ints per block: 256 byte counts, "&" + a/g/l/q, CR + not LF, not CR + LF
#define TF_TILE 1024             // positions per wave of the emit passes
#define TF_MARGIN 8

struct TextFwd {
  const u8* sText; const u8* delim; const u64* sMap; const int32_t* sPos; int sMapN; int sCount;   // static dictionary: text, delimiter set, (slot, e0, e1) triples
  u64* map;            // [A][slotsPer][2]: hash | (length << 24 | number) << 32,  position | valid << 32 | static << 33
  u32* tok;            // [A][NS]
  u32* wpos;           // [A][TF_MAXDICT] position of a learned word's first letter, by word number
  int32_t* tileSum;    // [A][maxTiles] output bytes per tile, then their exclusive prefix
  int32_t* stats;      // [A][TF_STATS]
  int32_t* mode;       // [A] the block's mode byte (TextCodec.java:48-52); -1: not text (k_tf_pairs then says what detectType says); -2: not for the device
  int32_t* tdt;        // [A] blocks that are not text: the "dataType" entry TEXT leaves when it declines (:650-665), -1: the host decides
  int32_t* fail;       // [A] the walk gave up
  const int32_t* ord;  // [B] dense index of the blocks this launch takes, -1: not taken
  int32_t* outLen;     // [B] produced bytes, -1: not done (host stage)
  int64_t slotsPer, NS;
  int maxTiles;
  u32 mask;
};

typedef u32 __attribute__((aligned(1))) tf_u32u;
__device__ __forceinline__ bool tf_is_text(u32 c) { const u32 l = c | 0x20u; return l >= 'a' && l <= 'z'; }     // (c < 256: bytes >= 0x80 never are)
__device__ __forceinline__ u32 tf_sx(u32 c) { return (u32)(int32_t)(int8_t)c; }                                 // Java's bytes are signed
__device__ __forceinline__ u64 tf_ld(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tf_st(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- computeStats, text branch (:269-384): the order-0 histogram and the pair counts the text path reads ----
__global__ __launch_bounds__(256) void k_tf_stats(const u8* __restrict__ srcAll, int64_t stride, const int32_t* __restrict__ d_len, TextFwd G, int B) {
  const int b = blockIdx.y;
  const int a = G.ord[b];
  if (a < 0) return;
  const int n = d_len[b];
  __shared__ u32 hist[4][256];
  for (int i = threadIdx.x; i < 1024; i += 256) (&hist[0][0])[i] = 0;
  __syncthreads();
  const u8* src = srcAll + (int64_t)b * stride;
  const int per = (((n + (int)gridDim.x - 1) / (int)gridDim.x) + 255) & ~255;
  const int beg = blockIdx.x * per, end = min(n, beg + per);
  const int wave = threadIdx.x >> 6, lane = kz_lane();
  u32 amp = 0, crx = 0, xlf = 0;
  for (int i = beg + threadIdx.x; i < end + 255; i += 256) {
    const bool valid = i < end;
    const u32 c = valid ? (u32)src[i] : 0u;
    const u32 p = (valid && i > 0) ? (u32)src[i - 1] : 0u;                          // the first byte's previous byte is 0 (:283)
    // one LDS atomic per distinct byte of the row would be ideal; rows of one byte (spaces, runs) are the expensive case: add once
    const uint64_t vm = kz_ballot(valid);
    if (vm == 0) break;
    const u32 c0 = (u32)__builtin_amdgcn_readfirstlane((int)c);
    if (kz_ballot(valid && c == c0) == vm) { if (lane == (int)__builtin_ctzll(vm)) atomicAdd(&hist[wave][c0], (u32)__popcll(vm)); }
    else if (valid) atomicAdd(&hist[wave][c], 1u);
    if (valid) {
      amp += (p == '&') & ((c == 'a') | (c == 'g') | (c == 'l') | (c == 'q'));
      crx += (p == TF_CR) & (c != TF_LF);
      xlf += (c == TF_LF) & (p != TF_CR);
    }
  }
  amp = kz_wave_sum(amp); crx = kz_wave_sum(crx); xlf = kz_wave_sum(xlf);
  int32_t* st = G.stats + (int64_t)a * TF_STATS;
  if (lane == 0) { if (amp) atomicAdd(&st[256], (int32_t)amp); if (crx) atomicAdd(&st[257], (int32_t)crx); if (xlf) atomicAdd(&st[258], (int32_t)xlf); }
  __syncthreads();
  const u32 v = hist[0][threadIdx.x] + hist[1][threadIdx.x] + hist[2][threadIdx.x] + hist[3][threadIdx.x];
  if (v) atomicAdd(&st[threadIdx.x], (int32_t)v);
}

// one thread per block: text or not (:310-331), the mode byte (:333-381)
__global__ __launch_bounds__(64) void k_tf_decide(const u8* __restrict__ srcAll, int64_t stride, const int32_t* __restrict__ d_len, TextFwd G, int B) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const int a = G.ord[b];
  if (a < 0) return;
  const int n = d_len[b];
  const int32_t* f = G.stats + (int64_t)a * TF_STATS;
  int mode = -2;
  if (n >= TF_MINBLOCK && n < TF_MAXBLOCK && mm_magic_type(srcAll + (int64_t)b * stride) == 0) {       // :272-273, :491-492
    mode = -1;
    long long letters = (long long)f[TF_CR] + f[TF_LF], ascii = 0;
    for (int c = 0; c < 128; c++) { if (tf_is_text((u32)c)) letters += f[c]; ascii += f[c]; }
    const long long bin = n - ascii;
    bool notText = bin > (n >> 2);
    if (!notText) notText = letters < n / 4 || f[32] < n / 50;                                         // :321, :326 (not strict)
    if (!notText) {
      int m = 0;
      if (bin <= n - n / 10) {                                                                           // :336-356
        const int lt = f['<'], gt = f['>'];
        const int minFreq = max((int)((n - bin) >> 9), 2);
        if (lt >= minFreq && gt >= minFreq && f[256] > 0) {
          const int lo = min(lt, gt), hi = max(lt, gt);
          if (lo == hi || lo >= hi - hi / 100) m |= TF_XML;
        }
      }
      if (f[TF_CR] != 0 && f[TF_CR] == f[TF_LF] && f[257] == 0 && f[258] == 0) m |= TF_CRLF;           // :358-372
      mode = m;
    }
  }
  G.mode[a] = mode;
}

// Blocks that are not text: TextCodec.detectType (:386-440) = Global.detectSimpleType on the order-0 histogram, else the UTF-8
// plausibility rules on the pair histogram (a lead byte followed by a byte outside its range anywhere: not UTF-8; continuation bytes
// >= 1/8 of the block: UTF-8).  Only pairs whose FIRST byte is a lead byte (>= 0xC2) are consulted: 64 rows of the pair table in LDS.
// The pair statistics start from a previous byte of 0 (:283), which is no lead byte.
__global__ __launch_bounds__(256) void k_tf_pairs(const u8* __restrict__ srcAll, int64_t stride, const int32_t* __restrict__ d_len, TextFwd G, int B) {
  const int b = blockIdx.x;
  const int a = G.ord[b];
  if (a < 0 || G.mode[a] != -1) return;
  const int n = d_len[b];
  const u8* src = srcAll + (int64_t)b * stride;
  __shared__ u32 tab[64][256];
  __shared__ long long lds4[4];
  __shared__ int anyBad;
  for (int i = threadIdx.x; i < 64 * 256; i += 256) (&tab[0][0])[i] = 0;
  if (threadIdx.x == 0) anyBad = 0;
  __syncthreads();
  for (int i = 4 * (int)threadIdx.x; i < n; i += 1024) {                 // four bytes per thread and step (slack behind the block: masked by i + k < n)
    const u32 w = *(const u32*)(src + i);
    u32 p = (i > 0) ? (u32)src[i - 1] : 0u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u32 c = (w >> (8 * k)) & 0xFFu;
      if (i + k < n && p >= 0xC0u) atomicAdd(&tab[p - 0xC0u][c], 1u);
      p = c;
    }
  }
  __syncthreads();
  const int32_t* f = G.stats + (int64_t)a * TF_STATS;
  const int i = threadIdx.x;
  const int simple = kz_detect_simple_type_wg(n, f[i], f['='], lds4);
#define TF_P(row, col) ((long long)tab[(row) - 0xC0][col])
  long long sbad = 0;
  if (i < 0xA0 || i > 0xBF) sbad += TF_P(0xE0, i);
  if (i < 0x80 || i > 0x9F) sbad += TF_P(0xED, i);
  if (i < 0x90 || i > 0xBF) sbad += TF_P(0xF0, i);
  if (i < 0x80 || i > 0x8F) sbad += TF_P(0xF4, i);
  long long cont = 0;
  if (i < 0x80 || i > 0xBF) {
    for (int j = 0xC2; j <= 0xDF; j++) sbad += TF_P(j, i);
    for (int j = 0xE1; j <= 0xEC; j++) sbad += TF_P(j, i);
    sbad += TF_P(0xF1, i) + TF_P(0xF2, i) + TF_P(0xF3, i) + TF_P(0xEE, i) + TF_P(0xEF, i);
  } else cont = f[i];
#undef TF_P
  if (i == 0xC0 || i == 0xC1 || i >= 0xF5) sbad += f[i];                 // bytes no UTF-8 text holds
  if (sbad) anyBad = 1;
  const long long contAll = kz_wg256_sum64(cont, lds4);                  // (its barriers also publish anyBad)
  if (threadIdx.x == 0) G.tdt[a] = (simple != DT_UNDEFINED) ? simple : ((!anyBad && contAll >= n / 8) ? DT_UTF8 : DT_UNDEFINED);
}

// the static dictionary's map entries (the map itself was cleared by the host call)
__global__ __launch_bounds__(256) void k_tf_init(TextFwd G, int B) {
  const int b = blockIdx.y;
  const int a = G.ord[b];
  if (a < 0 || G.mode[a] < 0) return;
  u64* map = G.map + (int64_t)a * G.slotsPer * 2;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < G.sMapN; i += gridDim.x * 256) {
    const u64 slot = G.sMap[3 * i];
    map[2 * slot] = G.sMap[3 * i + 1];
    map[2 * slot + 1] = G.sMap[3 * i + 2];
  }
}

#define TF_VALID (1ULL << 32)
#define TF_STATIC (1ULL << 33)

// ---- the walk: which words are found, under which number ----
// One row of 64 bytes per step.  A lane whose byte closes a word (a delimiter behind 2 .. 31 letters) takes the word's letters from an
// LDS ring (the previous row and this one) into registers, hashes it both ways and requests the two map entries; four rows are
// prepared and their entries requested before the first of them is looked at, so the map's latency is paid once per four rows.  A lookup that
// matches hash and length counts as found here; that the letters match too (sameWords :720-723) is checked for every found word by
// the size pass (k_tf_emit<false>) afterwards, in parallel -- a block with a single mismatch (a 32-bit hash collision at equal length) goes to the host
// stage whole, so the speculation never shows.  Rows in which a word could be learned replay their words one by one from that word
// on with fresh loads; the entries of the rows prepared behind it are then requested again.
struct TfRow { uint64_t candM; u32 h1, h2; int len, ws; u64 a0, a1, b0, b1; bool cand; };

__global__ __launch_bounds__(64) void k_tf_walk(const u8* __restrict__ srcAll, int64_t stride, const int32_t* __restrict__ d_len, TextFwd G, int B) {
  const int b = blockIdx.x;
  if (b >= B) return;
  const int a = G.ord[b];
  if (a < 0 || G.mode[a] < 0) return;
  const int lane = kz_lane();
  const int n = d_len[b];
  const u8* src = srcAll + (int64_t)b * stride;
  u64* map = G.map + (int64_t)a * G.slotsPer * 2;
  u32* tok = G.tok + (int64_t)a * G.NS;
  u32* wpos = G.wpos + (int64_t)a * TF_MAXDICT;
  const u32 mask = G.mask;
  __shared__ u32 ring32[32];                      // 128 bytes: the previous row and this one (a word is at most 31 letters)
  u8* ring = (u8*)ring32;
  const uint64_t dm0 = kz_ballot(G.delim[lane] != 0), dm1 = kz_ballot(G.delim[64 + lane] != 0), dm2 = kz_ballot(G.delim[128 + lane] != 0), dm3 = kz_ballot(G.delim[192 + lane] != 0);
#define TF_DELIM(cc) ((((cc) < 64u ? dm0 : ((cc) < 128u ? dm1 : ((cc) < 192u ? dm2 : dm3))) >> ((cc) & 63u)) & 1ULL)
  const uint64_t lt = kz_lanemask_lt();
  int words = G.sCount;                           // the next word number (TextCodec2: the static words, nothing else fixed)
  int carry = -1;                                 // position of the last non-letter in front of the row (:694: a letter first = -1 / the last leading space)
  bool failed = false;
  // one letter of the words the row closes (letters: no sign to extend), and a group of four behind a test that some word is that long
#define TF_HSTEP(R, k)                                                                                                \
  { const bool act = R.cand && (k) < R.len;                                                                           \
    const u32 t = ((xw[(k) >> 2] >> (8 * ((k) & 3))) & 0xFFu) * TF_HASH2;                                             \
    const u32 n1 = h1 * TF_HASH1 ^ t, n2 = h2 * TF_HASH1 ^ t;                                                         \
    h1 = act ? n1 : h1; h2 = act ? n2 : h2; }
#define TF_HGROUP(R, i) if (kz_ballot(R.cand && 4 * (i) < R.len) != 0) { TF_HSTEP(R, 4 * (i)) TF_HSTEP(R, 4 * (i) + 1) TF_HSTEP(R, 4 * (i) + 2) TF_HSTEP(R, 4 * (i) + 3)
  // prepare the row at `row` (its bytes: cc): word ends, hashes, map requests
#define TF_PREPARE(R, row, cc)                                                                                        \
  { const u32 c = (cc);                                                                                               \
    const int p = (row) + lane;                                                                                       \
    __builtin_amdgcn_wave_barrier();      /* one wave: its LDS accesses stay in order; a workgroup barrier would also wait for the map loads in flight */ \
    ring[p & 127] = (u8)c;                                                                                            \
    __builtin_amdgcn_wave_barrier();                                                                                  \
    const bool inb = p < n;                                                                                           \
    const bool isT = inb && tf_is_text(c);                                                                            \
    const uint64_t NT = ~kz_ballot(isT);                                                                              \
    const uint64_t below = NT & lt;                                                                                   \
    const int anchor = below ? (row) + 63 - (int)__builtin_clzll(below) : carry;                                      \
    R.len = p - anchor - 1;                                                                                           \
    R.ws = anchor + 1;                                                                                                \
    R.cand = inb && !isT && TF_DELIM(c) && R.len >= 2 && R.len <= TF_MAXWORD;                          /* :704, :708 */ \
    if (NT) carry = (row) + 63 - (int)__builtin_clzll(NT);                                                            \
    R.candM = kz_ballot(R.cand);                                                                                      \
    R.h1 = 0; R.h2 = 0; R.a0 = 0; R.a1 = 0; R.b0 = 0; R.b1 = 0;                                                       \
    if (R.candM) {                                                                                                    \
      u32 xw[8];                                                                                                      \
      { const int wb = (R.ws & 127) >> 2; const u32 sh = (u32)R.ws & 3u;                                              \
        u32 rr[9];                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 9; i++) rr[i] = ring32[(wb + i) & 31];                                  \
        _Pragma("unroll") for (int i = 0; i < 8; i++) xw[i] = __builtin_amdgcn_alignbyte(rr[i + 1], rr[i], sh); }     \
      const u32 w0 = xw[0] & 0xFFu;                                                                    /* :709-718 */ \
      u32 h1 = TF_HASH1 * TF_HASH1 ^ w0 * TF_HASH2, h2 = TF_HASH1 * TF_HASH1 ^ (w0 ^ 0x20u) * TF_HASH2;               \
      /* four letters per group with constant register and shift (a loop over k indexes xw dynamically: seven selects per letter) */ \
      TF_HSTEP(R, 1) TF_HSTEP(R, 2) TF_HSTEP(R, 3)                                                                    \
      TF_HGROUP(R, 1) TF_HGROUP(R, 2) TF_HGROUP(R, 3) TF_HGROUP(R, 4) TF_HGROUP(R, 5) TF_HGROUP(R, 6) TF_HGROUP(R, 7)  \
      }}}}}}}                                                                                                         \
      R.h1 = h1; R.h2 = h2;                                                                                           \
      TF_REQUEST(R)                                                                                                   \
    } }
#define TF_REQUEST(R)                                                                                                 \
  if (R.cand) {                                                                                                       \
    const u32 s1_ = R.h1 & mask, s2_ = R.h2 & mask;                                                                   \
    R.a0 = tf_ld(map + 2 * (u64)s1_); R.a1 = tf_ld(map + 2 * (u64)s1_ + 1);                                           \
    R.b0 = tf_ld(map + 2 * (u64)s2_); R.b1 = tf_ld(map + 2 * (u64)s2_ + 1);                                           \
  }
  // decide the row S (at `row`); `learned` is set when the dictionary changed: the rows prepared behind it ask again
#define TF_DECIDE(S)                                                                                                  \
  if (S.candM && !failed) {                                                                                           \
    const bool v1 = (S.a1 & TF_VALID) != 0, v2 = (S.b1 & TF_VALID) != 0;                                              \
    const bool m1 = S.cand && v1 && (u32)S.a0 == S.h1 && (int)(S.a0 >> 56) == S.len;                                  \
    const bool m2 = S.cand && !m1 && v2 && (u32)S.b0 == S.h2 && (int)(S.b0 >> 56) == S.len;                           \
    const u64 e0 = m1 ? S.a0 : S.b0;                                                                                  \
    bool found = m1 || m2;                                                                                            \
    u32 number = (u32)(e0 >> 32) & 0x00FFFFFFu;                                                                       \
    bool flip = m2 && !(v1 && ((u32)(S.a0 >> 32) & 0x00FFFFFFu) == number);          /* :761 `e == e1`: the same entry through both slots */ \
    /* a row that can learn a word: from that word on, one word at a time */                                           \
    const uint64_t learnM = kz_ballot(S.cand && !found && !v1 && (S.len > 3 || (S.len == 3 && words < TF_T2)));        \
    if (learnM) {                                                                                                     \
      uint64_t rest = S.candM & ~((1ULL << (int)__builtin_ctzll(learnM)) - 1ULL);                                     \
      while (rest) {                                                                                                  \
        const int j = (int)__builtin_ctzll(rest);                                                                     \
        rest &= rest - 1;                                                                                             \
        const u32 uh1 = (u32)__builtin_amdgcn_readlane((int)S.h1, j), uh2 = (u32)__builtin_amdgcn_readlane((int)S.h2, j); \
        const int ulen = __builtin_amdgcn_readlane(S.len, j), uws = __builtin_amdgcn_readlane(S.ws, j);               \
        const u32 us1 = uh1 & mask, us2 = uh2 & mask;                                                                 \
        const u64 ua0 = tf_ld(map + 2 * (u64)us1), ua1 = tf_ld(map + 2 * (u64)us1 + 1);                               \
        const u64 ub0 = tf_ld(map + 2 * (u64)us2), ub1 = tf_ld(map + 2 * (u64)us2 + 1);                               \
        const bool uv1 = (ua1 & TF_VALID) != 0, uv2 = (ub1 & TF_VALID) != 0;                                          \
        const bool um1 = uv1 && (u32)ua0 == uh1 && (int)(ua0 >> 56) == ulen;                                          \
        const bool um2 = !um1 && uv2 && (u32)ub0 == uh2 && (int)(ub0 >> 56) == ulen;                                  \
        const u64 ue0 = um1 ? ua0 : ub0;                                                                              \
        const bool ufound = um1 || um2;                                                                               \
        const u32 unum = (u32)(ue0 >> 32) & 0x00FFFFFFu;                                                              \
        const bool uflip = um2 && !(uv1 && ((u32)(ua0 >> 32) & 0x00FFFFFFu) == unum);                                 \
        if (!ufound && !uv1 && (ulen > 3 || (ulen == 3 && words < TF_T2))) {                           /* :725-747 */ \
          /* the word takes the next number; that number's record is a fresh one (hash 0): its "old" slot, slot 0, leaves the map (:729-731) */ \
          if (lane == 0) {                                                                                            \
            tf_st(map + 1, 0ULL);                                                                                     \
            tf_st(map + 2 * (u64)us1, (u64)uh1 | ((u64)(((u32)ulen << 24) | (u32)words) << 32));                      \
            tf_st(map + 2 * (u64)us1 + 1, (u64)(u32)uws | TF_VALID);                                                  \
            wpos[words] = (u32)uws;                                                                                   \
          }                                                                                                           \
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                            \
          learned = true;                                                                                             \
          words++;                                                                                                    \
          if (words >= TF_MAXDICT - 1) { failed = true; break; }        /* the numbering would restart (:742-746): host stage */ \
        }                                                                                                             \
        if (lane == j) { found = ufound; number = unum; flip = uflip; }                                               \
      }                                                                                                               \
    }                                                                                                                 \
    if (found && S.cand) tok[S.ws] = (number + 1u) | ((u32)S.len << 24) | (flip ? 0x80000000u : 0u);                  \
  }
  // four rows per step: all four are prepared and their entries requested before the first is decided (the loads are used inside the
  // step that issued them, so the compiler counts them exactly: no wait for the youngest load at the loop's back edge)
#define TF_SRC(row) (((row) + lane < n) ? (u32)src[(row) + lane] : 0u)
  u32 cN0 = TF_SRC(0), cN1 = TF_SRC(64), cN2 = TF_SRC(128), cN3 = TF_SRC(192);
  for (int row = 0; row < n && !failed; row += 256) {
    const u32 c0 = cN0, c1 = cN1, c2 = cN2, c3 = cN3;
    cN0 = TF_SRC(row + 256); cN1 = TF_SRC(row + 320); cN2 = TF_SRC(row + 384); cN3 = TF_SRC(row + 448);
    TfRow R0, R1, R2, R3;
    TF_PREPARE(R0, row, c0)
    TF_PREPARE(R1, row + 64, c1)
    TF_PREPARE(R2, row + 128, c2)
    TF_PREPARE(R3, row + 192, c3)
    bool learned = false;
    TF_DECIDE(R0)
    if (learned) { TF_REQUEST(R1) TF_REQUEST(R2) TF_REQUEST(R3) learned = false; }
    TF_DECIDE(R1)
    if (learned) { TF_REQUEST(R2) TF_REQUEST(R3) learned = false; }
    TF_DECIDE(R2)
    if (learned) { TF_REQUEST(R3) learned = false; }
    TF_DECIDE(R3)
  }
  if (lane == 0 && failed) G.fail[a] = 1;
#undef TF_PREPARE
#undef TF_HSTEP
#undef TF_HGROUP
#undef TF_REQUEST
#undef TF_DECIDE
#undef TF_SRC
#undef TF_DELIM
}

// ---- output: sizes per tile, then the bytes (:752-771 between the words, emitSymbols :1304-1367, emitWordIndex :1370-1394) ----
template <bool WRITE>
__global__ __launch_bounds__(256) void k_tf_emit(const u8* __restrict__ srcAll, u8* __restrict__ dstAll, int64_t stride, const int32_t* __restrict__ d_len, TextFwd G, int B) {
  const int b = blockIdx.y;
  const int a = G.ord[b];
  if (a < 0) return;
  const int mode = G.mode[a];
  if (mode < 0) return;
  if (WRITE && G.outLen[b] < 0) return;
  const int n = d_len[b];
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int base = t * TF_TILE;
  if (base >= n) return;
  const int lane = kz_lane();
  const u8* src = srcAll + (int64_t)b * stride;
  u8* dst = dstAll + (int64_t)b * stride;
  const u32* tok = G.tok + (int64_t)a * G.NS;
  const bool crlf = (mode & TF_CRLF) != 0;
  const uint64_t le = kz_lanemask_lt() | (1ULL << lane);
  u32 off = WRITE ? 1u + (u32)G.tileSum[(int64_t)a * G.maxTiles + t] : 0u;
  if (WRITE && t == 0 && lane == 0) dst[0] = (u8)(mode | TF_CODEC2);                                  // TextCodec.java:496-501
  u32 tPrev = (base >= 64) ? tok[base - 64 + lane] : 0u;
  for (int r = 0; r < TF_TILE / 64; r++) {
    const int row = base + r * 64;
    if (row >= n) break;
    const int p = row + lane;
    const bool inb = p < n;
    const u32 tCur = inb ? tok[p] : 0u;
    const u32 tNxt = (p + 1 < n) ? tok[p + 1] : 0u;
    const u32 c = inb ? (u32)src[p] : 0u;
    const uint64_t curM = kz_ballot(tCur != 0), prevM = kz_ballot(tPrev != 0);
    // the last found word that starts at or in front of p: does it cover p, does it end right in front of p
    const uint64_t m = curM & le;
    const int sl = m ? 63 - (int)__builtin_clzll(m) : (prevM ? 63 - (int)__builtin_clzll(prevM) : 0);
    const u32 tA = (u32)__shfl((int)tCur, sl, 64), tB = (u32)__shfl((int)tPrev, sl, 64);
    const u32 tS = m ? tA : (prevM ? tB : 0u);
    const int sPos = m ? row + sl : row - 64 + sl;
    const int L = (int)((tS >> 24) & 31u);
    const bool cov = tS != 0 && p < sPos + L;
    const bool ends = tS != 0 && p == sPos + L;
    const bool skipSp = ends && c == ' ' && tNxt != 0;                                                  // :752: a single space between two references
    u32 lit = 0;
    if (inb && !cov && !skipSp) lit = (c == TF_ESC1) ? 2u : ((c == TF_CR) ? (crlf ? 0u : 1u) : 1u + (c >> 7));
    const u32 v = tCur & 0x00FFFFFFu;
    const u32 code = tCur ? ((tCur >> 31) + (v >= TF_T4 ? 3u : (v >= TF_T3 ? 2u : 1u))) : 0u;
    if (!WRITE && tCur) {
      // sameWords (:720-723), which the walk left out: the letters behind the first against the dictionary word's, four at a time
      // (both buffers have slack behind them).  A mismatch -- a hash collision at equal length -- fails the block: host stage.
      const int number = (int)v - 1, len = (int)((tCur >> 24) & 31u);
      const u8* w = (number < G.sCount) ? G.sText + G.sPos[number] : src + G.wpos[(int64_t)a * TF_MAXDICT + number];
      u32 diff = (*(const tf_u32u*)(src + p) ^ *(const tf_u32u*)w) & (len >= 4 ? 0xFFFFFF00u : ((1u << (8 * len)) - 1u) & 0xFFFFFF00u);
      for (int k = 4; k < len; k += 4) {
        const u32 mm = (len - k >= 4) ? 0xFFFFFFFFu : (1u << (8 * (len - k))) - 1u;
        diff |= (*(const tf_u32u*)(src + p + k) ^ *(const tf_u32u*)(w + k)) & mm;
      }
      if (diff) G.fail[a] = 1;
    }
    const u32 sz = lit + code;
    const u32 incl = kz_wave_incl_sum(sz);
    if (WRITE && sz) {
      u8* o = dst + off + incl - sz;
      if (tCur) {
        if (tCur >> 31) *o++ = 0x80;
        if (v >= TF_T4) { o[0] = (u8)(0xF0u | (v >> 16)); o[1] = (u8)(v >> 8); o[2] = (u8)v; }
        else if (v >= TF_T3) { o[0] = (u8)(0xC0u | (v >> 8)); o[1] = (u8)v; }
        else o[0] = (u8)(0x80u | v);
      } else if (c == TF_ESC1) { o[0] = (u8)TF_ESC1; o[1] = (u8)TF_ESC1; }
      else if (c & 0x80u) { o[0] = (u8)TF_ESC1; o[1] = (u8)c; }
      else o[0] = (u8)c;
    }
    off += (u32)__builtin_amdgcn_readlane((int)incl, 63);
    tPrev = tCur;
  }
  if (!WRITE && lane == 0) G.tileSum[(int64_t)a * G.maxTiles + t] = (int32_t)off;
}

// tile offsets and the verdict: the reference gives up when the output comes within 3 bytes of the block's length at a word
// (:756) or runs over it at the end (:797); a block that ends at least TF_MARGIN bytes short of its length never meets either
__global__ __launch_bounds__(256) void k_tf_scan(const int32_t* __restrict__ d_len, TextFwd G, int B) {
  const int b = blockIdx.x;
  const int a = G.ord[b];
  if (a < 0) return;
  if (G.mode[a] < 0) { if (threadIdx.x == 0) G.outLen[b] = -1; return; }
  const int n = d_len[b];
  const int nt = (n + TF_TILE - 1) / TF_TILE;
  const int per = (nt + 255) / 256;
  int32_t* ts = G.tileSum + (int64_t)a * G.maxTiles;
  __shared__ u32 lds[32];
  u32 run = 0;
  for (int i = 0; i < per; i++) { const int t = threadIdx.x * per + i; if (t < nt) run += (u32)ts[t]; }
  u32 total;
  u32 ex = kz_wg_excl_sum(run, lds, &total);
  for (int i = 0; i < per; i++) { const int t = threadIdx.x * per + i; if (t < nt) { const u32 v = (u32)ts[t]; ts[t] = (int32_t)ex; ex += v; } }
  if (threadIdx.x == 0) G.outLen[b] = (!G.fail[a] && (long long)total + 1 <= (long long)n - TF_MARGIN) ? (int32_t)(total + 1) : -1;
}

__global__ __launch_bounds__(256) void k_tf_copy_back(const u8* __restrict__ src, u8* __restrict__ dst, int64_t stride, const int32_t* __restrict__ len, const int32_t* __restrict__ cond) {
  const int b = blockIdx.y;
  if (!cond[b]) return;
  const int n16 = (len[b] + 15) >> 4;
  const uint4* s = (const uint4*)(src + (int64_t)b * stride);
  uint4* d = (uint4*)(dst + (int64_t)b * stride);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) d[i] = s[i];
}

static int tf_log_v2(int blockSize) { int l = 13; if (blockSize >= 32) l = std::max(std::min(31 - __builtin_clz((unsigned)(blockSize / 32)), 24), 13); return l; }   // :1068-1081

// bytes of scratch the stage wants for a batch of B blocks of at most maxLen bytes, all of them text (kz_api.hip sizes the arena with it):
// 4 bytes of `tok` per input byte, 16 bytes per hash slot and 4 per word number
size_t kz_text_fwd_gpu_scratch(int B, int blockSize, int maxLen) {
  const size_t NS = kz_align((size_t)maxLen + 64, 256);
  return (size_t)B * (NS * 4 + ((size_t)32 << tf_log_v2(blockSize)) + (size_t)TF_MAXDICT * 4 + (size_t)(maxLen / TF_TILE + 2) * 4 + TF_STATS * 4 + 96) + (1 << 20);
}

// The stage in three steps, so that the caller can run the host stages of the blocks the device does not keep while the walk runs:
//   kz_text_fwd_gpu_classify: statistics of the blocks with take[b] != 0 in bt.buf[cur]; keeps[b] = 1 for the blocks the device goes on
//                             with (text by computeStats' rules, no Magic number, 1 KiB .. 16 MiB); declined[b] >= 0 for the blocks
//                             that are NOT text: TEXT is done with them (it declines: data untouched) and leaves that "dataType"
//                             (detectType on the device); -1 and not kept: the host stage decides; the stream is idle on return;
//   kz_text_fwd_gpu_launch:   queues the walk and the output passes for them (no wait);
//   kz_text_fwd_gpu_finish:   waits; done[b] = 1 for the blocks that were finished: their bytes are in their slots, bt.h_len / bt.d_len
//                             updated -- their skip bit, and "dataType" = TEXT, are the caller's; every other block is untouched (host stage).
// A job that found no room for its scratch keeps nothing.
struct TextFwdJob {
  TextFwd G;
  std::vector<int32_t> ord;
  int32_t *dOrd = nullptr, *dOut = nullptr, *dCond = nullptr;
  size_t mark = 0;
  kz_ctx* holder = nullptr;                         // the context whose arena holds this job's scratch above `mark` (null: nothing held)
  int A = 0, kept = 0;
  bool live = false;
};
TextFwdJob* kz_text_fwd_gpu_new() { return new TextFwdJob(); }
// the job's owner drops it on every path out of the encode call: an early error return gives the arena back here
void kz_text_fwd_gpu_free(TextFwdJob* J) { if (J && J->holder) J->holder->arenaTop = J->mark; delete J; }

int kz_text_fwd_gpu_classify(kz_ctx* ctx, kz_batch& bt, int blockSize, const std::vector<int32_t>& take, std::vector<int32_t>& keeps,
                             std::vector<int32_t>& declined, TextFwdJob& J) {
  const int B = bt.B;
  keeps.assign(B, 0);
  declined.assign(B, -1);
  J.ord.assign(B, -1);
  J.live = false;
  int A = 0, maxLen = 0;
  for (int b = 0; b < B; b++) if (take[b] && bt.h_len[b] >= TF_MINBLOCK && bt.h_len[b] < TF_MAXBLOCK) { J.ord[b] = A++; maxLen = std::max(maxLen, bt.h_len[b]); }
  J.A = A;
  if (A == 0) return 0;
  static std::vector<uint32_t> hHash, hLenIdx; static std::vector<int32_t> hPos; static std::vector<uint8_t> hText, hDelim; static int hCount = -1;
  static std::once_flag once;
  std::call_once(once, [] { kz_text_static_tables(hHash, hPos, hLenIdx, hText, hDelim, &hCount); });
  TextFwd& G = J.G;
  const int logV2 = tf_log_v2(blockSize);
  G.slotsPer = (int64_t)1 << logV2; G.mask = (u32)(G.slotsPer - 1);
  G.NS = (int64_t)kz_align((size_t)maxLen + 64, 256);
  G.maxTiles = maxLen / TF_TILE + 2;
  G.sCount = hCount;
  // the static words' map entries for this map size: the words take their slots in order, a later word takes a shared slot (:1109-1113)
  std::vector<u64> sMap;
  {
    std::unordered_map<u32, int> owner;
    for (int i = 0; i < hCount; i++) owner[hHash[i] & G.mask] = i;
    for (const auto& kv : owner) {
      const int i = kv.second;
      sMap.push_back((u64)kv.first);
      sMap.push_back((u64)hHash[i] | ((u64)hLenIdx[i] << 32));
      sMap.push_back((u64)(u32)hPos[i] | TF_VALID | TF_STATIC);
    }
  }
  G.sMapN = (int)(sMap.size() / 3);
  hipStream_t st = ctx->stream;
  J.mark = ctx->arenaTop; J.holder = ctx;
  u8* dText = (u8*)kz_arena_alloc(ctx, hText.size() + 64);
  u8* dDelim = (u8*)kz_arena_alloc(ctx, 256);
  u64* dSMap = (u64*)kz_arena_alloc(ctx, sMap.size() * 8 + 64);
  int32_t* dSPos = (int32_t*)kz_arena_alloc(ctx, (size_t)(hCount + 2) * 4);
  J.dOrd = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  J.dOut = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  J.dCond = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  G.mode = (int32_t*)kz_arena_alloc(ctx, (size_t)A * 4);
  G.fail = (int32_t*)kz_arena_alloc(ctx, (size_t)A * 4);
  G.stats = (int32_t*)kz_arena_alloc(ctx, (size_t)A * TF_STATS * 4);
  G.tdt = (int32_t*)kz_arena_alloc(ctx, (size_t)A * 4);
  if (!dText || !dDelim || !dSMap || !dSPos || !J.dOrd || !J.dOut || !J.dCond || !G.mode || !G.fail || !G.stats || !G.tdt) { ctx->arenaTop = J.mark; J.holder = nullptr; return 0; }
  KZ_HIP(hipMemcpyAsync(dText, hText.data(), hText.size(), hipMemcpyHostToDevice, st));
  KZ_HIP(hipMemcpyAsync(dDelim, hDelim.data(), 256, hipMemcpyHostToDevice, st));
  KZ_HIP(hipMemcpyAsync(dSMap, sMap.data(), sMap.size() * 8, hipMemcpyHostToDevice, st));
  KZ_HIP(hipMemcpyAsync(dSPos, hPos.data(), (size_t)(hCount + 2) * 4, hipMemcpyHostToDevice, st));
  KZ_HIP(hipMemcpyAsync(J.dOrd, J.ord.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
  KZ_HIP(hipMemsetAsync(J.dOut, 0xFF, (size_t)B * 4, st));
  KZ_HIP(hipMemsetAsync(G.stats, 0, (size_t)A * TF_STATS * 4, st));
  KZ_HIP(hipMemsetAsync(G.fail, 0, (size_t)A * 4, st));
  KZ_HIP(hipMemsetAsync(G.tdt, 0xFF, (size_t)A * 4, st));
  G.sText = dText; G.delim = dDelim; G.sMap = dSMap; G.sPos = dSPos; G.ord = J.dOrd; G.outLen = J.dOut;
  G.map = nullptr; G.tok = nullptr; G.wpos = nullptr; G.tileSum = nullptr;
  const u8* src = bt.buf[bt.cur];
  KZ_LAUNCH(ctx, KID_TEXT_FWD, k_tf_stats, dim3(16, B), dim3(256), src, bt.stride, bt.d_len, G, B);
  KZ_LAUNCH(ctx, KID_TEXT_FWD, k_tf_decide, dim3((B + 63) / 64), dim3(64), src, bt.stride, bt.d_len, G, B);
  KZ_LAUNCH(ctx, KID_TEXT_FWD, k_tf_pairs, dim3(B), dim3(256), src, bt.stride, bt.d_len, G, B);
  std::vector<int32_t> mode(A), tdt(A);
  KZ_HIP(hipMemcpyAsync(mode.data(), G.mode, (size_t)A * 4, hipMemcpyDeviceToHost, st));
  KZ_HIP(hipMemcpyAsync(tdt.data(), G.tdt, (size_t)A * 4, hipMemcpyDeviceToHost, st));
  KZ_HIP(kz_stream_sync(ctx, st));              // (also: the pageable sources above)
  for (int b = 0; b < B; b++) if (J.ord[b] >= 0 && mode[J.ord[b]] == -1) declined[b] = tdt[J.ord[b]];    // not text: TEXT declines, this is the "dataType" it leaves
  // the blocks that are text get new dense numbers: the big tables are sized by them
  std::vector<int32_t> ord2(B, -1);
  int K = 0;
  for (int b = 0; b < B; b++) if (J.ord[b] >= 0 && mode[J.ord[b]] >= 0) ord2[b] = K++;
  J.kept = K;
  if (K == 0) { ctx->arenaTop = J.mark; J.holder = nullptr; return 0; }
  int32_t* dMode2 = (int32_t*)kz_arena_alloc(ctx, (size_t)K * 4);
  int32_t* dFail2 = (int32_t*)kz_arena_alloc(ctx, (size_t)K * 4);
  G.tileSum = (int32_t*)kz_arena_alloc(ctx, (size_t)K * (size_t)G.maxTiles * 4);
  G.wpos = (u32*)kz_arena_alloc(ctx, (size_t)K * TF_MAXDICT * 4);
  G.map = (u64*)kz_arena_alloc(ctx, (size_t)K * (size_t)G.slotsPer * 16);
  G.tok = (u32*)kz_arena_alloc(ctx, (size_t)K * (size_t)G.NS * 4);
  if (!dMode2 || !dFail2 || !G.tileSum || !G.wpos || !G.map || !G.tok) {
    if (ctx->sw.textGpuTrace) fprintf(stderr, "[textfwd] no scratch for %d blocks: host stage\n", K);
    ctx->arenaTop = J.mark; J.holder = nullptr; J.kept = 0; return 0;
  }
  std::vector<int32_t> mode2(K);
  for (int b = 0; b < B; b++) if (ord2[b] >= 0) { mode2[ord2[b]] = mode[J.ord[b]]; keeps[b] = 1; }
  J.ord = ord2;
  KZ_HIP(hipMemcpyAsync(J.dOrd, J.ord.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
  KZ_HIP(hipMemcpyAsync(dMode2, mode2.data(), (size_t)K * 4, hipMemcpyHostToDevice, st));
  KZ_HIP(hipMemsetAsync(dFail2, 0, (size_t)K * 4, st));
  KZ_HIP(kz_stream_sync(ctx, st));              // pageable sources
  G.mode = dMode2; G.fail = dFail2;
  J.live = true;
  return 0;
}

int kz_text_fwd_gpu_launch(kz_ctx* ctx, kz_batch& bt, TextFwdJob& J) {
  if (!J.live) return 0;
  const int B = bt.B, K = J.kept;
  TextFwd& G = J.G;
  hipStream_t st = ctx->stream;
  KZ_HIP(hipMemsetAsync(G.tileSum, 0, (size_t)K * (size_t)G.maxTiles * 4, st));
  KZ_HIP(hipMemsetAsync(G.map, 0, (size_t)K * (size_t)G.slotsPer * 16, st));
  KZ_HIP(hipMemsetAsync(G.tok, 0, (size_t)K * (size_t)G.NS * 4, st));
  const u8* src = bt.buf[bt.cur]; u8* dst = bt.buf[bt.cur ^ 1];
  KZ_LAUNCH(ctx, KID_TEXT_FWD, k_tf_init, dim3(2, B), dim3(256), G, B);
  KZ_LAUNCH(ctx, KID_TEXT_WALK, k_tf_walk, dim3(B), dim3(64), src, bt.stride, bt.d_len, G, B);
  KZ_LAUNCH(ctx, KID_TEXT_FWD, k_tf_emit<false>, dim3((G.maxTiles + 3) / 4, B), dim3(256), src, dst, bt.stride, bt.d_len, G, B);
  KZ_LAUNCH(ctx, KID_TEXT_FWD, k_tf_scan, dim3(B), dim3(256), bt.d_len, G, B);
  KZ_LAUNCH(ctx, KID_TEXT_FWD, k_tf_emit<true>, dim3((G.maxTiles + 3) / 4, B), dim3(256), src, dst, bt.stride, bt.d_len, G, B);
  return 0;
}

int kz_text_fwd_gpu_finish(kz_ctx* ctx, kz_batch& bt, TextFwdJob& J, std::vector<int32_t>& done) {
  const int B = bt.B;
  done.assign(B, 0);
  if (!J.live) return 0;
  hipStream_t st = ctx->stream;
  std::vector<int32_t> outLen(B);
  KZ_HIP(hipMemcpyAsync(outLen.data(), J.dOut, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  KZ_HIP(kz_stream_sync(ctx, st));
  std::vector<int32_t> cond(B, 0), newLen(bt.h_len);
  int any = 0, nDone = 0;
  for (int b = 0; b < B; b++) if (J.ord[b] >= 0 && outLen[b] >= 0) { cond[b] = 1; newLen[b] = outLen[b]; done[b] = 1; any = 1; nDone++; }
  if (any) {
    KZ_HIP(hipMemcpyAsync(J.dCond, cond.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    KZ_HIP(hipMemcpyAsync(J.dOut, newLen.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_tf_copy_back, dim3(64, B), dim3(256), 0, st, bt.buf[bt.cur ^ 1], bt.buf[bt.cur], bt.stride, J.dOut, J.dCond);
    for (int b = 0; b < B; b++) bt.h_len[b] = newLen[b];
    KZ_HIP(hipMemcpyAsync(bt.d_len, bt.h_len.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    KZ_HIP(kz_stream_sync(ctx, st));
  }
  KZ_HIP(hipGetLastError());
  if (ctx->sw.textGpuTrace) fprintf(stderr, "[textfwd] took %d blocks, finished %d (kept %d after the statistics)\n", J.A, nDone, J.kept);
  ctx->arenaTop = J.mark; J.holder = nullptr;
  J.live = false;
  return 0;
}
