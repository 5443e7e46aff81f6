// kz_text_gpu.hip -- TEXT inverse on the device (round 5, first form): K/transform/TextCodec.java:873-1100 (TextCodec1.inverse) and
// :1413-1600 (TextCodec2.inverse) for the blocks of a decoded batch that are still in HBM.
//
// The host form (kz_text.hip: text_inverse) stays the reference of this file and the fallback: every block this kernel does not
// finish cleanly (malformed input, output that does not fit, a variant the call did not size its tables for) is reported as
// "not done" and goes through the host stage, which gives the reference's verdict.  So the device form only has to be exact on
// what it accepts.
//
// One wave per block, the reference's token walk with wave-uniform control flow: the coded bytes come through a 64-byte row
// register (one coalesced load per row, the next row requested a row ahead), runs of letters are copied by the lanes of the row,
// dictionary words by the lanes of the word; the dictionary (hash slots + word table, Dictionary in kz_text.hip: same rules, same
// quirks) lives in the block's scratch in HBM and is read with uniform loads.  A token is one to three dependent loads (382 ns each
// with 2 048 blocks side by side, tools/ubench_gather chaseblk), so a block takes a few hundred ms -- all blocks of the batch side
// by side; what it buys is a decoder that does not wait for host CPUs.
#include "kz_device.h"
#include "kz_internal.h"
#include <algorithm>
#include <mutex>
#include <vector>

typedef uint8_t u8;
typedef uint32_t u32;

void kz_text_static_tables(std::vector<uint32_t>& hash, std::vector<int32_t>& pos, std::vector<uint32_t>& lenIdx, std::vector<uint8_t>& text,
                           std::vector<uint8_t>& delim, int* count);   // kz_text.hip

#define TG_T1 128            // TextCodec.java:32-35
#define TG_T2 (128 * 128)
#define TG_MAXDICT (1 << 19)
#define TG_MAXWORD 31
#define TG_LF 0x0Au
#define TG_CR 0x0Du
#define TG_ESC1 0x0Fu
#define TG_ESC2 0x0Eu
#define TG_HASH1 0x7FEB352Du
#define TG_HASH2 0x846CA68Bu
#define TG_CRLF 0x40u
#define TG_CODEC2 0x10u
#define TG_IDXMASK 0x0007FFFFu
#define TG_RING 2048            // output ring in LDS (bytes)
#define TG_SMAX 1032            // static words (1024) + TextCodec1's two escape words, rounded up
#define TG_STEXT 6400           // their text
#define TG_STATIC 0x00800000u    // word record: its text is the static dictionary's (else the block's coded bytes)

struct TgWord { u32 hash; int32_t pos; u32 lenIdx; };

struct TextGpu {
  const u32* sHash; const int32_t* sPos; const u32* sLenIdx; const u8* sText; const u8* delim; int sCount; int sTextLen;   // static tables (device)
  int32_t* slots;       // [A][slotsPer]
  TgWord* words;        // [A][TG_MAXDICT]
  const int32_t* ord;   // [B] dense index of the blocks this call takes, -1: not taken
  int32_t* outLen;      // [B] produced bytes, -1: not done (host stage)
  int64_t slotsPer;
  int logV1, logV2;     // hash map sizes of the two variants for the stream's block size
  int variantCap;       // largest variant the slots were sized for (1: both)
  int llog;             // initial size of the word list
  int dstCap;
};

__device__ __forceinline__ bool tg_is_text(u32 c) { const u32 l = c | 0x20u; return l >= 'a' && l <= 'z' && c < 0x80u; }
__device__ __forceinline__ u32 tg_u(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }

__global__ __launch_bounds__(64) void k_text_inv(const u8* __restrict__ srcAll, u8* __restrict__ dstAll, int64_t stride, const int32_t* __restrict__ d_len, TextGpu G, int B) {
  const int b = blockIdx.x;
  if (b >= B) return;
  const int a = G.ord[b];
  if (a < 0) return;
  const int lane = kz_lane();
  const int n = d_len[b];
  const u8* src = srcAll + (int64_t)b * stride;
  u8* dst = dstAll + (int64_t)b * stride;
  if (n <= 0) { if (lane == 0) G.outLen[b] = -1; return; }
  const u32 mode = tg_u(src[0]);
  const int variant = (mode & TG_CODEC2) ? 2 : 1;
  if (variant == 1 && G.variantCap < 1) { if (lane == 0) G.outLen[b] = -1; return; }
  const bool crlf = (mode & TG_CRLF) != 0;
  volatile int32_t* slots = G.slots + (int64_t)a * G.slotsPer;
  volatile TgWord* words = G.words + (int64_t)a * TG_MAXDICT;
  const u32 mask = (1u << (variant == 1 ? G.logV1 : G.logV2)) - 1u;
  const int fixed = G.sCount + (variant == 1 ? 2 : 0);
  int size = 1 << G.llog;
  int next = fixed;
  // ---- dictionary of the block (Dictionary::Dictionary): empty map, the static words, empty records behind them ----
  for (u32 i = (u32)lane; i <= mask; i += 64) slots[i] = -1;
  for (int i = lane; i < size; i += 64) {
    TgWord w;
    if (i < fixed) { w.hash = G.sHash[i]; w.pos = G.sPos[i]; w.lenIdx = G.sLenIdx[i] | TG_STATIC; }
    else { w.hash = 0; w.pos = -1; w.lenIdx = (u32)i; }
    words[i].hash = w.hash; words[i].pos = w.pos; words[i].lenIdx = w.lenIdx;
  }
  __syncthreads();
  if (lane == 0) for (int i = 0; i < fixed; i++) slots[G.sHash[i] & mask] = i;          // in order: a later word takes a shared slot
  __syncthreads();
  // the delimiter set as four 64-bit masks, the static dictionary (records and text) in LDS: most references of a text go there
  const uint64_t dm0 = kz_ballot(G.delim[lane] != 0), dm1 = kz_ballot(G.delim[64 + lane] != 0), dm2 = kz_ballot(G.delim[128 + lane] != 0), dm3 = kz_ballot(G.delim[192 + lane] != 0);
#define TG_DELIM(cc) ((((cc) < 64u ? dm0 : ((cc) < 128u ? dm1 : ((cc) < 192u ? dm2 : dm3))) >> ((cc) & 63u)) & 1ULL)
  __shared__ u32 sRec[TG_SMAX * 2];          // pos | len << 24 ... per static word: [2k] = pos, [2k+1] = lenIdx
  __shared__ u8 sTxt[TG_STEXT];
  for (int k = lane; k < fixed; k += 64) { sRec[2 * k] = (u32)G.sPos[k]; sRec[2 * k + 1] = G.sLenIdx[k] | TG_STATIC; }
  for (int k = lane; k < G.sTextLen; k += 64) sTxt[k] = G.sText[k];
  __syncthreads();
  // output through an LDS ring, written out 1 KiB at a time with 16-byte stores: a global store per token would put a memory
  // round trip in front of the token's next load (loads and stores share vmcnt on gfx9: the wait for a load covers the stores before it)
  __shared__ __attribute__((aligned(16))) u8 obuf[TG_RING];
  int flushed = 0;
#define TG_OUT(pos, v) obuf[(pos) & (TG_RING - 1)] = (u8)(v)
#define TG_FLUSH()                                                                                   \
  while (at - flushed >= 1024) {                                                                      \
    __syncthreads();                                                                                  \
    *(uint4*)(dst + flushed + 16 * lane) = *(const uint4*)(obuf + ((flushed + 16 * lane) & (TG_RING - 1)));  \
    flushed += 1024;                                                                                  \
    __syncthreads();                                                                                  \
  }
  const int end = G.dstCap;
  int i = 1, at = 0;
  bool ok = true, afterWord = false;
  if (i < n) {
    // row register: coded bytes [rowBase, rowBase + 64), one per lane; nxtRow = the row behind it
    int rowBase = 0;
    u32 row = (lane < n) ? (u32)src[lane] : 0u;
    u32 nxtRow = (64 + lane < n) ? (u32)src[64 + lane] : 0u;
#define TG_BYTE(idx) (((idx) - rowBase) < 64 ? (u32)__builtin_amdgcn_readlane((int)row, (idx) - rowBase) \
                      : (((idx) - rowBase) < 128 ? (u32)__builtin_amdgcn_readlane((int)nxtRow, (idx) - rowBase - 64) : tg_u((u32)src[idx])))
#define TG_ADVANCE_ROW()                                                                              \
    while (i - rowBase >= 64) { rowBase += 64; row = nxtRow; nxtRow = (rowBase + 64 + lane < n) ? (u32)src[rowBase + 64 + lane] : 0u; }
    int last = tg_is_text(TG_BYTE(1)) ? 0 : 1;
    while (i < n && at < end) {
      TG_ADVANCE_ROW()
      const int off = i - rowBase;
      u32 c = (u32)__builtin_amdgcn_readlane((int)row, off);
      if (tg_is_text(c)) {
        // a run of letters inside the row: plain bytes (:880-882), copied by their lanes
        const uint64_t nt = kz_ballot(!tg_is_text(row) || rowBase + lane >= n);
        const uint64_t m = nt >> off;
        int run = m ? (int)__builtin_ctzll(m) : 64 - off;
        run = min(run, end - at);
        if (lane >= off && lane < off + run) TG_OUT(at + lane - off, row);
        at += run; i += run;
        TG_FLUSH()
        continue;
      }
      if (i > last + 3 && TG_DELIM(c)) {                                     // the decoder learns only words of at least three letters (:891)
        const int len = i - last - 1;
        if (len <= TG_MAXWORD) {
          const u32 wb = (lane < len) ? (u32)src[last + 1 + lane] : 0u;       // the word's letters, one per lane
          u32 h = TG_HASH1;
          for (int k = 0; k < len; k++) {
            const u32 ch = (u32)__builtin_amdgcn_readlane((int)wb, k);
            h = h * TG_HASH1 ^ (u32)(int32_t)(int8_t)ch * TG_HASH2;
          }
          const int s1 = (int)tg_u((u32)slots[h & mask]);
          bool known = false;
          if (s1 >= 0) {
            const u32 eh = tg_u(words[s1].hash), eli = tg_u(words[s1].lenIdx);
            if (eh == h && (int)(eli >> 24) == len) {
              const int epos = (int)tg_u((u32)words[s1].pos);
              const u8* et = (eli & TG_STATIC) ? G.sText : src;
              const bool diff = (lane >= 1 && lane < len) && (u32)et[epos + lane] != wb;
              known = kz_ballot(diff) == 0;
            }
          }
          if (!known && (len > 3 || next < TG_T2) && s1 < 0) {                // Dictionary::learn
            const u32 oli = tg_u(words[next].lenIdx);
            if ((int)(oli & TG_IDXMASK) >= fixed) {
              const u32 oh = tg_u(words[next].hash);
              if (lane == 0) {
                slots[oh & mask] = -1;
                words[next].hash = h; words[next].pos = last + 1; words[next].lenIdx = ((u32)len << 24) | (u32)next;
              }
            }
            if (lane == 0) slots[h & mask] = next;
            next++;
            if (next >= size) {
              if (size >= TG_MAXDICT) next = fixed;
              else {
                for (int q = size + lane; q < 2 * size; q += 64) { words[q].hash = 0; words[q].pos = -1; words[q].lenIdx = (u32)q; }
                size *= 2;
              }
            }
            __syncthreads();                                                  // one wave: the stores above are ordered before what follows
          }
        }
      }
      i++;
      const bool ref = (variant == 1) ? (c == TG_ESC1 || c == TG_ESC2) : (c & 0x80u) != 0;
      if (!ref) {
        if (variant == 2 && c == TG_ESC1) {                                   // escaped byte >= 0x80 or a literal 0x0F (:1577-1578)
          if (i >= n) { ok = false; break; }
          const u32 lit = TG_BYTE(i);
          if (lane == 0) TG_OUT(at, lit);
          at++; i++;
        } else {
          if (crlf && c == TG_LF) { if (lane == 0) TG_OUT(at, TG_CR); at++; if (at >= end) { ok = false; break; } }
          if (lane == 0) TG_OUT(at, c);
          at++;
        }
        afterWord = false;
        last = i - 1;
        TG_FLUSH()
        continue;
      }
      int number;
      u32 flip = 0;
      if (variant == 1) {                                                     // :945-961
        if (i >= n) { ok = false; break; }
        number = (int)TG_BYTE(i); i++;
        if (number >= 128) {
          number &= 0x7F;
          if (i >= n) { ok = false; break; }
          int b2 = (int)(int8_t)TG_BYTE(i); i++;
          if (b2 & 0x80) {
            number = ((number & 0x1F) << 7) | (b2 & 0x7F);
            if (i >= n) { ok = false; break; }
            b2 = (int)(TG_BYTE(i) & 0x7Fu); i++;
          }
          number = (number << 7) | b2;
          if (number >= size) { ok = false; break; }
        }
        flip = (c == TG_ESC2) ? 0x20u : 0u;
      } else {                                                                // :1503-1537
        if (c == 0x80u) { flip = 0x20u; if (i >= n) { ok = false; break; } c = TG_BYTE(i); i++; }
        number = (int)(c & 0x7Fu);
        if (number >= 64) {
          if (number >= 112) { if (i + 2 > n) { ok = false; break; } number = ((number & 0x0F) << 16) | (int)(TG_BYTE(i) << 8) | (int)TG_BYTE(i + 1); i += 2; }
          else { if (i >= n) { ok = false; break; } number = ((number & 0x1F) << 8) | (int)TG_BYTE(i); i++; }
          if (number > size) { ok = false; break; }
        } else if (number == 0) { ok = false; break; }
        number--;
      }
      if (number < 0 || number >= size) { ok = false; break; }
      const bool isStat = number < fixed;
      const u32 eli = isStat ? tg_u(sRec[2 * number + 1]) : tg_u(words[number].lenIdx);
      const int epos = isStat ? (int)tg_u(sRec[2 * number]) : (int)tg_u((u32)words[number].pos);
      const int len = (int)(eli >> 24) & 0xFF;
      if (afterWord && len > 1) { if (at >= end) { ok = false; break; } if (lane == 0) TG_OUT(at, ' '); at++; }   // the implied space (:970-971)
      if (epos < 0 || at + len >= end) { ok = false; break; }                 // :974-977
      if (lane < len) {
        const u32 ch = isStat ? (u32)sTxt[epos + lane] : (u32)src[epos + lane];
        TG_OUT(at + lane, ch ^ (lane == 0 ? flip : 0u));
      }
      at += len;
      TG_FLUSH()
      if (len > 1) { afterWord = true; last = i; } else { afterWord = false; last = i - 1; }
    }
#undef TG_BYTE
#undef TG_ADVANCE_ROW
  }
  __syncthreads();
  for (int p = flushed + lane; p < at; p += 64) dst[p] = obuf[p & (TG_RING - 1)];        // the ring's last bytes
  if (lane == 0) G.outLen[b] = (ok && i == n) ? at : -1;
}

// ---- second form: a ROW of 64 coded bytes at a time (TextCodec2 blocks) ------------------------------------------------------
// The token boundaries of TextCodec2 depend on the coded bytes alone, and so does the delimiter anchor (the only dictionary
// words of one letter are static ones, known from LDS), so a row's tokens are found and classified by its lanes; what is
// serial is kept short: (1) the walk over the row's multi-byte-capable lead bytes (>= 0x80 or the escape) that decides which
// bytes start tokens, (2) the dictionary update for the row's literal-word candidates (a few thousand per block: hash, slot probe,
// tail compare, learn: the code of the first form), (3) the carry between rows (bytes owned by the previous row's last token, the
// anchor, the "after a word" flag, the output position).  References are resolved by their lanes (static words from LDS, learned
// words with one gather of the records and one 16-byte gather of the text each), output sizes are a wave scan, bytes go through the
// LDS ring.  A reference to a word learned LATER in the stream is what a serial decoder would have refused: the learn position of a
// record is pos + length, a reference in front of it -- like every other anomaly: a token cut by the block's end, a number out of
// range, an output that may not fit, a dictionary that wraps at 2^19 words -- gives the block to the host stage.
#define TG2_RING 4096
#ifdef TG_PROF
__device__ unsigned long long g_tgprof[8];
#define TG_T(k) { const long long t_ = clock64(); tacc[k] += t_ - tprev; tprev = t_; }
#else
#define TG_T(k)
#endif
__global__ __launch_bounds__(64) void k_text_inv2(const u8* __restrict__ srcAll, u8* __restrict__ dstAll, int64_t stride, const int32_t* __restrict__ d_len, TextGpu G, int B) {
  const int b = blockIdx.x;
  if (b >= B) return;
  const int a = G.ord[b];
  if (a < 0) return;
  const int lane = kz_lane();
  const int n = d_len[b];
  const u8* src = srcAll + (int64_t)b * stride;
  u8* dst = dstAll + (int64_t)b * stride;
  if (n <= 1) { if (lane == 0) G.outLen[b] = -1; return; }
  const u32 mode = tg_u(src[0]);
  if (!(mode & TG_CODEC2)) { if (lane == 0) G.outLen[b] = -1; return; }      // TextCodec1 blocks: the first form / the host
  const bool crlf = (mode & TG_CRLF) != 0;
  volatile int32_t* slots = G.slots + (int64_t)a * G.slotsPer;
  volatile TgWord* words = G.words + (int64_t)a * TG_MAXDICT;
  const u32 mask = (1u << G.logV2) - 1u;
  const int fixed = G.sCount;
  int size = 1 << G.llog;
  int next = fixed;
  for (u32 i = (u32)lane; i <= mask; i += 64) slots[i] = -1;
  for (int i = lane; i < size; i += 64) {
    if (i < fixed) { words[i].hash = G.sHash[i]; words[i].pos = G.sPos[i]; words[i].lenIdx = G.sLenIdx[i] | TG_STATIC; }
    else { words[i].hash = 0; words[i].pos = -1; words[i].lenIdx = (u32)i; }
  }
  __syncthreads();
  if (lane == 0) for (int i = 0; i < fixed; i++) slots[G.sHash[i] & mask] = i;
  const uint64_t dm0 = kz_ballot(G.delim[lane] != 0), dm1 = kz_ballot(G.delim[64 + lane] != 0);   // delimiters are ASCII: two masks
  __shared__ __attribute__((aligned(16))) u8 sT16[TG_SMAX * 16];     // the static words, 16 bytes each (the longest has 14 letters)
  __shared__ u8 sLen8[TG_SMAX];
  __shared__ __attribute__((aligned(16))) u8 obuf[TG2_RING];
  for (int k = lane; k < fixed; k += 64) {
    const int wl = (int)(G.sLenIdx[k] >> 24), wp = G.sPos[k];
    sLen8[k] = (u8)wl;
    for (int j = 0; j < 16; j++) sT16[16 * k + j] = j < wl ? G.sText[wp + j] : (u8)0;
  }
  __syncthreads();
  const int end = G.dstCap;
  int at = 0, flushed = 0;
  int skip = 1;                                  // leading bytes of the row that belong to a token of the row before (byte 0: the mode byte)
  int last = tg_is_text(tg_u((u32)src[1])) ? 0 : 1;
  bool afterWord = false, bad = false;
  const uint64_t ltm = kz_lanemask_lt();
  u32 nxt = (lane < n) ? (u32)src[lane] : 0u;
  u32 nxt2 = (64 + lane < n) ? (u32)src[64 + lane] : 0u;                       // two rows ahead: a row's last lanes look into the next one
#ifdef TG_PROF
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tprev = clock64();
#endif
  for (int rowBase = 0; rowBase < n && !bad; rowBase += 64) {
    TG_T(7)
    const u32 row = nxt;
    nxt = nxt2;
    nxt2 = (rowBase + 128 + lane < n) ? (u32)src[rowBase + 128 + lane] : 0u;
    const int pabs = rowBase + lane;
    const uint64_t validM = kz_ballot(pabs < n);
    // the three bytes behind every byte (the next row's first bytes for the last lanes)
    u32 b1 = (u32)__shfl_down((int)row, 1, 64), b2 = (u32)__shfl_down((int)row, 2, 64), b3 = (u32)__shfl_down((int)row, 3, 64);
    { const u32 n0 = (u32)__builtin_amdgcn_readlane((int)nxt, 0), n1 = (u32)__builtin_amdgcn_readlane((int)nxt, 1), n2 = (u32)__builtin_amdgcn_readlane((int)nxt, 2);
      if (lane == 63) { b1 = n0; b2 = n1; b3 = n2; } else if (lane == 62) { b2 = n0; b3 = n1; } else if (lane == 61) b3 = n0; }
    const bool isT = tg_is_text(row);
    const bool hi = (row & 0x80u) != 0;
    const bool pre = row == 0x80u;
    const u32 c2 = pre ? b1 : row;
    const u32 idx7 = c2 & 0x7Fu;
    const int ext = idx7 >= 112u ? 2 : (idx7 >= 64u ? 1 : 0);
    const int L = isT ? 1 : (hi ? (pre ? 2 : 1) + ext : (row == TG_ESC1 ? 2 : 1));
    // which bytes start tokens: walk the multi-byte-capable lead bytes in order, every other byte is a token of its own
    uint64_t consumed = skip >= 64 ? ~0ULL : ((1ULL << skip) - 1ULL);
    int newSkip = skip >= 64 ? skip - 64 : 0;
    uint64_t Cw = kz_ballot((hi || row == TG_ESC1) && pabs < n) & ~consumed;
    while (Cw) {
      const int p = (int)__builtin_ctzll(Cw);
      const int Lp = __builtin_amdgcn_readlane(L, p);
      const uint64_t m = (Lp >= 64 ? ~0ULL : ((1ULL << Lp) - 1ULL)) << p;
      consumed |= m & ~(1ULL << p);
      if (p + Lp > 64) newSkip = p + Lp - 64;
      Cw &= ~m;
    }
    TG_T(0)
    const uint64_t starts = validM & ~consumed;
    const bool st = (starts >> lane) & 1ULL;
    if (kz_ballot(st && pabs + L > n)) { bad = true; break; }                // a token cut by the end of the block
    const bool isRef = st && hi, isEsc = st && !isT && !hi && row == TG_ESC1, isLit = st && !isT && !hi && row != TG_ESC1;
    const uint64_t NT = kz_ballot(st && !isT);                               // non-letter tokens: they move the anchor
    // ---- references: their numbers; static ones know their word at once ----
    int num = -1; u32 flip = pre ? 0x20u : 0u;
    bool numBad = false;
    if (isRef) {
      const u32 nb1 = pre ? b2 : b1, nb2 = pre ? b3 : b2;
      int v = (int)idx7;
      if (ext == 2) v = (int)(((idx7 & 0x0Fu) << 16) | (nb1 << 8) | nb2);
      else if (ext == 1) v = (int)(((idx7 & 0x1Fu) << 8) | nb1);
      if (v == 0) numBad = true;                                             // (numbers past the dictionary: checked below, after the row's own words are in)
      num = v - 1;
    }
    const bool isStatRef = isRef && num >= 0 && num < fixed;
    u32 wLenIdx = 0; int wPos = -1;
    int wlen = isStatRef ? (int)sLen8[num] : (isRef ? 3 : 0);           // learned words have three letters or more: enough for the anchor
    // anchor behind every non-letter token; the anchor a token sees = that of the non-letter token before it (or the row's carry)
    const int anchorAfter = (isRef && wlen > 1) ? pabs + L : pabs + L - 1;
    const uint64_t below = NT & ltm;
    const int q = below ? 63 - (int)__builtin_clzll(below) : -1;
    const int qAnchor = __shfl(anchorAfter, q < 0 ? 0 : q, 64);
    const int myLast = q < 0 ? last : qAnchor;
    // ---- literal-word candidates of the row, in order: the dictionary update of the first form ----
    TG_T(1)
    uint64_t cand = kz_ballot(isLit && row < 128u && (((row < 64u ? dm0 : dm1) >> (row & 63u)) & 1ULL) && pabs > myLast + 3);
    while (cand) {
      const int p = (int)__builtin_ctzll(cand);
      cand &= cand - 1;
      const int ci = rowBase + p;
      const int clast = __builtin_amdgcn_readlane(myLast, p);
      const int len = ci - clast - 1;
      if (len > TG_MAXWORD) continue;
      const u32 wb = (lane < len) ? (u32)src[clast + 1 + lane] : 0u;
      u32 h = TG_HASH1;
      for (int k = 0; k < len; k++) {
        const u32 ch = (u32)__builtin_amdgcn_readlane((int)wb, k);
        h = h * TG_HASH1 ^ (u32)(int32_t)(int8_t)ch * TG_HASH2;
      }
      const int s1 = (int)tg_u((u32)slots[h & mask]);
      bool known = false;
      if (s1 >= 0) {
        const u32 eh = tg_u(words[s1].hash), eli = tg_u(words[s1].lenIdx);
        if (eh == h && (int)(eli >> 24) == len) {
          const int epos = (int)tg_u((u32)words[s1].pos);
          const u8* et = (eli & TG_STATIC) ? G.sText : src;
          const bool diff = (lane >= 1 && lane < len) && (u32)et[epos + lane] != wb;
          known = kz_ballot(diff) == 0;
        }
      }
      if (!known && (len > 3 || next < TG_T2) && s1 < 0) {
        const u32 oli = tg_u(words[next].lenIdx);
        if ((int)(oli & TG_IDXMASK) >= fixed) {
          const u32 oh = tg_u(words[next].hash);
          if (lane == 0) {
            slots[oh & mask] = -1;
            words[next].hash = h; words[next].pos = clast + 1; words[next].lenIdx = ((u32)len << 24) | (u32)next;
          }
        }
        if (lane == 0) slots[h & mask] = next;
        next++;
        if (next >= size) {
          if (size >= TG_MAXDICT) { bad = true; break; }                      // the numbering would restart: records change under references
          for (int k = size + lane; k < 2 * size; k += 64) { words[k].hash = 0; words[k].pos = -1; words[k].lenIdx = (u32)k; }
          size *= 2;
        }
        __syncthreads();
      }
    }
    if (bad) break;
    TG_T(2)
    // ---- learned words: one gather of the records, validity as a serial decoder would have seen it ----
    const bool isDynRef = isRef && !isStatRef;
    if (isDynRef) {
      if (num < 0 || num >= size) numBad = true;
      else {
        wLenIdx = words[num].lenIdx; wPos = words[num].pos;
        wlen = (int)(wLenIdx >> 24);
        if (wPos < 0 || wPos + wlen >= pabs || wlen < 3 || wlen > TG_MAXWORD) numBad = true;   // not learned yet at this point of the stream
      }
    }
    if (kz_ballot(numBad)) { bad = true; break; }
    TG_T(3)
    // ---- output sizes: a letter, a literal (CR LF for LF in CRLF mode), an escaped byte, a word with its implied space ----
    const bool qIsWord = __shfl((int)(isRef && wlen > 1), q < 0 ? 0 : q, 64) != 0;
    const bool afterW = q < 0 ? afterWord : qIsWord;
    const int sp = (isRef && afterW && wlen > 1) ? 1 : 0;
    int olen = 0;
    if (st) olen = isRef ? wlen + sp : ((isLit && crlf && row == TG_LF) ? 2 : 1);
    const u32 inc = kz_wave_incl_sum((u32)olen);
    const int total = (int)__builtin_amdgcn_readlane((int)inc, 63);
    if (at + total + 2 >= end) { bad = true; break; }                          // may not fit: the host stage decides
    const int o = at + (int)inc - olen;
    // texts: 16 bytes per lane from the static table in LDS or in one gather from the block (a second one for longer words)
    typedef u32 tg_u32x4 __attribute__((ext_vector_type(4)));
    typedef tg_u32x4 __attribute__((aligned(1))) tg_u32x4_u;
    tg_u32x4 ta = {0, 0, 0, 0}, tb = {0, 0, 0, 0};
    if (isStatRef) ta = *(const tg_u32x4*)(sT16 + 16 * num);
    else if (isDynRef) {
      ta = *(const tg_u32x4_u*)(src + wPos);                                    // (slots have >= 4 KiB of slack behind the block)
      if (wlen > 16) tb = *(const tg_u32x4_u*)(src + wPos + 16);
    }
    TG_T(4)
    if (st && !isRef) {
      if (isEsc) obuf[o & (TG2_RING - 1)] = (u8)b1;
      else if (olen == 2) { obuf[o & (TG2_RING - 1)] = (u8)TG_CR; obuf[(o + 1) & (TG2_RING - 1)] = (u8)row; }
      else obuf[o & (TG2_RING - 1)] = (u8)row;
    }
    if (isRef && sp) obuf[o & (TG2_RING - 1)] = (u8)' ';
    {
      const u32 mxv = kz_wave_incl_max((u32)(isRef ? wlen : 0));
      const int mxAll = __builtin_amdgcn_readlane((int)mxv, 63);
      const int ob = o + sp;
      const int wl = isRef ? wlen : 0;
      ta.x ^= flip;
      const u32 tw[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
#pragma unroll
      for (int kk = 0; kk < 8; kk++) {
        if (kk * 4 >= mxAll) break;
#pragma unroll
        for (int j = 0; j < 4; j++) if (kk * 4 + j < wl) obuf[(ob + kk * 4 + j) & (TG2_RING - 1)] = (u8)(tw[kk] >> (8 * j));
      }
    }
    at += total;
    TG_T(5)
    // ---- carries into the next row ----
    if (NT) {
      const int hq = 63 - (int)__builtin_clzll(NT);
      last = __builtin_amdgcn_readlane(anchorAfter, hq);
      afterWord = __builtin_amdgcn_readlane((int)(isRef && wlen > 1), hq) != 0;
    }
    skip = newSkip;
    while (at - flushed >= 1024) {
      __syncthreads();
      *(uint4*)(dst + flushed + 16 * lane) = *(const uint4*)(obuf + ((flushed + 16 * lane) & (TG2_RING - 1)));
      flushed += 1024;
      __syncthreads();
    }
    TG_T(6)
  }
#ifdef TG_PROF
  if (lane == 0) for (int k = 0; k < 8; k++) atomicAdd(&g_tgprof[k], (unsigned long long)tacc[k]);
#endif
  __syncthreads();
  if (!bad) for (int p = flushed + lane; p < at; p += 64) dst[p] = obuf[p & (TG2_RING - 1)];
  if (lane == 0) G.outLen[b] = (!bad && skip == 0) ? at : -1;
}

// ---- third form: the row form as three waves in lockstep, one step (barrier) per row --------------------------------------------
// wave 0 finds the token starts of row t, wave 1 does the anchors, the dictionary update and the record gather of row t-1 and
// completes row t-2 (whose records have arrived by then), wave 2 sizes row t-3 and fetches its texts, and writes row t-4 (whose
// texts have arrived).  What the waves hand on goes through LDS, two buffers deep: a 64-bit mask of token starts, then two words per
// lane (the token's class, its length, the static word's number or the learned word's position).  The barrier waits for LDS only,
// so that the gathers stay in flight across it.  The token starts are the fixed point of "a lead byte of a multi-byte token starts
// one unless an earlier one covers it", iterated on 64-bit masks: one more token of a chain of overlapping candidates is right per
// iteration, two or three iterations for most rows.
#define TG3_ST 0x00200000u
#define TG3_REF 0x00400000u
#define TG3_ESC 0x00800000u
#define TG3_STAT 0x01000000u
#define TG3_SP 0x02000000u
#define TG3_FLIP 0x04000000u
#define TG3_LF2 0x08000000u
// dictionary accesses of the wave that owns it: relaxed device-scope atomics (served by L2, no wait of their own: volatile ones are waited for one by one)
__device__ __forceinline__ u32 tg_ld(const volatile void* p) { return __hip_atomic_load((const u32*)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tg_st(volatile void* p, u32 v) { __hip_atomic_store((u32*)(uintptr_t)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__global__ __launch_bounds__(192) void k_text_inv3(const u8* __restrict__ srcAll, u8* __restrict__ dstAll, int64_t stride, const int32_t* __restrict__ d_len, TextGpu G, int B) {
  const int b = blockIdx.x;
  if (b >= B) return;
  const int a = G.ord[b];
  if (a < 0) return;
  const int tid = (int)threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // (wave: uniform, so that what each role carries stays in scalar registers)
  const int n = d_len[b];
  const u8* src = srcAll + (int64_t)b * stride;
  u8* dst = dstAll + (int64_t)b * stride;
  if (n <= 1) { if (tid == 0) G.outLen[b] = -1; return; }
  const u32 mode = tg_u(src[0]);
  if (!(mode & TG_CODEC2)) { if (tid == 0) G.outLen[b] = -1; return; }       // TextCodec1 blocks: the first form / the host
  const bool crlf = (mode & TG_CRLF) != 0;
  volatile int32_t* slots = G.slots + (int64_t)a * G.slotsPer;
  volatile TgWord* words = G.words + (int64_t)a * TG_MAXDICT;
  const u32 mask = (1u << G.logV2) - 1u;
  const int fixed = G.sCount;
  for (u32 i = (u32)tid; i <= mask; i += 192) slots[i] = -1;
  for (int i = tid; i < (1 << G.llog); i += 192) {
    if (i < fixed) { words[i].hash = G.sHash[i]; words[i].pos = G.sPos[i]; words[i].lenIdx = G.sLenIdx[i] | TG_STATIC; }
    else { words[i].hash = 0; words[i].pos = -1; words[i].lenIdx = (u32)i; }
  }
  __shared__ __attribute__((aligned(16))) u8 sT16[TG_SMAX * 16];     // the static words, 16 bytes each (the longest has 14 letters)
  __shared__ u8 sLen8[TG_SMAX];
  __shared__ __attribute__((aligned(16))) u8 obuf[TG2_RING + 256];   // (+ a spare word per lane)
  __shared__ uint64_t qStarts[2];
  __shared__ u32 qA[2][64], qB[2][64];
  __shared__ int sBad[2];
  for (int k = tid; k < fixed; k += 192) {
    const int wl = (int)(G.sLenIdx[k] >> 24), wp = G.sPos[k];
    sLen8[k] = (u8)wl;
    for (int j = 0; j < 16; j++) sT16[16 * k + j] = j < wl ? G.sText[wp + j] : (u8)0;
  }
  if (tid < 2) sBad[tid] = 0;
  __syncthreads();
  if (tid == 0) for (int i = 0; i < fixed; i++) slots[G.sHash[i] & mask] = i;
  __syncthreads();
  const int R = (n + 63) >> 6;
  const uint64_t ltm = kz_lanemask_lt();
  typedef u32 tg_u32x4 __attribute__((ext_vector_type(4)));
  typedef tg_u32x4 __attribute__((aligned(1))) tg_u32x4_u;
  // wave 0
  int skip = 1;                                  // leading bytes of the row that belong to a token of the row before (byte 0: the mode byte)
  // waves 0 and 1: the row and the two behind it (wave 1 runs one row late)
  u32 nxt = 0, nxt2 = 0;
  if (wave < 2) { nxt = (lane < n) ? (u32)src[lane] : 0u; nxt2 = (64 + lane < n) ? (u32)src[64 + lane] : 0u; }
  // wave 1
  const uint64_t dm0 = kz_ballot(G.delim[lane] != 0), dm1 = kz_ballot(G.delim[64 + lane] != 0);   // delimiters are ASCII: two masks
  int size = 1 << G.llog, next = fixed;
  int last = tg_is_text(tg_u((u32)src[1])) ? 0 : 1;
  bool afterWord = false;
  bool pHave = false;                            // the row whose records are in flight
  u32 pA = 0, pLenIdx = 0; int pNum = -1, pPos = -1, pPabs = 0; bool pDyn = false, pNumBad = false, pAfterW = false;
  // wave 2
  const int end = G.dstCap;
  int at = 0, done = 0, flushed = 0;             // sized up to / written up to / stored up to
  bool wHave = false;                            // the row whose texts are in flight
  u32 wA = 0; int wO = 0; tg_u32x4 wTa = {0, 0, 0, 0}, wTb = {0, 0, 0, 0}; int wEnd = 0;
  bool bad = false;
#ifdef TG_PROF
  long long tBody = 0, tWait = 0, tprev = clock64();
#endif
  for (int t = 0; t < R + 5; t++) {
    bool myBad = false;
#ifdef TG_PROF
    { const long long t_ = clock64(); tWait += t_ - tprev; tprev = t_; }
#endif
    if (wave < 2) {
      if (wave == 1 && pHave) {
        // ---- complete row t-2: its learned words as a serial decoder would have seen them ----
        int wlen = (int)((pA >> 16) & 31u);
        bool numBad = pNumBad;
        if (pDyn && !numBad) {
          wlen = (int)(pLenIdx >> 24);
          if (pPos < 0 || pPos + wlen >= pPabs || wlen < 3 || wlen > TG_MAXWORD) numBad = true;    // not learned yet at this point of the stream
        }
        if (kz_ballot(numBad)) myBad = true;
        u32 A = (pA & ~(31u << 16)) | ((u32)(wlen & 31) << 16);
        if ((A & TG3_REF) && pAfterW && wlen > 1) A |= TG3_SP;
        const int rp = t - 2;
        qA[rp & 1][lane] = A;
        qB[rp & 1][lane] = (A & TG3_STAT) ? (u32)pNum : (u32)pPos;
        pHave = false;
      }
      const int r = t - wave;
      if (r >= 0 && r < R) {
        const int rowBase = r << 6;
        const u32 row = nxt;
        nxt = nxt2;
        nxt2 = (rowBase + 128 + lane < n) ? (u32)src[rowBase + 128 + lane] : 0u;
        const int pabs = rowBase + lane;
        u32 b1 = (u32)__shfl_down((int)row, 1, 64), b2 = (u32)__shfl_down((int)row, 2, 64), b3 = (u32)__shfl_down((int)row, 3, 64);
        { const u32 n0 = (u32)__builtin_amdgcn_readlane((int)nxt, 0), n1 = (u32)__builtin_amdgcn_readlane((int)nxt, 1), n2 = (u32)__builtin_amdgcn_readlane((int)nxt, 2);
          if (lane == 63) { b1 = n0; b2 = n1; b3 = n2; } else if (lane == 62) { b2 = n0; b3 = n1; } else if (lane == 61) b3 = n0; }
        const bool isT = tg_is_text(row);
        const bool hi = (row & 0x80u) != 0;
        const bool pre = row == 0x80u;
        const u32 c2 = pre ? b1 : row;
        const u32 idx7 = c2 & 0x7Fu;
        const int ext = idx7 >= 112u ? 2 : (idx7 >= 64u ? 1 : 0);
        const int L = isT ? 1 : (hi ? (pre ? 2 : 1) + ext : (row == TG_ESC1 ? 2 : 1));
        if (wave == 0) {
          // which bytes start tokens: a lead byte of a multi-byte token starts one unless an earlier token covers it
          const uint64_t validM = kz_ballot(pabs < n);
          const uint64_t M2 = kz_ballot(L == 2 && pabs < n), M3 = kz_ballot(L == 3 && pabs < n), M4 = kz_ballot(L == 4 && pabs < n);
          const uint64_t M = M2 | M3 | M4;
          const uint64_t carry = skip >= 64 ? ~0ULL : ((1ULL << skip) - 1ULL);
          uint64_t Rl = M & ~carry, C;
          for (;;) {
            const uint64_t r34 = Rl & (M3 | M4), r4 = Rl & M4;
            C = carry | (Rl << 1) | (r34 << 2) | (r4 << 3);
            const uint64_t Rn = M & ~C;
            if (Rn == Rl) break;
            Rl = Rn;
          }
          const u32 t4 = (u32)((Rl & M4) >> 61), t3 = (u32)((Rl & M3) >> 62), t2 = (u32)((Rl & M2) >> 63);   // tokens that run into the next row
          skip = (t4 & 4u) ? 3 : (((t4 & 2u) | (t3 & 2u)) ? 2 : (((t4 & 1u) | (t3 & 1u) | t2) ? 1 : 0));
          const uint64_t starts = validM & ~C;
          const bool st = (starts >> lane) & 1ULL;
          if (kz_ballot(st && pabs + L > n)) myBad = true;                     // a token cut by the end of the block
          if (lane == 0) qStarts[r & 1] = starts;
        } else {
          const uint64_t starts = qStarts[r & 1];
          const bool st = (starts >> lane) & 1ULL;
          const bool isRef = st && hi, isEsc = st && !isT && !hi && row == TG_ESC1, isLit = st && !isT && !hi && row != TG_ESC1;
          const uint64_t NT = kz_ballot(st && !isT);                           // non-letter tokens: they move the anchor
          int num = -1;
          bool numBad = false;
          if (isRef) {
            const u32 nb1 = pre ? b2 : b1, nb2 = pre ? b3 : b2;
            int v = (int)idx7;
            if (ext == 2) v = (int)(((idx7 & 0x0Fu) << 16) | (nb1 << 8) | nb2);
            else if (ext == 1) v = (int)(((idx7 & 0x1Fu) << 8) | nb1);
            if (v == 0) numBad = true;                                         // (numbers past the dictionary: checked below, after the row's own words are in)
            num = v - 1;
          }
          const bool isStatRef = isRef && num >= 0 && num < fixed;
          const int wlen = isStatRef ? (int)sLen8[num] : (isRef ? 3 : 0);      // learned words have three letters or more: enough for the anchor
          const int anchorAfter = (isRef && wlen > 1) ? pabs + L : pabs + L - 1;
          const uint64_t below = NT & ltm;
          const int q = below ? 63 - (int)__builtin_clzll(below) : -1;
          const int qAnchor = __shfl(anchorAfter, q < 0 ? 0 : q, 64);
          const int myLast = q < 0 ? last : qAnchor;
          const bool qIsWord = __shfl((int)(isRef && wlen > 1), q < 0 ? 0 : q, 64) != 0;
          const bool afterW = q < 0 ? afterWord : qIsWord;
          // literal-word candidates of the row, in order: the dictionary update of the first form
          uint64_t cand = kz_ballot(isLit && row < 128u && (((row < 64u ? dm0 : dm1) >> (row & 63u)) & 1ULL) && pabs > myLast + 3);
          while (cand) {
            const int p = (int)__builtin_ctzll(cand);
            cand &= cand - 1;
            const int ci = rowBase + p;
            const int clast = __builtin_amdgcn_readlane(myLast, p);
            const int len = ci - clast - 1;
            if (len > TG_MAXWORD) continue;
            const u32 wb = (lane < len) ? (u32)src[clast + 1 + lane] : 0u;
            u32 h = TG_HASH1;
            for (int k = 0; k < len; k++) {
              const u32 ch = (u32)__builtin_amdgcn_readlane((int)wb, k);
              h = h * TG_HASH1 ^ (u32)(int32_t)(int8_t)ch * TG_HASH2;
            }
            const u32 oli = tg_ld(&words[next].lenIdx), oh = tg_ld(&words[next].hash);          // (the record a new word would take)
            const int s1 = (int)tg_u(tg_ld(&slots[h & mask]));
            bool known = false;
            if (s1 >= 0) {
              const u32 eh = tg_ld(&words[s1].hash), eli = tg_ld(&words[s1].lenIdx);
              const int epos = (int)tg_ld(&words[s1].pos);
              if (tg_u(eh) == h && (int)(tg_u(eli) >> 24) == len) {
                const u8* et = (tg_u(eli) & TG_STATIC) ? G.sText : src;
                const bool diff = (lane >= 1 && lane < len) && (u32)et[(int)tg_u((u32)epos) + lane] != wb;
                known = kz_ballot(diff) == 0;
              }
            }
            if (!known && (len > 3 || next < TG_T2) && s1 < 0) {
              if ((int)(tg_u(oli) & TG_IDXMASK) >= fixed) {
                if (lane == 0) {
                  tg_st(&slots[tg_u(oh) & mask], (u32)-1);
                  tg_st(&words[next].hash, h); tg_st(&words[next].pos, (u32)(clast + 1)); tg_st(&words[next].lenIdx, ((u32)len << 24) | (u32)next);
                }
              }
              if (lane == 0) tg_st(&slots[h & mask], (u32)next);
              next++;
              if (next >= size) {
                if (size >= TG_MAXDICT) { myBad = true; break; }                // the numbering would restart: records change under references
                for (int k = size + lane; k < 2 * size; k += 64) { tg_st(&words[k].hash, 0u); tg_st(&words[k].pos, (u32)-1); tg_st(&words[k].lenIdx, (u32)k); }
                size *= 2;
              }
              __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");             // one wave: the stores above are done before the loads that follow
            }
          }
          // learned words: one gather of the records, looked at a step later
          pDyn = isRef && !isStatRef;
          pNumBad = numBad;
          if (pDyn) {
            if (num < 0 || num >= size) pNumBad = true;
            else { pLenIdx = tg_ld(&words[num].lenIdx); pPos = (int)tg_ld(&words[num].pos); }
          }
          u32 A = (row & 0xFFu) | ((b1 & 0xFFu) << 8) | ((u32)(wlen & 31) << 16);
          if (st) A |= TG3_ST;
          if (isRef) A |= TG3_REF;
          if (isEsc) A |= TG3_ESC;
          if (isStatRef) A |= TG3_STAT;
          if (isRef && pre) A |= TG3_FLIP;
          if (isLit && crlf && row == TG_LF) A |= TG3_LF2;
          pA = A; pNum = num; pPabs = pabs; pAfterW = afterW; pHave = true;
          if (NT) {
            const int hq = 63 - (int)__builtin_clzll(NT);
            last = __builtin_amdgcn_readlane(anchorAfter, hq);
            afterWord = __builtin_amdgcn_readlane((int)(isRef && wlen > 1), hq) != 0;
          }
        }
      }
    } else {
      if (wHave) {
        // ---- write row t-4: its texts have arrived ----
        const bool st = (wA & TG3_ST) != 0, isRef = (wA & TG3_REF) != 0;
        const int wlen = (int)((wA >> 16) & 31u);
        const int sp = (wA & TG3_SP) ? 1 : 0;
        const int o = wO;
        if (st && !isRef) {
          if (wA & TG3_ESC) obuf[o & (TG2_RING - 1)] = (u8)(wA >> 8);
          else if (wA & TG3_LF2) { obuf[o & (TG2_RING - 1)] = (u8)TG_CR; obuf[(o + 1) & (TG2_RING - 1)] = (u8)wA; }
          else obuf[o & (TG2_RING - 1)] = (u8)wA;
        }
        if (isRef && sp) obuf[o & (TG2_RING - 1)] = (u8)' ';
        const u32 mxv = kz_wave_incl_max((u32)(isRef ? wlen : 0));
        const int mxAll = __builtin_amdgcn_readlane((int)mxv, 63);
        const int ob = o + sp;
        const int wl = isRef ? wlen : 0;
        if (wA & TG3_FLIP) wTa.x ^= 0x20u;
        const u32 tw[8] = {wTa.x, wTa.y, wTa.z, wTa.w, wTb.x, wTb.y, wTb.z, wTb.w};
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {                                       // (bytes past a word's end go to a spare byte of the lane: no branches)
          if (kk * 4 >= mxAll) break;
#pragma unroll
          for (int j = 0; j < 4; j++) obuf[(kk * 4 + j < wl) ? ((ob + kk * 4 + j) & (TG2_RING - 1)) : (TG2_RING + 4 * lane)] = (u8)(tw[kk] >> (8 * j));
        }
        done = wEnd;
        while (done - flushed >= 1024) {                                       // (the ring is this wave's alone)
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup", "local");
          *(uint4*)(dst + flushed + 16 * lane) = *(const uint4*)(obuf + ((flushed + 16 * lane) & (TG2_RING - 1)));
          flushed += 1024;
        }
        wHave = false;
      }
      const int r = t - 3;
      if (r >= 0 && r < R) {
        // ---- size row t-3 and fetch its texts ----
        const u32 A = qA[r & 1][lane], Bv = qB[r & 1][lane];
        const bool st = (A & TG3_ST) != 0, isRef = (A & TG3_REF) != 0;
        const int wlen = (int)((A >> 16) & 31u);
        const int sp = (A & TG3_SP) ? 1 : 0;
        int olen = 0;
        if (st) olen = isRef ? wlen + sp : ((A & TG3_LF2) ? 2 : 1);
        const u32 inc = kz_wave_incl_sum((u32)olen);
        const int total = (int)__builtin_amdgcn_readlane((int)inc, 63);
        if (at + total + 2 >= end) myBad = true;                               // may not fit: the host stage decides
        // (the ring: < 1024 bytes not stored yet + this row's, 2048 at most)
        else {
          wTa = tg_u32x4{0, 0, 0, 0}; wTb = tg_u32x4{0, 0, 0, 0};
          if (isRef) {
            if (A & TG3_STAT) wTa = *(const tg_u32x4*)(sT16 + 16 * Bv);
            else {
              wTa = *(const tg_u32x4_u*)(src + (int)Bv);                       // (slots have >= 4 KiB of slack behind the block)
              if (wlen > 16) wTb = *(const tg_u32x4_u*)(src + (int)Bv + 16);
            }
          }
          wA = A; wO = at + (int)inc - olen;
          at += total;
          wEnd = at;
          wHave = true;
        }
      }
    }
    if (myBad && lane == 0) sBad[t & 1] = 1;
#ifdef TG_PROF
    { const long long t_ = clock64(); tBody += t_ - tprev; tprev = t_; }
#endif
    tg_lds_barrier();
    if (sBad[t & 1]) { bad = true; break; }
  }
#ifdef TG_PROF
  if (lane == 0) { atomicAdd(&g_tgprof[wave], (unsigned long long)tBody); atomicAdd(&g_tgprof[3 + wave], (unsigned long long)tWait); }
#endif
  __syncthreads();
  if (wave == 2) {
    if (!bad) for (int p = flushed + lane; p < at; p += 64) dst[p] = obuf[p & (TG2_RING - 1)];
    if (lane == 0) G.outLen[b] = bad ? -1 : at;
  }
}

// copy the finished blocks back to their slots (16 bytes per lane: slots are 256-byte aligned)
__global__ __launch_bounds__(256) void k_text_copy_back(const u8* __restrict__ src, u8* __restrict__ dst, int64_t stride, const int32_t* __restrict__ len, const int32_t* __restrict__ cond) {
  const int b = blockIdx.y;
  if (!cond[b]) return;
  const int n16 = (len[b] + 15) >> 4;
  const uint4* s = (const uint4*)(src + (int64_t)b * stride);
  uint4* d = (uint4*)(dst + (int64_t)b * stride);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) d[i] = s[i];
}

// bytes of scratch per block taken (kz_api.hip sizes the arena with it)
size_t kz_text_gpu_scratch_per_block(int blockSize) {
  int l1 = 13; if (blockSize >= 8) { l1 = 31 - __builtin_clz((unsigned)(blockSize / 8)); l1 = l1 > 26 ? 26 : (l1 < 13 ? 13 : l1); }
  return ((size_t)4 << l1) + (size_t)TG_MAXDICT * sizeof(TgWord) + 1024;
}

// TEXT inverse of the blocks with take[b] != 0: reads bt.buf[cur] (lengths bt.h_len), leaves the result of the blocks it finished in
// the same slots (bt.h_len / bt.d_len updated) and sets done[b] = 1 for them; every other block is untouched (the host stage takes
// it).  variant1 = the stream's entropy coder asks for TextCodec1 (larger hash map).  form: 1 = a row of 64 coded bytes at a time
// (TextCodec2 blocks; the others are left to the host), 2 = the serial token walk (both variants).  Returns 0 or a negative error.
int kz_stage_text_inverse_gpu(kz_ctx* ctx, kz_batch& bt, int blockSize, int dstCap, bool variant1, const std::vector<int32_t>& take, std::vector<int32_t>& done, int form) {
  const int B = bt.B;
  done.assign(B, 0);
  std::vector<int32_t> ord(B, -1);
  int A = 0;
  for (int b = 0; b < B; b++) if (take[b] && bt.h_len[b] > 0) ord[b] = A++;
  if (A == 0) return 0;
  TextGpu G;
  G.logV1 = 13; if (blockSize >= 8) G.logV1 = std::max(std::min(31 - __builtin_clz((unsigned)(blockSize / 8)), 26), 13);      // TextCodec.java:561-575
  G.logV2 = 13; if (blockSize >= 32) G.logV2 = std::max(std::min(31 - __builtin_clz((unsigned)(blockSize / 32)), 24), 13);    // :1068-1081
  G.variantCap = variant1 ? 1 : 0;
  G.slotsPer = (int64_t)1 << (variant1 ? G.logV1 : G.logV2);
  G.llog = 13; if (dstCap >= 1024) G.llog = std::max(std::min(31 - __builtin_clz((unsigned)(dstCap / 128)), 18), 13);          // :578-582
  G.dstCap = dstCap;
  static std::vector<uint32_t> hHash, hLenIdx; static std::vector<int32_t> hPos; static std::vector<uint8_t> hText, hDelim; static int hCount = -1;
  static std::once_flag once;
  std::call_once(once, [] { kz_text_static_tables(hHash, hPos, hLenIdx, hText, hDelim, &hCount); });
  if (hCount + 2 > TG_SMAX || (int)hText.size() > TG_STEXT) return 0;      // (cannot happen: the dictionary is a fixed table)
  const size_t mark = ctx->arenaTop;
  kz_arena_guard arenaGuard{ctx, mark};
  const int NW = hCount + 2;
  u32* dHash = (u32*)kz_arena_alloc(ctx, (size_t)NW * 4);
  int32_t* dPos = (int32_t*)kz_arena_alloc(ctx, (size_t)NW * 4);
  u32* dLenIdx = (u32*)kz_arena_alloc(ctx, (size_t)NW * 4);
  u8* dText = (u8*)kz_arena_alloc(ctx, hText.size() + 64);
  u8* dDelim = (u8*)kz_arena_alloc(ctx, 256);
  int32_t* dOrd = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  int32_t* dOut = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  int32_t* dCond = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  G.slots = (int32_t*)kz_arena_alloc(ctx, (size_t)A * (size_t)G.slotsPer * 4);
  G.words = (TgWord*)kz_arena_alloc(ctx, (size_t)A * TG_MAXDICT * sizeof(TgWord));
  if (!dHash || !dPos || !dLenIdx || !dText || !dDelim || !dOrd || !dOut || !dCond || !G.slots || !G.words) { ctx->arenaTop = mark; return 0; }   // no room: the host stage takes them all
  hipStream_t st = ctx->stream;
  KZ_HIP(hipMemcpyAsync(dHash, hHash.data(), (size_t)NW * 4, hipMemcpyHostToDevice, st));
  KZ_HIP(hipMemcpyAsync(dPos, hPos.data(), (size_t)NW * 4, hipMemcpyHostToDevice, st));
  KZ_HIP(hipMemcpyAsync(dLenIdx, hLenIdx.data(), (size_t)NW * 4, hipMemcpyHostToDevice, st));
  KZ_HIP(hipMemcpyAsync(dText, hText.data(), hText.size(), hipMemcpyHostToDevice, st));
  KZ_HIP(hipMemcpyAsync(dDelim, hDelim.data(), 256, hipMemcpyHostToDevice, st));
  KZ_HIP(hipMemcpyAsync(dOrd, ord.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
  KZ_HIP(hipMemsetAsync(dOut, 0xFF, (size_t)B * 4, st));
  KZ_HIP(hipMemcpyAsync(bt.d_len, bt.h_len.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
  G.sHash = dHash; G.sPos = dPos; G.sLenIdx = dLenIdx; G.sText = dText; G.delim = dDelim; G.sCount = hCount; G.sTextLen = (int)hText.size(); G.ord = dOrd; G.outLen = dOut;
  if (form == 2) { KZ_LAUNCH(ctx, KID_TEXT_INV, k_text_inv, dim3(B), dim3(64), bt.buf[bt.cur], bt.buf[bt.cur ^ 1], bt.stride, bt.d_len, G, B); }
  else if (form == 3) { KZ_LAUNCH(ctx, KID_TEXT_INV, k_text_inv2, dim3(B), dim3(64), bt.buf[bt.cur], bt.buf[bt.cur ^ 1], bt.stride, bt.d_len, G, B); }
  else { KZ_LAUNCH(ctx, KID_TEXT_INV, k_text_inv3, dim3(B), dim3(192), bt.buf[bt.cur], bt.buf[bt.cur ^ 1], bt.stride, bt.d_len, G, B); }
  std::vector<int32_t> outLen(B);
  KZ_HIP(hipMemcpyAsync(outLen.data(), dOut, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  KZ_HIP(kz_stream_sync(ctx, st));
  std::vector<int32_t> cond(B, 0), newLen(bt.h_len);
  int any = 0;
  for (int b = 0; b < B; b++) if (ord[b] >= 0 && outLen[b] >= 0) { cond[b] = 1; newLen[b] = outLen[b]; done[b] = 1; any = 1; }
  if (ctx->sw.textGpuTrace) { int nd = 0; for (int b = 0; b < B; b++) nd += done[b]; fprintf(stderr, "[textgpu] took %d blocks, finished %d\n", A, nd); }
  if (any) {
    // the finished blocks go back to the slots the rest of the decoder reads (the untouched ones are still there)
    KZ_HIP(hipMemcpyAsync(dCond, cond.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    KZ_HIP(hipMemcpyAsync(dOut, newLen.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_text_copy_back, dim3(64, B), dim3(256), 0, st, bt.buf[bt.cur ^ 1], bt.buf[bt.cur], bt.stride, dOut, dCond);
    for (int b = 0; b < B; b++) bt.h_len[b] = newLen[b];
    KZ_HIP(hipMemcpyAsync(bt.d_len, bt.h_len.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    KZ_HIP(kz_stream_sync(ctx, st));
  }
#ifdef TG_PROF
  { unsigned long long h[8]; KZ_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tgprof), sizeof h));
    fprintf(stderr, "[textgpu-prof] clocks per block (row form, one wave: walk anchors words records sizes+text write flush loop; three waves: busy 0 1 2, waiting 0 1 2): %llu %llu %llu %llu %llu %llu %llu %llu\n",
            h[0] / A, h[1] / A, h[2] / A, h[3] / A, h[4] / A, h[5] / A, h[6] / A, h[7] / A);
    memset(h, 0, sizeof h); KZ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tgprof), h, sizeof h)); }
#endif
  KZ_HIP(hipGetLastError());
  ctx->arenaTop = mark;
  return 0;
}

// ==== UTF inverse on the device (UTFCodec.java:224-330; host form: kz_text.hip utf_inverse) ==========================================
// The coded block: start & 3, adjust & 3, the symbol count, 3 bytes per symbol (the code points by rank), `start` raw bytes, then
// aliases of one byte (rank < 128) or two (0x80 | rank & 0x7F, rank >> 7) up to `body` = n - 4 + adjust, then the tail bytes as they
// are.  A byte >= 0x80 takes the byte behind it whatever that is, so whether a byte starts an alias is the PARITY of the run of
// bytes >= 0x80 right in front of it: nothing serial.  Four small kernels: the table of code points (`k_utf_map`), output bytes per
// tile of 2 048 coded bytes (`k_utf_tiles<false>`), the tile offsets and the block's verdict (`k_utf_scan`), the bytes
// (`k_utf_tiles<true>`).  A block the reference refuses (a rank past the table, a malformed entry, an alias cut by the end, an
// output that does not fit) is left alone with outLen = -1: the host stage gives it the reference's verdict.
#define UG_MAXSYM 32768
#define UG_TILE 2048
struct UtfGpu {
  const int32_t* ord;   // [B] index among the blocks taken, -1 = leave alone
  int32_t* outLen;      // [B]
  int32_t* fail;        // [A]
  u32* map;             // [A][UG_MAXSYM] the UTF-8 bytes of a rank, little endian in a word (the first byte tells how many)
  int32_t* tileOut;     // [A][maxTiles] output bytes per tile, then their exclusive prefix
  int maxTiles, dstCap;
};
__device__ __forceinline__ int ug_len(u32 le) { const u32 c = le & 0xFFu; return c < 0x80u ? 1 : (c < 0xE0u ? 2 : (c < 0xF0u ? 3 : 4)); }
__global__ __launch_bounds__(256) void k_utf_map(const u8* __restrict__ srcAll, int64_t stride, const int32_t* __restrict__ d_len, UtfGpu G, int B) {
  const int b = blockIdx.y;
  const int a = G.ord[b];
  if (a < 0) return;
  const int n = d_len[b];
  const u8* src = srcAll + (int64_t)b * stride;
  if (n < 4) { if (blockIdx.x == 0 && threadIdx.x == 0) G.fail[a] = 1; return; }
  const int start = src[0] & 3, adjust = src[1] & 3;
  const int nsym = ((int)src[2] << 8) | (int)src[3];
  const int i0 = 4 + 3 * nsym + start, body = n - 4 + adjust;
  if (nsym == 0 || nsym >= UG_MAXSYM || 3 * nsym >= n || i0 > n || i0 > body) { if (blockIdx.x == 0 && threadIdx.x == 0) G.fail[a] = 1; return; }
  for (int r = blockIdx.x * 256 + (int)threadIdx.x; r < nsym; r += gridDim.x * 256) {
    const u8* q = src + 4 + 3 * r;
    const u32 key = ((u32)q[0] << 16) | ((u32)q[1] << 8) | (u32)q[2];
    u32 le = 0; bool ok = true;
    switch (key >> 19) {                                                       // unpackV1 (:514-548)
      case 0: le = key; break;
      case 1: le = ((key & 0xFFu) << 8) | ((key >> 8) & 0xFFu); break;
      case 2: le = (((key >> 12) & 0x0Fu) | 0xE0u) | ((((key >> 6) & 0x3Fu) | 0x80u) << 8) | (((key & 0x3Fu) | 0x80u) << 16); break;
      case 4: case 5: case 6: case 7:
        le = (((key >> 18) & 0x07u) | 0xF0u) | ((((key >> 12) & 0x3Fu) | 0x80u) << 8) | ((((key >> 6) & 0x3Fu) | 0x80u) << 16) | (((key & 0x3Fu) | 0x80u) << 24); break;
      default: ok = false; break;
    }
    // (the count of bytes is read back from the first one: a one-byte entry >= 0x80 or a two-byte entry outside 0xC0-0xDF cannot be told apart later)
    const int want = (key >> 19) == 0 ? 1 : ((key >> 19) == 1 ? 2 : ((key >> 19) == 2 ? 3 : 4));
    if (!ok || ug_len(le) != want) G.fail[a] = 1;
    G.map[(int64_t)a * UG_MAXSYM + r] = le;
  }
}
// one tile: the aliases that start in [i0 + t * UG_TILE, + UG_TILE) below `body`; WRITE = false: count their output bytes
template <bool WRITE>
__global__ __launch_bounds__(256) void k_utf_tiles(const u8* __restrict__ srcAll, u8* __restrict__ dstAll, int64_t stride, const int32_t* __restrict__ d_len, UtfGpu G, int B) {
  const int b = blockIdx.y;
  const int a = G.ord[b];
  if (a < 0) return;
  if (__syncthreads_or(G.fail[a])) return;                                     // (one answer for the workgroup: other tiles may be setting it)
  const int n = d_len[b];
  const u8* src = srcAll + (int64_t)b * stride;
  const int start = src[0] & 3, adjust = src[1] & 3;
  const int nsym = ((int)src[2] << 8) | (int)src[3];
  const int i0 = 4 + 3 * nsym + start, body = n - 4 + adjust;
  const int t = blockIdx.x;
  const int t0 = i0 + t * UG_TILE;
  if (t0 >= body) return;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u32* map = G.map + (int64_t)a * UG_MAXSYM;
  u8* dst = dstAll + (int64_t)b * stride;
  __shared__ int wsum[4];
  __shared__ int sBase;
  int base = WRITE ? start + G.tileOut[(int64_t)a * G.maxTiles + t] : 0;     // output position of the tile's first alias
  int total = 0;
  bool bad = false;
  const uint64_t ltm = kz_lanemask_lt();
  for (int s0 = t0; s0 < min(t0 + UG_TILE, body); s0 += 256) {
    const int p = s0 + tid, wbase = s0 + wave * 64;
    const u32 c = p < n ? (u32)src[p] : 0u;
    const u32 c1 = p + 1 < n ? (u32)src[p + 1] : 0u;
    const bool in = p < body;
    const uint64_t hiM = kz_ballot(in && c >= 0x80u);
    // bytes >= 0x80 right in front of the wave's first byte (not past i0): their count's parity carries in
    int carry = 0;
    for (int k = 0;; k += 64) {
      const int q = wbase - 1 - k - lane;
      const uint64_t m = kz_ballot(q >= i0 && (u32)src[q < 0 ? 0 : q] >= 0x80u);
      const int cnt = ~m ? (int)__builtin_ctzll(~m) : 64;
      carry += cnt;
      if (cnt < 64) break;
    }
    const uint64_t z = ~hiM & ltm;                                             // bytes < 0x80 below this lane
    const int run = z ? lane - 1 - (63 - (int)__builtin_clzll(z)) : lane + carry;
    const bool st = in && (run & 1) == 0;
    int len = 0; u32 le = 0;
    if (st) {
      int alias = (int)c;
      if (c >= 0x80u) { if (p + 1 >= body) bad = true; alias = (int)(c1 << 7) + (int)(c & 0x7Fu); }   // (p + 1 == body: the alias runs into the tail, refused at :296-297)
      if (alias >= nsym) bad = true;
      else { le = map[alias]; len = ug_len(le); }
    }
    // exclusive scan of len over the 256 threads
    const u32 inc = kz_wave_incl_sum((u32)len);
    if (lane == 63) wsum[wave] = (int)inc;
    __syncthreads();
    int before = 0, all = 0;
    for (int w = 0; w < 4; w++) { const int v = wsum[w]; if (w < wave) before += v; all += v; }
    if (WRITE && len) {
      u8* o = dst + base + total + before + (int)inc - len;
      o[0] = (u8)le;
      if (len > 1) o[1] = (u8)(le >> 8);
      if (len > 2) o[2] = (u8)(le >> 16);
      if (len > 3) o[3] = (u8)(le >> 24);
    }
    total += all;
    __syncthreads();
  }
  if (!WRITE) {
    if (__syncthreads_or(bad ? 1 : 0)) { if (tid == 0) G.fail[a] = 1; }
    if (tid == 0) G.tileOut[(int64_t)a * G.maxTiles + t] = total;
  }
}
// per block: the tiles' offsets, the verdict, the raw bytes in front of and behind the aliases
__global__ __launch_bounds__(256) void k_utf_scan(const u8* __restrict__ srcAll, u8* __restrict__ dstAll, int64_t stride, const int32_t* __restrict__ d_len, UtfGpu G, int B) {
  const int b = blockIdx.x;
  const int a = G.ord[b];
  if (a < 0) return;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (G.fail[a]) { if (tid == 0) G.outLen[b] = -1; return; }
  const int n = d_len[b];
  const u8* src = srcAll + (int64_t)b * stride;
  u8* dst = dstAll + (int64_t)b * stride;
  const int start = src[0] & 3, adjust = src[1] & 3;
  const int nsym = ((int)src[2] << 8) | (int)src[3];
  const int i0 = 4 + 3 * nsym + start, body = n - 4 + adjust;
  const int tiles = (body - i0 + UG_TILE - 1) / UG_TILE;
  int32_t* to = G.tileOut + (int64_t)a * G.maxTiles;
  __shared__ int wsum[4];
  int run = 0;
  for (int t0 = 0; t0 < tiles; t0 += 256) {
    const int t = t0 + tid;
    const int v = t < tiles ? to[t] : 0;
    const u32 inc = kz_wave_incl_sum((u32)v);
    if (lane == 63) wsum[wave] = (int)inc;
    __syncthreads();
    int before = 0, all = 0;
    for (int w = 0; w < 4; w++) { const int x = wsum[w]; if (w < wave) before += x; all += x; }
    if (t < tiles) to[t] = run + before + (int)inc - v;
    run += all;
    __syncthreads();
  }
  const int at = start + run, tail = n - body;
  const int end = G.dstCap - 4;
  if (at >= end - tail) { if (tid == 0) { G.outLen[b] = -1; G.fail[a] = 1; } return; }     // :290-291 (and the loop's own `at < end`)
  if (tid < start) dst[tid] = src[4 + 3 * nsym + tid];
  if (tid < tail) dst[at + tid] = src[body + tid];
  if (tid == 0) G.outLen[b] = at + tail;
}

size_t kz_utf_gpu_scratch_per_block(int maxLen) { return (size_t)UG_MAXSYM * 4 + (size_t)(maxLen / UG_TILE + 2) * 4 + 256; }

// UTF inverse of the blocks with take[b] != 0, like kz_stage_text_inverse_gpu: finished blocks are left in their slots (lengths
// updated) with done[b] = 1, every other block is untouched.
int kz_stage_utf_inverse_gpu(kz_ctx* ctx, kz_batch& bt, int dstCap, const std::vector<int32_t>& take, std::vector<int32_t>& done) {
  const int B = bt.B;
  done.assign(B, 0);
  std::vector<int32_t> ord(B, -1);
  int A = 0, maxLen = 0;
  for (int b = 0; b < B; b++) if (take[b] && bt.h_len[b] > 0) { ord[b] = A++; maxLen = std::max(maxLen, bt.h_len[b]); }
  if (A == 0) return 0;
  UtfGpu G;
  G.maxTiles = maxLen / UG_TILE + 2; G.dstCap = dstCap;
  const size_t mark = ctx->arenaTop;
  kz_arena_guard arenaGuard{ctx, mark};
  int32_t* dOrd = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  int32_t* dOut = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  int32_t* dCond = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  G.fail = (int32_t*)kz_arena_alloc(ctx, (size_t)A * 4);
  G.map = (u32*)kz_arena_alloc(ctx, (size_t)A * UG_MAXSYM * 4);
  G.tileOut = (int32_t*)kz_arena_alloc(ctx, (size_t)A * (size_t)G.maxTiles * 4);
  if (!dOrd || !dOut || !dCond || !G.fail || !G.map || !G.tileOut) { ctx->arenaTop = mark; return 0; }     // no room: the host stage takes them all
  G.ord = dOrd; G.outLen = dOut;
  hipStream_t st = ctx->stream;
  KZ_HIP(hipMemcpyAsync(dOrd, ord.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
  KZ_HIP(hipMemsetAsync(dOut, 0xFF, (size_t)B * 4, st));
  KZ_HIP(hipMemsetAsync(G.fail, 0, (size_t)A * 4, st));
  const u8* src = bt.buf[bt.cur]; u8* dst = bt.buf[bt.cur ^ 1];
  KZ_LAUNCH(ctx, KID_UTF_INV, k_utf_map, dim3(16, B), dim3(256), src, bt.stride, bt.d_len, G, B);
  KZ_LAUNCH(ctx, KID_UTF_INV, k_utf_tiles<false>, dim3(G.maxTiles, B), dim3(256), src, dst, bt.stride, bt.d_len, G, B);
  KZ_LAUNCH(ctx, KID_UTF_INV, k_utf_scan, dim3(B), dim3(256), src, dst, bt.stride, bt.d_len, G, B);
  KZ_LAUNCH(ctx, KID_UTF_INV, k_utf_tiles<true>, dim3(G.maxTiles, B), dim3(256), src, dst, bt.stride, bt.d_len, G, B);
  std::vector<int32_t> outLen(B);
  KZ_HIP(hipMemcpyAsync(outLen.data(), dOut, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  KZ_HIP(kz_stream_sync(ctx, st));
  std::vector<int32_t> cond(B, 0), newLen(bt.h_len);
  int any = 0, nd = 0;
  for (int b = 0; b < B; b++) if (ord[b] >= 0 && outLen[b] >= 0) { cond[b] = 1; newLen[b] = outLen[b]; done[b] = 1; any = 1; nd++; }
  if (ctx->sw.textGpuTrace) fprintf(stderr, "[utfgpu] took %d blocks, finished %d\n", A, nd);
  if (any) {
    KZ_HIP(hipMemcpyAsync(dCond, cond.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    KZ_HIP(hipMemcpyAsync(dOut, newLen.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_text_copy_back, dim3(64, B), dim3(256), 0, st, bt.buf[bt.cur ^ 1], bt.buf[bt.cur], bt.stride, dOut, dCond);
    for (int b = 0; b < B; b++) bt.h_len[b] = newLen[b];
    KZ_HIP(hipMemcpyAsync(bt.d_len, bt.h_len.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    KZ_HIP(kz_stream_sync(ctx, st));
  }
  KZ_HIP(hipGetLastError());
  ctx->arenaTop = mark;
  return 0;
}
