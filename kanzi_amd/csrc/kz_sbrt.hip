// kz_sbrt.hip -- Sort-By-Rank transform (RANK = SBR(1/2), MTF = SBR(0), TIMESTAMP) on gfx950.
//
// Replaces K/transform/SBRT.java:87-151 (forward) and :154-214 (inverse).
//
// The reference keeps a 256-entry list ordered by q[] (descending), most recently moved first among
// equal q, never-seen symbols at the bottom in symbol order.  That order is a pure function of, per
// symbol, its last two occurrence positions:  key(s) = (q_s << 32) | (p_s + 256)  for a seen symbol,
// 255 - s for a never-seen one, and   rank(c) = #{ s : key(s) > key(c) }.
//
// forward (parallel):  1) per 8 KiB tile, last two occurrences of every symbol  2) per block, an
//   exclusive "last-two" scan over tiles (thread = symbol)  3) every tile replays independently:
//   one wave64 per tile holds the 256 keys in VGPRs (symbol s = element s>>6, lane s&63);
//   a rank is 4 v_cmp_gt_u64 ballots + s_bcnt1; runs of equal bytes are skipped from the third byte on.
// inverse (serial per block by nature: the list state depends on every decoded symbol): one wave
//   per block; the list lives by POSITION in VGPRs (row k = positions 64k..64k+63, or interleaved for rows of
//   high ranks), a move is a masked DPP shift; blocks of the batch decode concurrently, placed on the SIMDs
//   by cost (kz_place_blocks).  See "inverse v6" below.
#include "kz_device.h"
#include "kz_internal.h"
#include <algorithm>
#include <vector>
#include <stdlib.h>

typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint8_t u8;

#define SB_TS 8192               // bytes per tile

__device__ __forceinline__ u64 kz_readlane64(u64 v, int l) {
  u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, l);
  u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), l);
  return ((u64)hi << 32) | lo;
}

// ---- forward 1/3: last two occurrences per symbol in each tile -------------------------------
// tab[b][t][s] = (p1,p2) absolute positions, -1 = none.  64 bytes per step: one ballot match-any groups
// the lanes holding the same symbol; the highest lane of each group (distinct symbols -> distinct LDS
// slots, no atomics) merges the group's two highest positions into the running table.
__global__ __launch_bounds__(64) void k_sbrt_last2(const u8* __restrict__ src, int64_t stride, const int32_t* __restrict__ d_len,
                                                    int2* __restrict__ tab, int T) {
  const int b = blockIdx.y, t = blockIdx.x;
  const int n = d_len[b];
  const int start = t * SB_TS;
  if (start >= n) return;
  const int end = min(n, start + SB_TS);
  const u8* s = src + (int64_t)b * stride;
  const int lane = kz_lane();
  __shared__ int2 last[256];
  for (int i = lane; i < 256; i += 64) last[i] = make_int2(-1, -1);
  __syncthreads();
  for (int row = start; row < end; row += 64) {
    const int i = row + lane;
    const bool valid = i < end;
    const u32 c = valid ? (u32)s[i] : 0u;
    const uint64_t peers = kz_match8(c, valid);
    if (valid && (peers >> lane) == 1ULL) {                         // highest lane of its group
      const uint64_t rest = peers & ~(1ULL << lane);
      int2 v = last[c];
      if (rest) { v.y = row + (63 - (int)__builtin_clzll(rest)); } else { v.y = v.x; }
      v.x = i;
      last[c] = v;
    }
  }
  __syncthreads();
  int2* o = tab + ((int64_t)b * T + t) * 256;
#pragma unroll
  for (int q = 0; q < 4; q++) o[q * 64 + lane] = last[q * 64 + lane];
}

// ---- forward 2/3: exclusive scan over tiles (thread = symbol) --------------------------------
__global__ __launch_bounds__(256) void k_sbrt_scan(const int32_t* __restrict__ d_len, int2* __restrict__ tab, int T) {
  const int b = blockIdx.x;
  const int n = d_len[b];
  const int tiles = (n + SB_TS - 1) / SB_TS;
  int a1 = -1, a2 = -1;
  int2* base = tab + (int64_t)b * T * 256 + threadIdx.x;
  for (int t = 0; t < tiles; t++) {
    int2 v = base[(int64_t)t * 256];
    base[(int64_t)t * 256] = make_int2(a1, a2);
    if (v.x >= 0) { if (v.y >= 0) { a1 = v.x; a2 = v.y; } else { a2 = a1; a1 = v.x; } }
  }
}

// mode 4 = SRT's move-to-front (K/transform/SRT.java:131-150): plain MTF whose initial list is the order
// of first appearance, i.e. a never-seen symbol ranks after every seen one and never-seen symbols do not
// count each other: key 0.
__device__ __forceinline__ u64 kz_sbrt_key(int mode, int a1, int a2, int sym) {
  if (a1 < 0) return (mode == 4) ? 0ULL : (u64)(255 - sym);
  const u32 pprev = a2 < 0 ? 0u : (u32)a2;          // p[] starts at 0 (SBRT.java:113-118)
  u32 q;
  if (mode == 2) q = ((u32)a1 + pprev) >> 1;         // RANK: (i + p[c]) >> 1
  else if (mode == 1 || mode == 4) q = (u32)a1;      // MTF : i
  else q = pprev;                                    // TIMESTAMP: p[c]
  return ((u64)q << 32) | (u64)((u32)a1 + 256u);
}

typedef u64 kz_u64x8 __attribute__((ext_vector_type(8)));

// ---- forward 3/3: replay one tile per wave ----------------------------------------------------
// Keys per symbol (symbol s = lane s&63, element s>>6 of K; 8 elements so that uniform-index accesses stay
// VGPR-indexed moves).  A rank is 4 v_cmp_gt_u64 ballots + s_bcnt1.  From the third equal byte in a row on,
// the rank is 0 in every mode (after two consecutive occurrences the symbol is at the front) and only the
// symbol's own (q,p) change, in closed form: those positions are skipped with a ballot mask
// R[j] = byte[j]==byte[j-1]==byte[j-2] and the key is repaired before the next ranked symbol (after BWT
// most bytes sit in such runs).
#define KZ_SBRT_FIX_RUN(sym, plv)                                                              \
  { const u32 pl = (u32)(plv), pp = pl - 1u;                                                   \
    const u32 fq = (MODE == 2) ? ((pl + pp) >> 1) : ((MODE == 1 || MODE == 4) ? pl : pp);      \
    const int fs = (int)((sym) >> 6);                                                          \
    const u64 ok = K[fs];                                                                      \
    K[fs] = (lane == (int)((sym) & 63u)) ? (((u64)fq << 32) | (u64)(pl + 256u)) : ok; }

template <int MODE>
__device__ __forceinline__ void sbrt_replay_by_symbol(const u8* __restrict__ src, u8* __restrict__ dst, int64_t stride,
                                                      const int32_t* __restrict__ d_len, const int2* __restrict__ tab, int T) {
  const int b = blockIdx.y, t = blockIdx.x;
  const int n = d_len[b];
  const int start = t * SB_TS;
  if (start >= n) return;
  const int end = min(n, start + SB_TS);
  const u8* s = src + (int64_t)b * stride;
  u8* d = dst + (int64_t)b * stride;
  const int lane = kz_lane();
  const int2* pre = tab + ((int64_t)b * T + t) * 256;
  const int2 v0 = pre[lane], v1 = pre[64 + lane], v2 = pre[128 + lane], v3 = pre[192 + lane];
  kz_u64x8 K = (kz_u64x8)(0ULL);
  K[0] = kz_sbrt_key(MODE, v0.x, v0.y, lane);
  K[1] = kz_sbrt_key(MODE, v1.x, v1.y, 64 + lane);
  K[2] = kz_sbrt_key(MODE, v2.x, v2.y, 128 + lane);
  K[3] = kz_sbrt_key(MODE, v3.x, v3.y, 192 + lane);
  // bytes before the tile (run detection across the tile boundary)
  u32 cp = (start >= 1) ? (u32)s[start - 1] : 0x100u;            // symbol at the previous position (0x100: none)
  uint64_t carryE = (start >= 2 && s[start - 1] == s[start - 2]) ? 1ULL : 0ULL;
  u32 cur = (start + lane < end) ? (u32)s[start + lane] : 0u;
  for (int row = start; row < end; row += 64) {
    const int cnt = min(64, end - row);
    const int nrow = row + 64;
    const u32 nxt = (nrow + lane < end) ? (u32)s[nrow + lane] : 0u;  // prefetch the next row
    u32 prevb = (u32)__builtin_amdgcn_update_dpp(0, (int)cur, 0x138 /*wave_shr:1*/, 0xF, 0xF, true);
    if (lane == 0) prevb = cp;
    const uint64_t valid = (cnt == 64) ? ~0ULL : ((1ULL << cnt) - 1ULL);
    const uint64_t E = kz_ballot(cur == prevb) & valid;
    const uint64_t R = E & ((E << 1) | carryE);
    carryE = E >> 63;
    uint64_t N = valid & ~R;
    uint64_t F = N & (R << 1);                                        // ranked positions right behind a skipped stretch: repair the key first
    int jf;
    asm("s_ff1_i32_b64 %0, %1" : "=s"(jf) : "s"(F));                   // next of them (-1: none)
    u32 outv = 0;
    while (N) {
      const int j = (int)__builtin_ctzll(N);
      asm("s_bitset0_b64 %0, %1" : "+s"(N) : "s"(j));                  // (N &= N - 1 costs three scalar instructions)
      if (__builtin_expect(j == jf, 0)) {
        KZ_SBRT_FIX_RUN(cp, row + j - 1)
        asm("s_bitset0_b64 %0, %1" : "+s"(F) : "s"(j));
        asm("s_ff1_i32_b64 %0, %1" : "=s"(jf) : "s"(F));
      }
      const u32 c = (u32)__builtin_amdgcn_readlane((int)cur, j);
      const int cl = (int)(c & 63u), cs = (int)(c >> 6);
      const u64 ok = K[cs];
      const u64 kc = kz_readlane64(ok, cl);
      const u32 cntv = (u32)(__builtin_popcountll(kz_ballot(K[0] > kc)) + __builtin_popcountll(kz_ballot(K[1] > kc)) +
                             __builtin_popcountll(kz_ballot(K[2] > kc)) + __builtin_popcountll(kz_ballot(K[3] > kc)));
      const u32 lo = (u32)kc;
      const u32 pc = max(lo, 256u) - 256u;
      const u32 iv = (u32)(row + j);
      const u32 qc = (MODE == 2) ? ((iv + pc) >> 1) : ((MODE == 1 || MODE == 4) ? iv : pc);
      const u64 nk = ((u64)qc << 32) | (u64)(iv + 256u);
      u64 upd = (lane == cl) ? nk : ok;
      asm volatile("" : "+v"(upd));                                 // one 64-bit value: one indexed store (the halves took a gpr_idx region each)
      K[cs] = upd;
      outv = (lane == j) ? (u32)cntv : outv;                        // outv[lane j] = rank (no inline v_writelane: its lane select needs m0, which inline asm may not clobber)
      cp = c;
    }
    if ((R >> (cnt - 1)) & 1ULL) KZ_SBRT_FIX_RUN(cp, row + cnt - 1)    // the row ends inside a skipped stretch
    if (lane < cnt) d[row + lane] = (u8)outv;
    cur = nxt;
  }
}

template <int MODE>
__global__ __launch_bounds__(64) void k_sbrt_replay(const u8* __restrict__ src, u8* __restrict__ dst, int64_t stride,
                                                     const int32_t* __restrict__ d_len, const int2* __restrict__ tab, int T) {
  sbrt_replay_by_symbol<MODE>(src, dst, stride, d_len, tab, T);
}

// ---- forward 3/3, round 6: the same replay with the list BY POSITION on 64-bit keys (kz_sbrt_f64.h: hi = 0x40000000 | q,
// lo = p << 9 | touched << 8 | 255 - symbol; register pair k = positions 64 k .. 64 k + 63).  The rank of a symbol IS its position:
// one byte compare + ballot finds it (row 0 first: after a BWT most symbols sit there), and the update is the inverse's step -- the
// entry takes its new key x and every position above it takes max(min(x, left neighbour), own): no four 64-bit compares + bit
// counts per symbol, no indexed register file.  ~28 instructions per ranked symbol against ~45.  The tile's first list comes from
// the symbols' last two occurrences (k_sbrt_last2 / k_sbrt_scan) sorted by key: 256 keys ranked against each other once per 8 KiB.
// Blocks up to 2^23 bytes (p < 2^23); MODE 1 MTF, 2 RANK, 3 TIMESTAMP (SRT's variant, mode 4, keeps k_sbrt_replay).
__device__ __forceinline__ u64 kzr_min(u64 a, u64 b) { u64 r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ u64 kzr_max(u64 a, u64 b) { u64 r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// the left neighbour's key; lane 0 receives `first`
__device__ __forceinline__ u64 kzr_shr(u64 v, u64 first) {
  const u32 lo = (u32)__builtin_amdgcn_update_dpp((int)(u32)first, (int)(u32)v, 0x138 /*wave_shr:1*/, 0xF, 0xF, false);
  const u32 hi = (u32)__builtin_amdgcn_update_dpp((int)(u32)(first >> 32), (int)(u32)(v >> 32), 0x138, 0xF, 0xF, false);
  return ((u64)hi << 32) | lo;
}
template <int MODE>
__device__ __forceinline__ u64 kzr_key(u32 i, u32 p, u32 sym255) {     // the key of a symbol accessed at i whose last access was p
  const u32 q = (MODE == 2) ? ((i + p) >> 1) : ((MODE == 1) ? i : p);
  return ((u64)(0x40000000u | q) << 32) | (u64)((i << 9) | 0x100u | sym255);
}
#define KZR_FIX_RUN(plv)                                                                        \
  { const u32 pl = (u32)(plv), pp = pl - 1u;                                                   \
    const u32 fq = (MODE == 2) ? ((pl + pp) >> 1) : ((MODE == 1) ? pl : pp);                   \
    const u64 nk = ((u64)(0x40000000u | fq) << 32) | (u64)((pl << 9) | 0x100u | ((u32)R0 & 0xFFu)); \
    R0 = (lane == 0) ? nk : R0; }

template <int MODE>
__global__ __launch_bounds__(64) void k_sbrt_replay_keyed(const u8* __restrict__ src, u8* __restrict__ dst, int64_t stride,
                                                           const int32_t* __restrict__ d_len, const int2* __restrict__ tab, int T) {
  const int b = blockIdx.y, t = blockIdx.x;
  const int n = d_len[b];
  const int start = t * SB_TS;
  if (start >= n) return;
  const int end = min(n, start + SB_TS);
  const u8* s = src + (int64_t)b * stride;
  u8* d = dst + (int64_t)b * stride;
  const int lane = kz_lane();
  const int2* pre = tab + ((int64_t)b * T + t) * 256;
  __shared__ u64 sk[256];
  {
    // Tiles of many different bytes (incompressible data: three ranks of four lie beyond position 63, where this form pays a
    // row-spanning step of ~60 instructions) take the by-symbol replay, whose rank costs the same at every depth: the distinct
    // byte values among the tile's first 256 bytes decide (uniform bytes ~160, text and skewed bytes under 70).
    u32* flags = (u32*)sk;
#pragma unroll
    for (int k = 0; k < 4; k++) flags[64 * k + lane] = 0;
    __syncthreads();
    for (int k = 0; k < 4; k++) { const int i = start + 64 * k + lane; if (i < end) flags[s[i]] = 1; }
    __syncthreads();
    int distinct = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) distinct += __builtin_popcountll(kz_ballot(flags[64 * k + lane] != 0));
    __syncthreads();
    if (distinct > 96) { sbrt_replay_by_symbol<MODE>(src, dst, stride, d_len, tab, T); return; }
  }
  // keys by symbol (symbol = lane + 64 k), then each key's rank among the 256 = its position in the list
  u64 K[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int2 v = pre[64 * k + lane];
    const u32 sym255 = 255u - (u32)(64 * k + lane);
    if (v.x < 0) K[k] = ((u64)0x40000000u << 32) | sym255;                                       // never accessed: q = p = 0, by ascending symbol
    else {
      const u32 a1 = (u32)v.x, pp = v.y < 0 ? 0u : (u32)v.y;
      const u32 q = (MODE == 2) ? ((a1 + pp) >> 1) : ((MODE == 1) ? a1 : pp);
      K[k] = ((u64)(0x40000000u | q) << 32) | (u64)((a1 << 9) | 0x100u | sym255);
    }
    sk[64 * k + lane] = K[k];
  }
  __syncthreads();
  u32 rk[4] = {0, 0, 0, 0};
  for (int j = 0; j < 256; j++) {
    const u64 kj = sk[j];                                                                        // (uniform address: a broadcast read)
#pragma unroll
    for (int k = 0; k < 4; k++) rk[k] += kj > K[k] ? 1u : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; k++) sk[rk[k]] = K[k];
  __syncthreads();
  u64 R0 = sk[lane], R1 = sk[64 + lane], R2 = sk[128 + lane], R3 = sk[192 + lane];
  const u64 INF = 0x7FF0000000000000ULL;
  // The replay is bound by the scalar unit and the VALU alike (SQ counters: ~11 instructions of each per input byte, both near
  // their issue limits), so what a step needs from the symbol's position is prepared per ROW in VGPRs and fetched with one
  // v_readlane each: xl = the new key's low word (i << 9 | touched | 255 - symbol), ivb = i + 2^31 (RANK: (p + ivb) >> 1 is
  // 0x40000000 | (i + p) >> 1) or 0x40000000 | i (MTF).  sh0 = row 0 shifted up by one lane; its lane 0 is never written by the
  // DPP move (no source lane, bound_ctrl off) and keeps the +inf it starts with.
  u32 sh0lo = 0u, sh0hi = 0x7FF00000u;
  u32 pos = (u32)(start + lane);
  // bytes before the tile (run detection across the tile boundary)
  u32 cp = (start >= 1) ? (u32)s[start - 1] : 0x100u;
  uint64_t carryE = (start >= 2 && s[start - 1] == s[start - 2]) ? 1ULL : 0ULL;
  u32 cur = (start + lane < end) ? (u32)s[start + lane] : 0u;
  for (int row = start; row < end; row += 64, pos += 64u) {
    const int cnt = min(64, end - row);
    const int nrow = row + 64;
    const u32 nxt = (nrow + lane < end) ? (u32)s[nrow + lane] : 0u;
    u32 prevb = (u32)__builtin_amdgcn_update_dpp(0, (int)cur, 0x138, 0xF, 0xF, true);
    if (lane == 0) prevb = cp;
    const uint64_t valid = (cnt == 64) ? ~0ULL : ((1ULL << cnt) - 1ULL);
    const uint64_t E = kz_ballot(cur == prevb) & valid;
    const uint64_t R = E & ((E << 1) | carryE);
    carryE = E >> 63;
    uint64_t N = valid & ~R;
    uint64_t F = N & (R << 1);                                        // ranked positions right behind a skipped stretch: repair the front key first
    int jf;
    asm("s_ff1_i32_b64 %0, %1" : "=s"(jf) : "s"(F));
    const u32 xl = (pos << 9) | 0x100u | (255u - cur);
    const u32 ivb = (MODE == 2) ? pos + 0x80000000u : (0x40000000u | pos);
    u32 outv = 0;
    while (N) {
      const int j = (int)__builtin_ctzll(N);
      asm("s_bitset0_b64 %0, %1" : "+s"(N) : "s"(j));
      if (__builtin_expect(j == jf, 0)) {
        KZR_FIX_RUN(row + j - 1)
        asm("s_bitset0_b64 %0, %1" : "+s"(F) : "s"(j));
        asm("s_ff1_i32_b64 %0, %1" : "=s"(jf) : "s"(F));
      }
      const u32 xlo = (u32)__builtin_amdgcn_readlane((int)xl, j);
      const u32 tb = xlo & 0xFFu;
      const uint64_t m0 = kz_ballot(((u32)R0 & 0xFFu) == tb);
      // where the symbol sits (row 0 first: after a BWT most symbols are there) and the low word of its key (p << 9 | ...)
      u32 r, lo = 0;
      if (__builtin_expect(m0 != 0, 1)) {
        r = (u32)__builtin_ctzll(m0);
        if (MODE != 1) lo = (u32)__builtin_amdgcn_readlane((int)(u32)R0, (int)r);
      } else {
        const uint64_t m1 = kz_ballot(((u32)R1 & 0xFFu) == tb), m2 = kz_ballot(((u32)R2 & 0xFFu) == tb), m3 = kz_ballot(((u32)R3 & 0xFFu) == tb);
        if (m1) { r = 64u + (u32)__builtin_ctzll(m1); lo = (u32)__builtin_amdgcn_readlane((int)(u32)R1, (int)(r & 63u)); }
        else if (m2) { r = 128u + (u32)__builtin_ctzll(m2); lo = (u32)__builtin_amdgcn_readlane((int)(u32)R2, (int)(r & 63u)); }
        else { r = 192u + (u32)__builtin_ctzll(m3); lo = (u32)__builtin_amdgcn_readlane((int)(u32)R3, (int)(r & 63u)); }
      }
      const u32 ivs = (MODE == 3) ? 0u : (u32)__builtin_amdgcn_readlane((int)ivb, j);
      const u32 xhi = (MODE == 1) ? ivs : ((MODE == 2) ? (((lo >> 9) + ivs) >> 1) : (0x40000000u | (lo >> 9)));
      const u64 x = ((u64)xhi << 32) | xlo;
      if (__builtin_expect(r >= 64u, 0)) {
        // the rows below row 0 up to the symbol's (selects, not branches: a divergent branch in this loop makes the compiler copy row
        // 0, the shifted row and the output row into temporaries and back around every step: nine moves).  Row 0 itself moves as a
        // whole, which is what the step below does for r >= 63.
        const u64 c0 = kz_readlane64(R0, 63), c1 = kz_readlane64(R1, 63), c2 = kz_readlane64(R2, 63);
        const u64 t1 = kzr_min(x, kzr_shr(R1, c0)), t2 = kzr_min(x, kzr_shr(R2, c1)), t3 = kzr_min(x, kzr_shr(R3, c2));
        const u64 n1 = kzr_max(t1, R1), n2 = kzr_max(t2, R2), n3 = kzr_max(t3, R3);
        R1 = (64u + (u32)lane <= r) ? n1 : R1;
        R2 = (128u + (u32)lane <= r) ? n2 : R2;
        R3 = (192u + (u32)lane <= r) ? n3 : R3;
      }
      sh0lo = (u32)__builtin_amdgcn_update_dpp((int)sh0lo, (int)(u32)R0, 0x138 /*wave_shr:1*/, 0xF, 0xF, false);
      sh0hi = (u32)__builtin_amdgcn_update_dpp((int)sh0hi, (int)(u32)(R0 >> 32), 0x138, 0xF, 0xF, false);
      u64 t0;
      // positions <= r of row 0 take max(min(x, left neighbour), own), lane j of the output row takes the rank: both under EXEC masks
      // made here (EXEC is all ones in this loop and is again when the block ends; the scalar unit writes it: no VALU hazard applies)
      asm volatile("v_min_f64 %[t], %[x], %[sh]\n\t"
                   "v_cmp_ge_u32 vcc, %[r], %[lane]\n\t"
                   "s_mov_b64 exec, vcc\n\t"
                   "v_max_f64 %[R], %[t], %[R]\n\t"
                   "s_lshl_b64 exec, 1, %[j]\n\t"
                   "v_mov_b32 %[o], %[r]\n\t"
                   "s_mov_b64 exec, -1"
                   : [R] "+v"(R0), [o] "+v"(outv), [t] "=&v"(t0)
                   : [x] "s"(x), [sh] "v"(((u64)sh0hi << 32) | sh0lo), [r] "s"(r), [lane] "v"(lane), [j] "s"(j) : "vcc", "scc");   // (s_lshl_b64 writes SCC)
    }
    if ((R >> (cnt - 1)) & 1ULL) KZR_FIX_RUN(row + cnt - 1)            // the row ends inside a skipped stretch
    if (lane < cnt) d[row + lane] = (u8)outv;
    cp = (u32)__builtin_amdgcn_readlane((int)cur, cnt - 1);
    cur = nxt;
  }
}

// ---- inverse v6: the list lives by POSITION ----------------------------------------------------------------
// Row k (k = 0..3) holds list positions 64k..64k+63, one per lane, in two VGPRs:
//   Q = (q << 8) | symbol    (q < 2^24: blocks below 2^24 bytes, as before)      P = p (index of the last occurrence)
// The list is sorted by q descending at all times (a moved symbol's new q is never below its old one), so the bubble
// loop of SBRT.java:194-209 moves the entry at position r up to rp = #{q > qc} and the entries of [rp, r) down by one:
// in-range lanes M = {j <= r : Q_j <= TH}, TH = (qc << 8) | 0xFF, take their left neighbour (one DPP wave_shr per
// register), and the one lane of M whose incoming neighbour is still above TH (or lane 0 of row 0: a sentinel) is rp and
// takes the new entry.  Nothing on that chain leaves the VALU: the lane mask "<= r" is an EXEC mask built from the
// rank by the scalar unit ahead of time, M is EXEC after v_cmpx, v_cmp -> v_cndmask go through an SGPR pair.  v5 kept
// keys by symbol and counted rp with 4 ballots + s_bcnt1: three VALU->SALU hand-offs (~30 cycles each) and ~70
// instructions per non-zero rank whatever its value; here a rank below 64 costs 14 VALU + ~8 SALU instructions and
// each further row of 64 positions the move spans about 12 more.  x = 2q or 2q+1: RANK i + p, MTF 2i, TIMESTAMP 2p.
// Hazards (inline asm is not padded by the compiler): VALU-written SGPR -> VALU read 2 wait states, VALU-written
// VGPR -> DPP 2 / v_readlane 1, VALU-written EXEC -> DPP 5 / v_readlane, v_writelane 4.
// One asm statement walks the non-zero ranks of a 64-rank row (the compiler cannot be trusted to keep eleven loop-carried
// registers in place around per-step asm statements, nor to keep the rank uniform).  Physical temporaries, listed as
// clobbers: v74 tp, v75 tq, v76 xi, v77 th, v78 nq, v79 vi; s40 j, s41 jn, s42 r, s43 r_next, s44 r&63, s45 t,
// s[46:47] lanes <= r&63, s[48:49] non-zero positions left, s50 i, s51 j-prev, s52 pl, s53 x, s54 accessed Q, s55/s56 carries,
// s57 r>>6, s[58:59] insert mask, s60 2i.
#define KZ6_DPP " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define KZ6_XI_RANK(P) "v_add_u32 v76, s50, %[" P "]\n\t"
#define KZ6_XI_MTF(P)  "v_mov_b32 v76, s60\n\t"
#define KZ6_XI_TS(P)   "v_lshlrev_b32 v76, 1, %[" P "]\n\t"
// zero run in front of position j: the front entry (row 0, lane 0) repeats, only its (q,p) change (SBRT.java:194-201):
// p = pl, q = RANK (pl + (run >= 2 ? pl - 1 : p)) >> 1, MTF pl, TIMESTAMP run >= 2 ? pl - 1 : p
#define KZ6_ZQ_RANK "v_mov_b32 v77, s45\n\tv_cndmask_b32 v76, %[p0], v77, vcc\n\tv_add_u32 v76, s52, v76\n\tv_lshrrev_b32 v76, 1, v76\n\t"
#define KZ6_ZQ_MTF  "v_mov_b32 v76, s52\n\t"
#define KZ6_ZQ_TS   "v_mov_b32 v77, s45\n\tv_cndmask_b32 v76, %[p0], v77, vcc\n\t"
#define KZ6_ZERO(ZQ)                                           \
    "s_sub_u32 s52, s50, 1\n\t"                                  \
    "s_cmp_gt_u32 s51, 2\n\t"                                    \
    "s_cselect_b64 vcc, -1, 0\n\t"                               \
    "s_sub_u32 s45, s52, 1\n\t"                                  \
    "s_mov_b64 exec, 1\n\t"                                      \
    ZQ                                                           \
    "v_lshlrev_b32 v76, 8, v76\n\t"                              \
    "v_bfi_b32 %[q0], %[ff], %[q0], v76\n\t"                     \
    "v_mov_b32 %[p0], s52\n\t"                                   \
    "s_mov_b64 exec, -1\n\t"
// head: read the accessed entry (row KR, lane r&63): x and Q
#define KZ6_HEAD(XI, QK, PK)                                   \
    XI(PK)                                                       \
    "v_mov_b32 v79, s50\n\t"                                     \
    "v_readlane_b32 s53, v76, s44\n\t"                           \
    "v_readlane_b32 s54, %[" QK "], s44\n\t"
#define KZ6_THNQ                                               \
    "v_lshl_or_b32 v77, s53, 7, %[ff]\n\t"                       \
    "v_bfi_b32 v78, %[ff], s54, v77\n\t"
// rows 1..3: shifted copies with the carry from the row above in lane 0
#define KZ6_ROW_SHIFT(QK, PK, QM, PM)                          \
    "v_readlane_b32 s55, %[" QM "], 63\n\t"                      \
    "v_readlane_b32 s56, %[" PM "], 63\n\t"                      \
    "v_mov_b32_dpp v75, %[" QK "]" KZ6_DPP                       \
    "v_mov_b32_dpp v74, %[" PK "]" KZ6_DPP                       \
    "v_writelane_b32 v75, s55, 0\n\t"                            \
    "v_writelane_b32 v74, s56, 0\n\t"
#define KZ6_ROW_APPLY(QK, PK, TQ)                              \
    "v_cmpx_le_u32 vcc, %[" QK "], v77\n\t"                      \
    "v_cmp_lt_u32_e64 s[58:59], v77, " TQ "\n\t"                 \
    "s_nop 1\n\t"                                                \
    "v_cndmask_b32_e64 %[" QK "], " TQ ", v78, s[58:59]\n\t"     \
    "v_cndmask_b32_e64 %[" PK "], v74, v79, s[58:59]\n\t"        \
    "s_mov_b64 exec, -1\n\t"
#define KZ6_ROW0                                               \
    "v_mov_b32_dpp %[tq0], %[q0]" KZ6_DPP                        \
    "v_mov_b32_dpp v74, %[p0]" KZ6_DPP
// tail: lane j of outv <- accessed entry (low byte = symbol), lane j of fmv <- M of row 0 (bit 0: the symbol became the front)
#define KZ6_TAIL                                               \
    "s_mov_b32 m0, s40\n\t"                                      \
    "s_nop 0\n\t"                                                \
    "v_writelane_b32 %[outv], s54, m0\n\t"                       \
    "v_writelane_b32 %[fmv], vcc_lo, m0\n\t"
#define KZ6_SETLE "s_mov_b64 exec, s[46:47]\n\t"
// end of a step of the row walk: next non-zero position, or out (every variant carries its own copy: one taken branch less)
#define KZ6_NEXT                                               \
    "s_mov_b32 %[prev], s40\n\t"                                \
    "s_cmp_eq_u64 s[48:49], 0\n\t"                              \
    "s_cbranch_scc1 L_done%=\n\t"                               \
    "s_bitset0_b64 s[48:49], s41\n\t"                           \
    "s_mov_b32 s40, s41\n\t"                                    \
    "s_mov_b32 s42, s43\n\t"                                    \
    "s_branch L_step%=\n\t"

#define KZ6_ROWLOOP(XI, ZQ) asm volatile(                                                      \
    "s_mov_b64 s[48:49], %[nz]\n\t"                                                              \
    "s_mov_b32 %[prev], -1\n\t"                                                                  \
    "s_ff1_i32_b64 s40, s[48:49]\n\t"                                                            \
    "s_bitset0_b64 s[48:49], s40\n\t"                                                            \
    "v_readlane_b32 s42, %[cur], s40\n\t"                                                        \
  "L_step%=:\n\t"                                                                                \
    "s_ff1_i32_b64 s41, s[48:49]\n\t"                                                            \
    "s_max_i32 s41, s41, 0\n\t"                                                                  \
    "v_readlane_b32 s43, %[cur], s41\n\t"              /* rank of the next non-zero position: ready long before it is needed */ \
    "s_add_u32 s50, %[row], s40\n\t"                                                             \
    "s_lshl_b32 s60, s50, 1\n\t"                                                                 \
    "s_sub_u32 s51, s40, %[prev]\n\t"                                                            \
    "s_cmp_lt_u32 s51, 2\n\t"                                                                    \
    "s_cbranch_scc1 L_nz%=\n\t"                                                                  \
    KZ6_ZERO(ZQ)                                                                                 \
  "L_nz%=:\n\t"                                                                                  \
    "s_and_b32 s44, s42, 63\n\t"                                                                 \
    "s_xor_b32 s45, s44, 63\n\t"                                                                 \
    "s_lshr_b64 s[46:47], -1, s45\n\t"                                                           \
    "s_lshr_b32 s57, s42, 6\n\t"                                                                 \
    "s_cmp_lg_u32 s57, 0\n\t"                                                                    \
    "s_cbranch_scc1 L_cold%=\n\t"                                                                \
    KZ6_HEAD(XI, "q0", "p0") KZ6_ROW0 KZ6_THNQ KZ6_SETLE KZ6_ROW_APPLY("q0", "p0", "%[tq0]") KZ6_TAIL \
    KZ6_NEXT                                                                                     \
  "L_cold%=:\n\t"                                                                                \
    "s_cmp_eq_u32 s57, 1\n\t"                                                                    \
    "s_cbranch_scc1 L_k1%=\n\t"                                                                  \
    "s_cmp_eq_u32 s57, 2\n\t"                                                                    \
    "s_cbranch_scc1 L_k2%=\n\t"                                                                  \
    KZ6_HEAD(XI, "q3", "p3") KZ6_ROW_SHIFT("q3", "p3", "q2", "p2") KZ6_THNQ KZ6_SETLE KZ6_ROW_APPLY("q3", "p3", "v75") \
    KZ6_ROW_SHIFT("q2", "p2", "q1", "p1") KZ6_ROW_APPLY("q2", "p2", "v75") KZ6_ROW_SHIFT("q1", "p1", "q0", "p0") KZ6_ROW_APPLY("q1", "p1", "v75") \
    KZ6_ROW0 KZ6_ROW_APPLY("q0", "p0", "%[tq0]") KZ6_TAIL                                        \
    KZ6_NEXT                                                                                     \
  "L_k2%=:\n\t"                                                                                  \
    KZ6_HEAD(XI, "q2", "p2") KZ6_ROW_SHIFT("q2", "p2", "q1", "p1") KZ6_THNQ KZ6_SETLE KZ6_ROW_APPLY("q2", "p2", "v75") \
    KZ6_ROW_SHIFT("q1", "p1", "q0", "p0") KZ6_ROW_APPLY("q1", "p1", "v75") KZ6_ROW0 KZ6_ROW_APPLY("q0", "p0", "%[tq0]") KZ6_TAIL \
    KZ6_NEXT                                                                                     \
  "L_k1%=:\n\t"                                                                                  \
    KZ6_HEAD(XI, "q1", "p1") KZ6_ROW_SHIFT("q1", "p1", "q0", "p0") KZ6_THNQ KZ6_SETLE KZ6_ROW_APPLY("q1", "p1", "v75") \
    KZ6_ROW0 KZ6_ROW_APPLY("q0", "p0", "%[tq0]") KZ6_TAIL                                        \
    KZ6_NEXT                                                                                     \
  "L_done%=:\n\t"                                                                                \
    : [q0]"+v"(Q0), [p0]"+v"(P0), [q1]"+v"(Q1), [p1]"+v"(P1), [q2]"+v"(Q2), [p2]"+v"(P2), [q3]"+v"(Q3), [p3]"+v"(P3),           \
      [tq0]"+v"(tq0), [outv]"+v"(outv), [fmv]"+v"(fmv), [prev]"=&s"(prev)                                                      \
    : [cur]"v"(cur), [ff]"v"(ff), [nz]"s"(nz), [row]"s"(row)                                                                   \
    : "vcc", "scc", "v74", "v75", "v76", "v77", "v78", "v79", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", \
      "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60");

// Rows of 64 ranks with few zeros and few ranks >= 64 (poorly compressible data: the blocks that set the run time): straight-line
// code, one step per rank with constant lane numbers, zero ranks included (a rank 0 is an ordinary step: M = lane 0).  No
// loop control, no taken branch on the way; a rank >= 64 leaves through a stub to the row-spanning variants and comes back
// with s_setpc.  RC / RN: the SGPRs holding this step's and the next step's rank (s64 / s65 alternate).
#define KZ6A_STEP(J, JN, RC, RN, XI)                           \
    "v_readlane_b32 " RN ", %[cur], " JN "\n\t"                  \
    "s_and_b32 s44, " RC ", 63\n\t"                               \
    "s_xor_b32 s45, s44, 63\n\t"                                  \
    "s_lshr_b64 s[46:47], -1, s45\n\t"                            \
    "s_lshl_b32 s60, s50, 1\n\t"                                  \
    "s_cmp_ge_u32 " RC ", 64\n\t"                                 \
    "s_cbranch_scc1 L_stub" J "_%=\n\t"                           \
    KZ6_HEAD(XI, "q0", "p0") KZ6_ROW0 KZ6_THNQ KZ6_SETLE         \
    "v_cmpx_le_u32 vcc, %[q0], v77\n\t"                          \
    "v_cmp_lt_u32_e64 s[58:59], v77, %[tq0]\n\t"                 \
    "s_add_u32 s50, s50, 1\n\t"                                  \
    "s_nop 0\n\t"                                                \
    "v_cndmask_b32_e64 %[q0], %[tq0], v78, s[58:59]\n\t"         \
    "v_cndmask_b32_e64 %[p0], v74, v79, s[58:59]\n\t"            \
    "s_mov_b64 exec, -1\n\t"                                     \
    "v_writelane_b32 %[outv], s54, " J "\n\t"                    \
  "L_after" J "_%=:\n\t"
#define KZ6A_STUB(J, RC)                                       \
  "L_stub" J "_%=:\n\t"                                          \
    "s_mov_b32 s40, " J "\n\t"                                   \
    "s_mov_b32 s42, " RC "\n\t"                                  \
    "s_getpc_b64 s[62:63]\n\t"                                   \
  "L_pc" J "_%=:\n\t"                                            \
    "s_sub_u32 s62, s62, L_pc" J "_%=-L_after" J "_%=\n\t"       \
    "s_subb_u32 s63, s63, 0\n\t"                                 \
    "s_branch L_coldA_%=\n\t"
#define KZ6A_STEP2(J0, J1, J2, XI) KZ6A_STEP(J0, J1, "s64", "s65", XI) KZ6A_STEP(J1, J2, "s65", "s64", XI)
#define KZ6A_STUB2(J0, J1) KZ6A_STUB(J0, "s64") KZ6A_STUB(J1, "s65")
#define KZ6A_STEPS(XI) \
    KZ6A_STEP2("0", "1", "2", XI) KZ6A_STEP2("2", "3", "4", XI) KZ6A_STEP2("4", "5", "6", XI) KZ6A_STEP2("6", "7", "8", XI) \
    KZ6A_STEP2("8", "9", "10", XI) KZ6A_STEP2("10", "11", "12", XI) KZ6A_STEP2("12", "13", "14", XI) KZ6A_STEP2("14", "15", "16", XI) \
    KZ6A_STEP2("16", "17", "18", XI) KZ6A_STEP2("18", "19", "20", XI) KZ6A_STEP2("20", "21", "22", XI) KZ6A_STEP2("22", "23", "24", XI) \
    KZ6A_STEP2("24", "25", "26", XI) KZ6A_STEP2("26", "27", "28", XI) KZ6A_STEP2("28", "29", "30", XI) KZ6A_STEP2("30", "31", "32", XI) \
    KZ6A_STEP2("32", "33", "34", XI) KZ6A_STEP2("34", "35", "36", XI) KZ6A_STEP2("36", "37", "38", XI) KZ6A_STEP2("38", "39", "40", XI) \
    KZ6A_STEP2("40", "41", "42", XI) KZ6A_STEP2("42", "43", "44", XI) KZ6A_STEP2("44", "45", "46", XI) KZ6A_STEP2("46", "47", "48", XI) \
    KZ6A_STEP2("48", "49", "50", XI) KZ6A_STEP2("50", "51", "52", XI) KZ6A_STEP2("52", "53", "54", XI) KZ6A_STEP2("54", "55", "56", XI) \
    KZ6A_STEP2("56", "57", "58", XI) KZ6A_STEP2("58", "59", "60", XI) KZ6A_STEP2("60", "61", "62", XI) KZ6A_STEP2("62", "63", "0", XI)
#define KZ6A_STUBS \
    KZ6A_STUB2("0", "1") KZ6A_STUB2("2", "3") KZ6A_STUB2("4", "5") KZ6A_STUB2("6", "7") KZ6A_STUB2("8", "9") KZ6A_STUB2("10", "11") \
    KZ6A_STUB2("12", "13") KZ6A_STUB2("14", "15") KZ6A_STUB2("16", "17") KZ6A_STUB2("18", "19") KZ6A_STUB2("20", "21") KZ6A_STUB2("22", "23") \
    KZ6A_STUB2("24", "25") KZ6A_STUB2("26", "27") KZ6A_STUB2("28", "29") KZ6A_STUB2("30", "31") KZ6A_STUB2("32", "33") KZ6A_STUB2("34", "35") \
    KZ6A_STUB2("36", "37") KZ6A_STUB2("38", "39") KZ6A_STUB2("40", "41") KZ6A_STUB2("42", "43") KZ6A_STUB2("44", "45") KZ6A_STUB2("46", "47") \
    KZ6A_STUB2("48", "49") KZ6A_STUB2("50", "51") KZ6A_STUB2("52", "53") KZ6A_STUB2("54", "55") KZ6A_STUB2("56", "57") KZ6A_STUB2("58", "59") \
    KZ6A_STUB2("60", "61") KZ6A_STUB2("62", "63")
#define KZ6A_RET "s_add_u32 s50, s50, 1\n\t" "s_setpc_b64 s[62:63]\n\t"
#define KZ6_ROWDENSE(XI) asm volatile(                                                         \
    "s_mov_b32 s50, %[row]\n\t"                                                                  \
    "v_readlane_b32 s64, %[cur], 0\n\t"                                                          \
    KZ6A_STEPS(XI)                                                                               \
    "s_branch L_done%=\n\t"                                                                      \
    KZ6A_STUBS                                                                                   \
  "L_coldA_%=:\n\t"                                                                              \
    "s_lshr_b32 s57, s42, 6\n\t"                                                                 \
    "s_cmp_eq_u32 s57, 1\n\t"                                                                    \
    "s_cbranch_scc1 L_k1%=\n\t"                                                                  \
    "s_cmp_eq_u32 s57, 2\n\t"                                                                    \
    "s_cbranch_scc1 L_k2%=\n\t"                                                                  \
    KZ6_HEAD(XI, "q3", "p3") KZ6_ROW_SHIFT("q3", "p3", "q2", "p2") KZ6_THNQ KZ6_SETLE KZ6_ROW_APPLY("q3", "p3", "v75") \
    KZ6_ROW_SHIFT("q2", "p2", "q1", "p1") KZ6_ROW_APPLY("q2", "p2", "v75") KZ6_ROW_SHIFT("q1", "p1", "q0", "p0") KZ6_ROW_APPLY("q1", "p1", "v75") \
    KZ6_ROW0 KZ6_ROW_APPLY("q0", "p0", "%[tq0]") KZ6_TAIL KZ6A_RET                               \
  "L_k2%=:\n\t"                                                                                  \
    KZ6_HEAD(XI, "q2", "p2") KZ6_ROW_SHIFT("q2", "p2", "q1", "p1") KZ6_THNQ KZ6_SETLE KZ6_ROW_APPLY("q2", "p2", "v75") \
    KZ6_ROW_SHIFT("q1", "p1", "q0", "p0") KZ6_ROW_APPLY("q1", "p1", "v75") KZ6_ROW0 KZ6_ROW_APPLY("q0", "p0", "%[tq0]") KZ6_TAIL KZ6A_RET \
  "L_k1%=:\n\t"                                                                                  \
    KZ6_HEAD(XI, "q1", "p1") KZ6_ROW_SHIFT("q1", "p1", "q0", "p0") KZ6_THNQ KZ6_SETLE KZ6_ROW_APPLY("q1", "p1", "v75") \
    KZ6_ROW0 KZ6_ROW_APPLY("q0", "p0", "%[tq0]") KZ6_TAIL KZ6A_RET                               \
  "L_done%=:\n\t"                                                                                \
    : [q0]"+v"(Q0), [p0]"+v"(P0), [q1]"+v"(Q1), [p1]"+v"(P1), [q2]"+v"(Q2), [p2]"+v"(P2), [q3]"+v"(Q3), [p3]"+v"(P3),           \
      [tq0]"+v"(tq0), [outv]"+v"(outv), [fmv]"+v"(fmv)                                                                         \
    : [cur]"v"(cur), [ff]"v"(ff), [row]"s"(row)                                                                                \
    : "vcc", "scc", "v74", "v75", "v76", "v77", "v78", "v79", "s40", "s42", "s44", "s45", "s46", "s47",                        \
      "s50", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s62", "s63", "s64", "s65");

// Dense rows with many ranks >= 64 (incompressible data): a plain loop over the 64 positions, zero ranks as ordinary steps (no run
// bookkeeping, no s_ff1 walk), the row-spanning variant picked by two bit tests of the rank: K0 falls through, K1 / K2 take one
// branch, K3 two, plus the loop's back edge.
#define KZ6C_TAIL                                              \
    "s_mov_b32 m0, s40\n\t"                                      \
    "s_nop 0\n\t"                                                \
    "v_writelane_b32 %[outv], s54, m0\n\t"
#define KZ6C_NEXT                                              \
    "s_add_u32 s50, s50, 1\n\t"                                  \
    "s_add_u32 s40, s40, 1\n\t"                                  \
    "s_mov_b32 s42, s43\n\t"                                     \
    "s_cmp_lt_u32 s40, 64\n\t"                                   \
    "s_cbranch_scc1 L_step%=\n\t"                                \
    "s_branch L_done%=\n\t"
#define KZ6_ROWCOLD(XI) asm volatile(                                                          \
    "s_mov_b32 s40, 0\n\t"                                                                       \
    "s_mov_b32 s50, %[row]\n\t"                                                                  \
    "v_readlane_b32 s42, %[cur], 0\n\t"                                                          \
  "L_step%=:\n\t"                                                                                \
    "s_add_u32 s41, s40, 1\n\t"                                                                  \
    "s_and_b32 s41, s41, 63\n\t"                                                                 \
    "v_readlane_b32 s43, %[cur], s41\n\t"                                                        \
    "s_and_b32 s44, s42, 63\n\t"                                                                 \
    "s_xor_b32 s45, s44, 63\n\t"                                                                 \
    "s_lshr_b64 s[46:47], -1, s45\n\t"                                                           \
    "s_lshl_b32 s60, s50, 1\n\t"                                                                 \
    "s_bitcmp1_b32 s42, 7\n\t"                                                                   \
    "s_cbranch_scc1 L_hi%=\n\t"                                                                  \
    "s_bitcmp1_b32 s42, 6\n\t"                                                                   \
    "s_cbranch_scc1 L_k1%=\n\t"                                                                  \
    KZ6_HEAD(XI, "q0", "p0") KZ6_ROW0 KZ6_THNQ KZ6_SETLE KZ6_ROW_APPLY("q0", "p0", "%[tq0]") KZ6C_TAIL KZ6C_NEXT \
  "L_k1%=:\n\t"                                                                                  \
    KZ6_HEAD(XI, "q1", "p1") KZ6_ROW_SHIFT("q1", "p1", "q0", "p0") KZ6_THNQ KZ6_SETLE KZ6_ROW_APPLY("q1", "p1", "v75") \
    KZ6_ROW0 KZ6_ROW_APPLY("q0", "p0", "%[tq0]") KZ6C_TAIL KZ6C_NEXT                             \
  "L_hi%=:\n\t"                                                                                  \
    "s_bitcmp1_b32 s42, 6\n\t"                                                                   \
    "s_cbranch_scc1 L_k3%=\n\t"                                                                  \
    KZ6_HEAD(XI, "q2", "p2") KZ6_ROW_SHIFT("q2", "p2", "q1", "p1") KZ6_THNQ KZ6_SETLE KZ6_ROW_APPLY("q2", "p2", "v75") \
    KZ6_ROW_SHIFT("q1", "p1", "q0", "p0") KZ6_ROW_APPLY("q1", "p1", "v75") KZ6_ROW0 KZ6_ROW_APPLY("q0", "p0", "%[tq0]") KZ6C_TAIL KZ6C_NEXT \
  "L_k3%=:\n\t"                                                                                  \
    KZ6_HEAD(XI, "q3", "p3") KZ6_ROW_SHIFT("q3", "p3", "q2", "p2") KZ6_THNQ KZ6_SETLE KZ6_ROW_APPLY("q3", "p3", "v75") \
    KZ6_ROW_SHIFT("q2", "p2", "q1", "p1") KZ6_ROW_APPLY("q2", "p2", "v75") KZ6_ROW_SHIFT("q1", "p1", "q0", "p0") KZ6_ROW_APPLY("q1", "p1", "v75") \
    KZ6_ROW0 KZ6_ROW_APPLY("q0", "p0", "%[tq0]") KZ6C_TAIL KZ6C_NEXT                             \
  "L_done%=:\n\t"                                                                                \
    : [q0]"+v"(Q0), [p0]"+v"(P0), [q1]"+v"(Q1), [p1]"+v"(P1), [q2]"+v"(Q2), [p2]"+v"(P2), [q3]"+v"(Q3), [p3]"+v"(P3),           \
      [tq0]"+v"(tq0), [outv]"+v"(outv)                                                                                         \
    : [cur]"v"(cur), [ff]"v"(ff), [row]"s"(row)                                                                                \
    : "vcc", "scc", "v74", "v75", "v76", "v77", "v78", "v79", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47",          \
      "s50", "s53", "s54", "s55", "s56", "s58", "s59", "s60");

// ---- cold rows, interleaved layout -------------------------------------------------------------------------------------
// Rows of high ranks (uniform or badly predicted data: most ranks above 63) paid two or three taken branches per rank in
// KZ6_ROWCOLD (dispatch on the row of the accessed position, ~40 cycles each for a lone wave) plus the lane-63 -> lane-0
// carries between rows.  For such rows the list is re-laid out: position j sits in register (j & 3), lane (j >> 2).
// A shift by one position is then "register k takes register k-1 of the same lane" for k = 1..3 and one DPP wave_shr for
// register 0: every rank runs the SAME straight-line code (four identical register blocks, processed 3, 2, 1, 0 so that
// each reads its neighbour before it changes), no dispatch, no carries.  The accessed entry is fetched with a register-indexed v_mov
// (s_set_gpr_idx_on, index r & 3) + v_readlane (lane r >> 2); x, TH and the new Q are scalar arithmetic.
// The two layouts are converted with ds_bpermute when a cold row follows a warm one or the other way round.
#define KZ6I_XS_RANK "s_add_u32 s53, s50, s53\n\t"
#define KZ6I_XS_MTF  "s_lshl_b32 s53, s50, 1\n\t"
#define KZ6I_XS_TS   "s_lshl_b32 s53, s53, 1\n\t"
#define KZ6I_REG(POS, Q, P, NQ, NP, RC)                        \
    "v_cmpx_ge_u32 vcc, " RC ", " POS "\n\t"                    \
    "v_cmpx_ge_u32 vcc, s55, " Q "\n\t"                         \
    "v_cmp_lt_u32_e64 s[58:59], s55, " NQ "\n\t"                \
    "s_nop 1\n\t"                                               \
    "v_cndmask_b32_e64 " Q ", " NQ ", v78, s[58:59]\n\t"        \
    "v_cndmask_b32_e64 " P ", " NP ", v79, s[58:59]\n\t"        \
    "s_mov_b64 exec, -1\n\t"
// one rank at row position J (a constant); RC holds its rank, RN receives the next one (the two alternate)
#define KZ6I_STEP(J, JN, RC, RN, XS)                           \
    "s_and_b32 s45, " RC ", 3\n\t"                              \
    "v_readlane_b32 " RN ", %[cur], " #JN "\n\t"                \
    "s_lshr_b32 s44, " RC ", 2\n\t"                             \
    "s_set_gpr_idx_on s45, gpr_idx(SRC0)\n\t"                   \
    "v_mov_b32 v76, v64\n\t"                                    \
    "v_mov_b32 v77, v68\n\t"                                    \
    "s_set_gpr_idx_off\n\t"                                     \
    "v_mov_b32 v79, s50\n\t"                                    \
    "v_readlane_b32 s54, v76, s44\n\t"                          \
    "v_readlane_b32 s53, v77, s44\n\t"                          \
    "v_mov_b32_dpp %[tq0], v67" KZ6_DPP                         \
    "v_mov_b32_dpp v74, v71" KZ6_DPP                            \
    XS                                                          \
    "s_lshl_b32 s55, s53, 7\n\t"                                \
    "s_or_b32 s55, s55, 0xff\n\t"                               \
    "v_mov_b32 v77, s55\n\t"                                    \
    "v_bfi_b32 v78, %[ff], s54, v77\n\t"                        \
    KZ6I_REG("v59", "v67", "v71", "v66", "v70", RC)             \
    KZ6I_REG("v58", "v66", "v70", "v65", "v69", RC)             \
    KZ6I_REG("v57", "v65", "v69", "v64", "v68", RC)             \
    KZ6I_REG("v56", "v64", "v68", "%[tq0]", "v74", RC)          \
    "v_writelane_b32 %[outv], s54, " #J "\n\t"                  \
    "s_add_u32 s50, s50, 1\n\t"
#define KZ6I_STEP2(A, B, C, XS) KZ6I_STEP(A, B, "s42", "s43", XS) KZ6I_STEP(B, C, "s43", "s42", XS)
#define KZ6I_STEP8(A, B, C, D, E, F, G, H, I, XS) KZ6I_STEP2(A, B, C, XS) KZ6I_STEP2(C, D, E, XS) KZ6I_STEP2(E, F, G, XS) KZ6I_STEP2(G, H, I, XS)
#define KZ6I_ROWCOLD(XS) asm volatile(                                                         \
    "s_mov_b32 s41, m0\n\t"                      /* s_set_gpr_idx_on writes m0: saved and restored, not clobbered */ \
    "v_mov_b32 v64, %[q0]\n\tv_mov_b32 v65, %[q1]\n\tv_mov_b32 v66, %[q2]\n\tv_mov_b32 v67, %[q3]\n\t"  \
    "v_mov_b32 v68, %[p0]\n\tv_mov_b32 v69, %[p1]\n\tv_mov_b32 v70, %[p2]\n\tv_mov_b32 v71, %[p3]\n\t"  \
    "v_mov_b32 v56, %[lane4]\n\tv_add_u32 v57, 1, %[lane4]\n\tv_add_u32 v58, 2, %[lane4]\n\tv_add_u32 v59, 3, %[lane4]\n\t" \
    "s_mov_b32 s50, %[row]\n\t"                                                                  \
    "v_readlane_b32 s42, %[cur], 0\n\t"                                                          \
    KZ6I_STEP8(0, 1, 2, 3, 4, 5, 6, 7, 8, XS) KZ6I_STEP8(8, 9, 10, 11, 12, 13, 14, 15, 16, XS)    \
    KZ6I_STEP8(16, 17, 18, 19, 20, 21, 22, 23, 24, XS) KZ6I_STEP8(24, 25, 26, 27, 28, 29, 30, 31, 32, XS) \
    KZ6I_STEP8(32, 33, 34, 35, 36, 37, 38, 39, 40, XS) KZ6I_STEP8(40, 41, 42, 43, 44, 45, 46, 47, 48, XS) \
    KZ6I_STEP8(48, 49, 50, 51, 52, 53, 54, 55, 56, XS) KZ6I_STEP8(56, 57, 58, 59, 60, 61, 62, 63, 0, XS)  \
    "v_mov_b32 %[q0], v64\n\tv_mov_b32 %[q1], v65\n\tv_mov_b32 %[q2], v66\n\tv_mov_b32 %[q3], v67\n\t"  \
    "v_mov_b32 %[p0], v68\n\tv_mov_b32 %[p1], v69\n\tv_mov_b32 %[p2], v70\n\tv_mov_b32 %[p3], v71\n\t"  \
    "s_mov_b32 m0, s41\n\t"                                                                      \
    : [q0]"+v"(Q0), [p0]"+v"(P0), [q1]"+v"(Q1), [p1]"+v"(P1), [q2]"+v"(Q2), [p2]"+v"(P2), [q3]"+v"(Q3), [p3]"+v"(P3),           \
      [tq0]"+v"(tq0), [outv]"+v"(outv)                                                                                         \
    : [cur]"v"(cur), [lane4]"v"(lane4), [ff]"v"(ff), [row]"s"(row)                                                             \
    : "vcc", "scc", "s41", "v56", "v57", "v58", "v59", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71",                  \
      "v74", "v76", "v77", "v78", "v79", "s42", "s43", "s44", "s45", "s50", "s53", "s54", "s55", "s58", "s59");
// layout changes: by position (register k = positions 64k..64k+63) <-> interleaved (register k = positions 4l + k)
__device__ __forceinline__ void kz6_to_interleaved(u32& A0, u32& A1, u32& A2, u32& A3, int lane) {
  const int grp = lane >> 4;
  u32 N[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int a = (4 * (lane & 15) + r) << 2;
    const u32 t0 = (u32)__builtin_amdgcn_ds_bpermute(a, (int)A0), t1 = (u32)__builtin_amdgcn_ds_bpermute(a, (int)A1);
    const u32 t2 = (u32)__builtin_amdgcn_ds_bpermute(a, (int)A2), t3 = (u32)__builtin_amdgcn_ds_bpermute(a, (int)A3);
    N[r] = grp == 0 ? t0 : (grp == 1 ? t1 : (grp == 2 ? t2 : t3));
  }
  A0 = N[0]; A1 = N[1]; A2 = N[2]; A3 = N[3];
}
__device__ __forceinline__ void kz6_from_interleaved(u32& A0, u32& A1, u32& A2, u32& A3, int lane) {
  const int sel = lane & 3;
  u32 N[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int a = (16 * k + (lane >> 2)) << 2;
    const u32 t0 = (u32)__builtin_amdgcn_ds_bpermute(a, (int)A0), t1 = (u32)__builtin_amdgcn_ds_bpermute(a, (int)A1);
    const u32 t2 = (u32)__builtin_amdgcn_ds_bpermute(a, (int)A2), t3 = (u32)__builtin_amdgcn_ds_bpermute(a, (int)A3);
    N[k] = sel == 0 ? t0 : (sel == 1 ? t1 : (sel == 2 ? t2 : t3));
  }
  A0 = N[0]; A1 = N[1]; A2 = N[2]; A3 = N[3];
}

#include "kz_sbrt_f64.h"

// zero run of zr ranks ending at index pl, compiler form (row tails and all-zero rows)
#define KZ6_ZERO_RUN(zr, plv)                                                                  \
  { const u32 pl = (u32)(plv);                                                                 \
    const u32 t = (MODE == 1) ? pl : (((zr) >= 2) ? pl - 1u : ((MODE == 2) ? ((pl + P0) >> 1) : P0)); \
    const u32 nqf = (t << 8) | (Q0 & 0xFFu);                                                   \
    Q0 = isLane0 ? nqf : Q0; P0 = isLane0 ? pl : P0; }

template <int MODE>
__global__ __launch_bounds__(512) void k_sbrt_inverse(const u8* __restrict__ src, u8* __restrict__ dst, int64_t stride,
                                                       const int32_t* __restrict__ d_len, const int32_t* __restrict__ order, int wavesPerGroup, int prio) {
  const int b = __builtin_amdgcn_readfirstlane(order[blockIdx.x * wavesPerGroup + (int)(threadIdx.x >> 6)]);
  if (b < 0) return;
  // one serial chain per wave.  The overlapped decoder schedule (kz_api.hip) runs several launches of this kernel and HBM-bound
  // kernels on the same SIMDs: the launch on the critical path (the most expensive blocks) issues ahead of its neighbours
  const bool useOldCold = (prio & 8) != 0;                          // A/B switch (KZ_SBRT_OLDCOLD): cold rows in the by-position layout
  if ((prio & 7) >= 2) __builtin_amdgcn_s_setprio(3);
  else if ((prio & 7) == 1) __builtin_amdgcn_s_setprio(1);
  const int n = __builtin_amdgcn_readfirstlane(d_len[b]);
  const u8* s = src + (int64_t)b * stride;
  u8* d = dst + (int64_t)b * stride;
  const int lane = kz_lane();
  const bool isLane0 = lane == 0;
  u32 Q0 = (u32)lane, Q1 = 64u + (u32)lane, Q2 = 128u + (u32)lane, Q3 = 192u + (u32)lane;   // q = 0, symbols in order (SBRT.java:176-180)
  u32 P0 = 0, P1 = 0, P2 = 0, P3 = 0;
  u32 tq0 = 0xFFFFFFFFu;                      // lane 0 stays the sentinel "above every key": the DPP shift never writes it
  const u32 ff = 0xFFu;
  u32 fmv = 0;
  const u32 lane4 = 4u * (u32)lane;
  bool inter = false;                         // the list is in the interleaved layout (between consecutive cold rows)
  // Round 6: rows that cost the 32-bit forms more than 64 x 28 instructions (many ranks >= 64, or many non-zero ranks) run in the
  // 64-bit-key form of kz_sbrt_f64.h (interleaved layout, Qk / Pk then hold the low / high words of register k).
  bool keyed = false;
  int s0 = -1;                                // the symbol decoded at i = 0 (kzf_pack)
  const bool allowKeyed = (prio & 16) == 0 && n <= (1 << 23);
  u32 &A0 = Q0, &A1 = Q1, &A2 = Q2, &A3 = Q3, &B0 = P0, &B1 = P1, &B2 = P2, &B3 = P3;
  const u32 m256 = 0xFFFFFF00u, infHi = 0x7FF00000u, lanev = (u32)lane;
  const u32 c128 = 128u, c192 = 192u;
  const u32 c100 = 0x100u, c80 = 0x80u, c512 = 512u;                // 0x100 << 22 = 0x80 << 23 = bit 30 of a key's high word
  u32 cur = (lane < n) ? (u32)s[lane] : 0u;
  for (int row = 0; row < n; row += 64) {
    const int cnt = min(64, n - row);
    const int nrow = row + 64;
    const u32 nxt = (nrow + lane < n) ? (u32)s[nrow + lane] : 0u;  // prefetch the next row
    const uint64_t nz = kz_ballot(cur != 0 && lane < cnt);
    // non-zero lanes receive their symbol with v_writelane; zero ranks output the front symbol of their time
    u32 outv = 0;
    int prev = -1;
    const bool dense = cnt == 64 && __builtin_popcountll(nz) >= 48;
    const bool cold = dense && __builtin_popcountll(kz_ballot(cur >= 64u)) > 6;
    bool wantKeyed = false, wantFlat = false, flatDeep = false;                       // keyed rows: interleaved (any rank), or by position (no rank >= 64: 13 instructions per rank)
    if (allowKeyed && cnt == 64) {
      // what the 32-bit forms would spend on this row (instructions): the row walk pays ~40 for a non-zero rank below 64,
      // ~62 / 76 / 88 for one in rows 1 / 2 / 3 of the list and ~11 per zero run; the dense rows 22 per rank (+ ~50 per rank >= 64),
      // the interleaved 32-bit form 47 per rank.  The keyed form: 28 per rank, zero or not.  (Hysteresis: a change of form costs
      // two layout conversions.)
      const int nn = __builtin_popcountll(nz);
      if (keyed || nn >= 16) {                                      // (fewer than 16 non-zero ranks never reach the thresholds)
        const int nd = __builtin_popcountll(kz_ballot(cur >= 64u)), nd2 = __builtin_popcountll(kz_ballot(cur >= 160u));
        const int runs = __builtin_popcountll(nz & ~(nz << 1));
        const int est = dense ? (cold ? 47 * 64 : 22 * 64 + 50 * nd) : 14 + 40 * (nn - nd) + 65 * (nd - nd2) + 84 * nd2 + 11 * runs;
        const int flatCost = nd == 0 ? 64 * 13 : 64 * 15 + 70 * nd;             // by position: a rank >= 64 takes the row-spanning stub
        const int interCost = 64 * 28;
        if (nd <= 8 && flatCost <= interCost) wantFlat = est > flatCost + ((keyed && !inter) ? -10 : 190);
        else wantKeyed = est > interCost + ((keyed && inter) ? -90 : 160);
        flatDeep = nd > 0;
      }
    }
    const bool wantInter = wantKeyed || (!wantFlat && cold && !useOldCold);
    if (keyed && !(wantKeyed || wantFlat)) { kzf_unpack(A0, B0); kzf_unpack(A1, B1); kzf_unpack(A2, B2); kzf_unpack(A3, B3); keyed = false; }
    if (wantInter != inter) {                                             // (uniform) change of layout
      if (wantInter) { kz6_to_interleaved(Q0, Q1, Q2, Q3, lane); kz6_to_interleaved(P0, P1, P2, P3, lane); }
      else { kz6_from_interleaved(Q0, Q1, Q2, Q3, lane); kz6_from_interleaved(P0, P1, P2, P3, lane); }
      inter = wantInter;
    }
    const u32 f0 = (u32)__builtin_amdgcn_readfirstlane((int)Q0) & 0xFFu;   // the front symbol at the start of the row (32-bit forms; position 0 is lane 0 of Q0 in both layouts)
    if ((wantKeyed || wantFlat) && !keyed) { kzf_pack(Q0, P0, s0); kzf_pack(Q1, P1, s0); kzf_pack(Q2, P2, s0); kzf_pack(Q3, P3, s0); keyed = true; }
    if (keyed) {
      const u32 c2 = (((u32)row << 1) << 8) - 256u;                 // the step adds 512 first: ((2 i + 1) << 8) at step i
      const u32 h0 = 0x40000000u + (u32)row - 1u;                   // MTF: x.hi = 0x40000000 | i, counted up by the step
      if (inter) {
        if (MODE == 2) { if (row + 64 <= (1 << 22)) { KZF_ROW(KZF_X_RANK) } else { KZF_ROW(KZF_X_RANK_HI) } }
        else if (MODE == 1) { KZF_ROW(KZF_X_MTF) } else { KZF_ROW(KZF_X_TS) }
      } else {
        if (flatDeep) {
          if (MODE == 2) { if (row + 64 <= (1 << 22)) { KZD_ROWC(KZF_X_RANK) } else { KZD_ROWC(KZF_X_RANK_HI) } }
          else if (MODE == 1) { KZD_ROWC(KZF_X_MTF) } else { KZD_ROWC(KZF_X_TS) }
        } else {
          if (MODE == 2) { if (row + 64 <= (1 << 22)) { KZD_ROW(KZF_X_RANK) } else { KZD_ROW(KZF_X_RANK_HI) } }
          else if (MODE == 1) { KZD_ROW(KZF_X_MTF) } else { KZD_ROW(KZF_X_TS) }
        }
      }
      outv = ~outv;                                                 // (the keys hold 255 - symbol)
    } else if (dense) {
      if (!cold) {
        if (MODE == 2) { KZ6_ROWDENSE(KZ6_XI_RANK) } else if (MODE == 1) { KZ6_ROWDENSE(KZ6_XI_MTF) } else { KZ6_ROWDENSE(KZ6_XI_TS) }
      } else if (useOldCold) {
        if (MODE == 2) { KZ6_ROWCOLD(KZ6_XI_RANK) } else if (MODE == 1) { KZ6_ROWCOLD(KZ6_XI_MTF) } else { KZ6_ROWCOLD(KZ6_XI_TS) }
      } else {
        if (MODE == 2) { KZ6I_ROWCOLD(KZ6I_XS_RANK) } else if (MODE == 1) { KZ6I_ROWCOLD(KZ6I_XS_MTF) } else { KZ6I_ROWCOLD(KZ6I_XS_TS) }
      }
    } else if (nz) {
      if (MODE == 2) { KZ6_ROWLOOP(KZ6_XI_RANK, KZ6_ZQ_RANK) } else if (MODE == 1) { KZ6_ROWLOOP(KZ6_XI_MTF, KZ6_ZQ_MTF) } else { KZ6_ROWLOOP(KZ6_XI_TS, KZ6_ZQ_TS) }
    }
    if (!dense && !keyed) {
    { const int zr = cnt - prev - 1; if (zr > 0) KZ6_ZERO_RUN(zr, row + cnt - 1) }
    {
      // zero-rank lane l: symbol of the last front change before l (held by that lane), else the front at row start
      const uint64_t fm = kz_ballot((fmv & 1u) != 0) & nz;
      const uint64_t below = fm & kz_lanemask_lt();
      const int srcLane = below ? 63 - (int)__builtin_clzll(below) : 0;
      const u32 fv = (u32)__shfl((int)outv, srcLane, 64);
      if (!((nz >> lane) & 1ULL)) outv = below ? fv : f0;
    }
    }
    if (lane < cnt) d[row + lane] = (u8)outv;
    if (row == 0) s0 = __builtin_amdgcn_readfirstlane((int)(outv & 0xFFu));
    cur = nxt;
  }
}

// Blocks of 2^24-256 bytes and more (round 5): the packed (q << 8 | symbol) list of k_sbrt_inverse holds 24-bit timestamps, so
// longer blocks take this plain form of SBRT.inverse (SBRT.java:154-214): one wave per block, the list BY POSITION in LDS
// (symbol, q, p of the entry at position j), a step = the reference's bubble loop done at once: the entry at rank r moves up to
// rp = 1 + the highest position below r whose q is above the new qc, the entries of [rp, r) move down by one.  Zero ranks (most
// of a transformed block) only touch position 0.  ~200 ns per non-zero rank: a fallback for rare block sizes, not a fast path.
template <int MODE>
__global__ __launch_bounds__(64) void k_sbrt_inverse_wide(const u8* __restrict__ src, u8* __restrict__ dst, int64_t stride, const int32_t* __restrict__ d_len, int B) {
  const int b = blockIdx.x;
  if (b >= B) return;
  const int n = d_len[b];
  if (n <= 0) return;
  __shared__ u32 S[256], Q[256], P[256];
  const int lane = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; k++) { S[k * 64 + lane] = (u32)(k * 64 + lane); Q[k * 64 + lane] = 0; P[k * 64 + lane] = 0; }
  __syncthreads();
  const u8* in = src + (int64_t)b * stride;
  u8* out = dst + (int64_t)b * stride;
  constexpr u32 m1 = (MODE == 3) ? 0u : 0xFFFFFFFFu;     // MODE_TIMESTAMP
  constexpr u32 m2 = (MODE == 1) ? 0u : 0xFFFFFFFFu;     // MODE_MTF
  constexpr int sh = (MODE == 2) ? 1 : 0;                // MODE_RANK
  for (int base = 0; base < n; base += 64) {
    const int cnt = min(64, n - base);
    const u32 rv = lane < cnt ? (u32)in[base + lane] : 0u;
    u32 ov = 0;
    for (int t = 0; t < cnt; t++) {
      const int r = __builtin_amdgcn_readlane((int)rv, t);
      const u32 i = (u32)(base + t);
      const u32 c = S[r], pc = P[r];                      // uniform addresses: broadcast reads
      const u32 qc = ((i & m1) + (pc & m2)) >> sh;
      if (lane == t) ov = c;
      int rp = 0;
      if (r > 0) {
        // highest position j < r with Q[j] > qc (the bubble loop stops there: SBRT.java:203)
        int hi = -1;
#pragma unroll
        for (int k = 3; k >= 0; k--) {
          const int j = k * 64 + lane;
          const uint64_t g = kz_ballot(j < r && Q[j] > qc);
          if (g && hi < 0) hi = k * 64 + 63 - (int)__builtin_clzll(g);
        }
        rp = hi + 1;
        if (rp < r) {
          u32 ts[4], tq[4], tp[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int j = k * 64 + lane;
            const bool mv = j > rp && j <= r;
            ts[k] = mv ? S[j - 1] : 0; tq[k] = mv ? Q[j - 1] : 0; tp[k] = mv ? P[j - 1] : 0;
          }
          __syncthreads();                                  // every lane's loads before any lane's stores (lane 0 of group k+1 reads what lane 63 of group k writes)
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int j = k * 64 + lane;
            if (j > rp && j <= r) { S[j] = ts[k]; Q[j] = tq[k]; P[j] = tp[k]; }
          }
        }
      }
      if (lane == 0) { S[rp] = c; Q[rp] = qc; P[rp] = i; }
      __syncthreads();                                      // one wave: orders lane 0's entry before the next step's reads
    }
    if (lane < cnt) out[base + lane] = (u8)ov;
  }
}

__global__ void k_copy_len(const int32_t* a, int32_t* o, int32_t* flag, int B) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) { o[b] = a[b]; flag[b] = 1; }
}

size_t kz_sbrt_scratch(int B, int maxN) {
  const int T = (maxN + 64 + SB_TS - 1) / SB_TS + 1;
  return (size_t)B * T * 256 * sizeof(int2) + 4096;
}

int kz_stage_sbrt_forward(kz_ctx* ctx, kz_batch& bt, int mode) {
  const int B = bt.B;
  int maxN = 0;
  for (int b = 0; b < B; b++) if (bt.h_len[b] > maxN) maxN = bt.h_len[b];
  const int T = (maxN + SB_TS - 1) / SB_TS + 1;
  int2* tab = (int2*)kz_arena_alloc(ctx, (size_t)B * T * 256 * sizeof(int2));
  if (!tab) { snprintf(ctx->err, sizeof(ctx->err), "sbrt_forward: arena overflow"); return -KZ_ERR_DEVICE; }
  hipStream_t st = ctx->stream;
  const u8* src = bt.buf[bt.cur];
  u8* dst = bt.buf[bt.cur ^ 1];
  if (maxN > 0) {
    KZ_LAUNCH(ctx, KID_SBRT_LAST2, k_sbrt_last2, dim3(T, B), dim3(64), src, bt.stride, bt.d_len, tab, T);
    KZ_LAUNCH(ctx, KID_SBRT_SCAN, k_sbrt_scan, dim3(B), dim3(256), bt.d_len, tab, T);
    const bool keyed = maxN <= (1 << 23) && mode != 4 && ctx->sw.sbrtForm != 0;   // KZ_SBRT_FORM=0: the by-symbol replay of rounds 1-5 (A/B)
    switch (keyed ? mode + 10 : mode) {
      case 11: KZ_LAUNCH(ctx, KID_SBRT_REPLAY, k_sbrt_replay_keyed<1>, dim3(T, B), dim3(64), src, dst, bt.stride, bt.d_len, tab, T); break;
      case 12: KZ_LAUNCH(ctx, KID_SBRT_REPLAY, k_sbrt_replay_keyed<2>, dim3(T, B), dim3(64), src, dst, bt.stride, bt.d_len, tab, T); break;
      case 13: KZ_LAUNCH(ctx, KID_SBRT_REPLAY, k_sbrt_replay_keyed<3>, dim3(T, B), dim3(64), src, dst, bt.stride, bt.d_len, tab, T); break;
      case 1: KZ_LAUNCH(ctx, KID_SBRT_REPLAY, k_sbrt_replay<1>, dim3(T, B), dim3(64), src, dst, bt.stride, bt.d_len, tab, T); break;
      case 2: KZ_LAUNCH(ctx, KID_SBRT_REPLAY, k_sbrt_replay<2>, dim3(T, B), dim3(64), src, dst, bt.stride, bt.d_len, tab, T); break;
      case 4: KZ_LAUNCH(ctx, KID_SBRT_REPLAY, k_sbrt_replay<4>, dim3(T, B), dim3(64), src, dst, bt.stride, bt.d_len, tab, T); break;
      default: KZ_LAUNCH(ctx, KID_SBRT_REPLAY, k_sbrt_replay<3>, dim3(T, B), dim3(64), src, dst, bt.stride, bt.d_len, tab, T); break;
    }
  }
  KZ_LAUNCH(ctx, KID_COPY_LEN, k_copy_len, dim3((B + 255) / 256), dim3(256), bt.d_len, bt.d_len2, bt.d_flag, B);
  KZ_HIP(hipGetLastError());
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}

// ranks only (no length/flag bookkeeping): used by the SRT stage (mode 4)
int kz_sbrt_ranks(kz_ctx* ctx, const uint8_t* src, uint8_t* dst, int64_t stride, const int32_t* d_len, int B, int maxN, int mode) {
  const int T = (maxN + SB_TS - 1) / SB_TS + 1;
  int2* tab = (int2*)kz_arena_alloc(ctx, (size_t)B * T * 256 * sizeof(int2));
  if (!tab) { snprintf(ctx->err, sizeof(ctx->err), "sbrt_ranks: arena overflow"); return -KZ_ERR_DEVICE; }
  if (maxN > 0) {
    KZ_LAUNCH(ctx, KID_SBRT_LAST2, k_sbrt_last2, dim3(T, B), dim3(64), src, stride, d_len, tab, T);
    KZ_LAUNCH(ctx, KID_SBRT_SCAN, k_sbrt_scan, dim3(B), dim3(256), d_len, tab, T);
    switch (mode) {
      case 1: KZ_LAUNCH(ctx, KID_SBRT_REPLAY, k_sbrt_replay<1>, dim3(T, B), dim3(64), src, dst, stride, d_len, tab, T); break;
      case 2: KZ_LAUNCH(ctx, KID_SBRT_REPLAY, k_sbrt_replay<2>, dim3(T, B), dim3(64), src, dst, stride, d_len, tab, T); break;
      case 4: KZ_LAUNCH(ctx, KID_SBRT_REPLAY, k_sbrt_replay<4>, dim3(T, B), dim3(64), src, dst, stride, d_len, tab, T); break;
      default: KZ_LAUNCH(ctx, KID_SBRT_REPLAY, k_sbrt_replay<3>, dim3(T, B), dim3(64), src, dst, stride, d_len, tab, T); break;
    }
  }
  KZ_HIP(hipGetLastError());
  return 0;
}


// ---- placement of one-wave-per-block kernels ------------------------------------------------------
// The run time of a serial-per-block stage is proportional to a per-block cost the host knows (bt.h_cost: the length at
// the previous stage's input, e.g. the ZRLT-coded length ~ non-zero ranks), and two waves on one SIMD slow each other
// down.  Small batches get one wave per workgroup (spread over all CUs); large batches get 8 blocks per workgroup =
// per CU (wave w and w+4 share a SIMD), sorted so that every SIMD pairs an expensive block with a cheap one; 16 blocks
// per CU and more run as consecutive launches of 8 x CUs blocks (1.78 s vs 1.99 s for 4096 blocks).
int kz_place_blocks(kz_ctx* ctx, const kz_batch& bt, KzPlacement& PL) {
  const int B = bt.B;
  const int cus = ctx->numCUs > 0 ? ctx->numCUs : 256;
  const int wpg = B > 4 * cus ? 8 : (B > 2 * cus ? 4 : (B > cus ? 2 : 1));
  const int perLaunch = 8 * cus;
  const int R = (wpg == 8) ? std::max(1, B / perLaunch) : 1;       // only whole multiples pay: 1.5 x 8 x CUs is faster in one launch
  std::vector<int32_t> idx(B);
  for (int b = 0; b < B; b++) idx[b] = b;
  const bool haveCost = (wpg == 8) && ((int)bt.h_cost.size() == B);
  if (haveCost)
    std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) {
      const int64_t cx = bt.h_len[x] > 0 ? bt.h_cost[x] : -1, cy = bt.h_len[y] > 0 ? bt.h_cost[y] : -1;
      return cx > cy; });
  std::vector<int32_t> order;
  PL.wpg = wpg; PL.R = R; PL.G.assign(R, 0); PL.off.assign(R, 0);
  for (int rr = 0; rr < R; rr++) {
    const int nb = (B - rr + R - 1) / R;                            // blocks idx[rr], idx[rr + R], ... of this launch
    const int G = (nb + wpg - 1) / wpg;
    PL.G[rr] = G; PL.off[rr] = (int)order.size();
    order.resize(order.size() + (size_t)G * wpg, -1);
    int32_t* o = order.data() + PL.off[rr];
    for (int k = 0; k < nb; k++) {
      const int blk = idx[rr + k * R];
      if (haveCost) {
        const int p = k / G, q = k % G;
        const int wg = (p & 1) ? G - 1 - q : q;                    // snake over the workgroups
        const int slot0 = p < 4 ? p : 11 - p;                       // 4 most expensive on waves 0..3, the cheapest facing them
        const int slot = (slot0 & 4) | ((slot0 + bt.slotRot) & 3);  // wave w runs on SIMD w & 3: views in flight together start on different SIMDs
        o[(size_t)wg * 8 + slot] = blk;
      } else o[k] = blk;
    }
  }
  int32_t* d = (int32_t*)kz_arena_alloc(ctx, order.size() * 4);
  if (!d) { snprintf(ctx->err, sizeof(ctx->err), "placement: arena overflow"); return -KZ_ERR_DEVICE; }
  KZ_HIP(hipMemcpyAsync(d, order.data(), order.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  KZ_HIP(kz_stream_sync(ctx, ctx->stream));                        // order[] is a local
  PL.d_order = d;
  return 0;
}

int kz_stage_sbrt_inverse(kz_ctx* ctx, kz_batch& bt, int mode) {
  const int B = bt.B;
  hipStream_t st = ctx->stream;
  const u8* src = bt.buf[bt.cur];
  u8* dst = bt.buf[bt.cur ^ 1];
  bool wide = false;
  for (int b = 0; b < B; b++) wide |= bt.h_len[b] >= (1 << 24) - 256;
  if (wide) {                                                      // a block beyond the packed list's 24-bit timestamps: the plain form for the whole call
    if (mode == 2) { KZ_LAUNCH(ctx, KID_SBRT_INVERSE, k_sbrt_inverse_wide<2>, dim3(B), dim3(64), src, dst, bt.stride, bt.d_len, B); }
    else if (mode == 1) { KZ_LAUNCH(ctx, KID_SBRT_INVERSE, k_sbrt_inverse_wide<1>, dim3(B), dim3(64), src, dst, bt.stride, bt.d_len, B); }
    else { KZ_LAUNCH(ctx, KID_SBRT_INVERSE, k_sbrt_inverse_wide<3>, dim3(B), dim3(64), src, dst, bt.stride, bt.d_len, B); }
    KZ_LAUNCH(ctx, KID_COPY_LEN, k_copy_len, dim3((B + 255) / 256), dim3(256), bt.d_len, bt.d_len2, bt.d_flag, B);
    KZ_HIP(hipGetLastError());
    bt.cur ^= 1;
    { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
    return 0;
  }
  KzPlacement PL;
  { const int prc = kz_place_blocks(ctx, bt, PL); if (prc) return prc; }
  const int wpg = PL.wpg, R = PL.R;
  const std::vector<int>& launchG = PL.G;
  const std::vector<int>& launchOff = PL.off;
  const int32_t* d_order = PL.d_order;
  const int oldCold = 0;                                            // (bit 3 of prio = cold rows in the by-position layout: the round-3 A/B, closed)
  const int oldForms = ctx->sw.sbrtForm == 0 ? 16 : 0;              // KZ_SBRT_FORM=0 (prio bit 4): the 32-bit list forms of rounds 2-5 only (A/B)
  for (int rr = 0; rr < R; rr++) {
    const int G = launchG[rr];
    const int32_t* ord = d_order + launchOff[rr];
    if (mode == 2) { KZ_LAUNCH(ctx, KID_SBRT_INVERSE, k_sbrt_inverse<2>, dim3(G), dim3(64 * wpg), src, dst, bt.stride, bt.d_len, ord, wpg, bt.prio | oldCold | oldForms); }
    else if (mode == 1) { KZ_LAUNCH(ctx, KID_SBRT_INVERSE, k_sbrt_inverse<1>, dim3(G), dim3(64 * wpg), src, dst, bt.stride, bt.d_len, ord, wpg, bt.prio | oldCold | oldForms); }
    else { KZ_LAUNCH(ctx, KID_SBRT_INVERSE, k_sbrt_inverse<3>, dim3(G), dim3(64 * wpg), src, dst, bt.stride, bt.d_len, ord, wpg, bt.prio | oldCold | oldForms); }
  }
  KZ_LAUNCH(ctx, KID_COPY_LEN, k_copy_len, dim3((B + 255) / 256), dim3(256), bt.d_len, bt.d_len2, bt.d_flag, B);
  KZ_HIP(hipGetLastError());
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}
