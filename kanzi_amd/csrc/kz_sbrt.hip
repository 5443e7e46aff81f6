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
//   one wave64 per tile holds the 256 keys in 4 x u64 VGPRs (symbol s = reg s>>6, lane s&63);
//   a rank is 4 v_cmp_gt_u64 ballots + s_bcnt1.
// inverse (serial per block by nature: the list state depends on every decoded symbol): one wave
//   per block; the rank->symbol list lives in ONE VGPR (position j = lane j>>2, byte j&3) and a
//   move-up is a DPP wave shift + byte funnel; blocks of the batch decode concurrently.
#include "kz_device.h"
#include "kz_internal.h"

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
// tab[b][t][s] = (p1,p2) absolute positions, -1 = none
__global__ __launch_bounds__(64) void k_sbrt_last2(const u8* __restrict__ src, int64_t stride, const int32_t* __restrict__ d_len,
                                                    int2* __restrict__ tab, int T) {
  const int b = blockIdx.y, t = blockIdx.x;
  const int n = d_len[b];
  const int start = t * SB_TS;
  if (start >= n) return;
  const int end = min(n, start + SB_TS);
  const u8* s = src + (int64_t)b * stride;
  const int lane = kz_lane();
  int p1[4] = {-1, -1, -1, -1}, p2[4] = {-1, -1, -1, -1};
  for (int row = start; row < end; row += 256) {
    const int wi = row + lane * 4;
    u32 w = 0;
    if (wi + 3 < end) w = *(const u32*)(s + wi);
    else { for (int k = 0; k < 4; k++) if (wi + k < end) w |= (u32)s[wi + k] << (8 * k); }
    const int cnt = min(256, end - row);
    for (int j = 0; j < cnt; j++) {
      const u32 ww = (u32)__builtin_amdgcn_readlane((int)w, j >> 2);
      const int c = (ww >> (8 * (j & 3))) & 0xFF;
      const int i = row + j;
      const bool mine = lane == (c & 63);
      switch (c >> 6) {
        case 0: if (mine) { p2[0] = p1[0]; p1[0] = i; } break;
        case 1: if (mine) { p2[1] = p1[1]; p1[1] = i; } break;
        case 2: if (mine) { p2[2] = p1[2]; p1[2] = i; } break;
        default: if (mine) { p2[3] = p1[3]; p1[3] = i; } break;
      }
    }
  }
  int2* o = tab + ((int64_t)b * T + t) * 256;
#pragma unroll
  for (int q = 0; q < 4; q++) o[q * 64 + lane] = make_int2(p1[q], p2[q]);
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

#define KZ_SBRT_RANK_AND_UPDATE(CASE_Q)                                         \
  { const u64 kc = kz_readlane64(k##CASE_Q, c & 63);                            \
    cnt = (int)(__popcll(kz_ballot(k0 > kc)) + __popcll(kz_ballot(k1 > kc)) +   \
                __popcll(kz_ballot(k2 > kc)) + __popcll(kz_ballot(k3 > kc)));   \
    const u32 lo = (u32)kc;                                                     \
    const u32 pc = (lo >= 256u) ? lo - 256u : 0u;                               \
    const u32 qc = (mode == 2) ? (((u32)i + pc) >> 1) : ((mode == 1 || mode == 4) ? (u32)i : pc); \
    const u64 nk = ((u64)qc << 32) | (u64)((u32)i + 256u);                      \
    if (lane == (c & 63)) k##CASE_Q = nk; }

// ---- forward 3/3: replay one tile per wave ----------------------------------------------------
__global__ __launch_bounds__(64) void k_sbrt_replay(const u8* __restrict__ src, u8* __restrict__ dst, int64_t stride,
                                                     const int32_t* __restrict__ d_len, const int2* __restrict__ tab, int T, int mode) {
  const int b = blockIdx.y, t = blockIdx.x;
  const int n = d_len[b];
  const int start = t * SB_TS;
  if (start >= n) return;
  const int end = min(n, start + SB_TS);
  const u8* s = src + (int64_t)b * stride;
  u8* d = dst + (int64_t)b * stride;
  const int lane = kz_lane();
  const int2* pre = tab + ((int64_t)b * T + t) * 256;
  int2 v0 = pre[lane], v1 = pre[64 + lane], v2 = pre[128 + lane], v3 = pre[192 + lane];
  u64 k0 = kz_sbrt_key(mode, v0.x, v0.y, lane);
  u64 k1 = kz_sbrt_key(mode, v1.x, v1.y, 64 + lane);
  u64 k2 = kz_sbrt_key(mode, v2.x, v2.y, 128 + lane);
  u64 k3 = kz_sbrt_key(mode, v3.x, v3.y, 192 + lane);
  for (int row = start; row < end; row += 256) {
    const int wi = row + lane * 4;
    u32 w = 0;
    if (wi + 3 < end) w = *(const u32*)(s + wi);
    else { for (int k = 0; k < 4; k++) if (wi + k < end) w |= (u32)s[wi + k] << (8 * k); }
    const int cntRow = min(256, end - row);
    u32 outw = 0;
    u32 acc = 0;
    for (int j = 0; j < cntRow; j++) {
      const u32 ww = (u32)__builtin_amdgcn_readlane((int)w, j >> 2);
      const int c = (ww >> (8 * (j & 3))) & 0xFF;
      const int i = row + j;
      int cnt;
      switch (c >> 6) {
        case 0: KZ_SBRT_RANK_AND_UPDATE(0) break;
        case 1: KZ_SBRT_RANK_AND_UPDATE(1) break;
        case 2: KZ_SBRT_RANK_AND_UPDATE(2) break;
        default: KZ_SBRT_RANK_AND_UPDATE(3) break;
      }
      acc |= (u32)cnt << (8 * (j & 3));
      if ((j & 3) == 3 || j == cntRow - 1) { if (lane == (j >> 2)) outw = acc; acc = 0; }
    }
    if (wi + 3 < end) *(u32*)(d + wi) = outw;
    else { for (int k = 0; k < 4; k++) if (wi + k < end) d[wi + k] = (u8)(outw >> (8 * k)); }
  }
}

// ---- inverse: one wave per block ---------------------------------------------------------------
// State: the rank->symbol list kept sorted by position: position j lives in register j>>6, lane j&63
// as ONE 64-bit value  key' = (q << 40) | ((p + 256) << 8) | symbol  (never seen: ((255-s) << 8) | s).
// (q,p) is unique per symbol so the extra low byte never changes the order, and symbol + key move
// together: a step costs two readlanes, one v_cmp_gt_u64 ballot per register up to r>>6 (new position
// = number of keys above the new key) and one DPP wave_shr:1 per touched register half.  Zero ranks
// never move the list: runs of zeros are skipped in O(1) with a ballot of the non-zero lanes of each
// 64-byte row (after BWT most ranks are zero).  Valid for n < 2^24 - 256.
#define KZ_DPP_SHR1(x) ((u32)__builtin_amdgcn_update_dpp(0, (int)(x), 0x138 /*wave_shr:1*/, 0xF, 0xF, false))
#define KZ_K64(k) (((u64)hi##k << 32) | (u64)lo##k)

// shift register k (positions 64k..64k+63) for a move of position r up to rp; K1 = k-1 (carry source)
#define KZ_SBRT_SHIFT(k, K1, HAS_PREV)                                                     \
  if ((k) <= R && (k) >= RP) {                                                             \
    u32 slo = KZ_DPP_SHR1(lo##k), shi = KZ_DPP_SHR1(hi##k);                                \
    if (HAS_PREV && (k) > RP) {                                                            \
      const u32 clo = (u32)__builtin_amdgcn_readlane((int)lo##K1, 63);                     \
      const u32 chi = (u32)__builtin_amdgcn_readlane((int)hi##K1, 63);                     \
      if (lane == 0) { slo = clo; shi = chi; }                                             \
    }                                                                                      \
    const int pos = 64 * (k) + lane;                                                       \
    const bool in = (pos > rp) && (pos <= r);                                              \
    lo##k = in ? slo : lo##k; hi##k = in ? shi : hi##k;                                    \
    if (pos == rp) { lo##k = nlo; hi##k = nhi; }                                           \
  }

__global__ __launch_bounds__(64) void k_sbrt_inverse(const u8* __restrict__ src, u8* __restrict__ dst, int64_t stride,
                                                      const int32_t* __restrict__ d_len, int mode) {
  const int b = blockIdx.x;
  const int n = d_len[b];
  const u8* s = src + (int64_t)b * stride;
  u8* d = dst + (int64_t)b * stride;
  const int lane = kz_lane();
  // r2s[j] = j, all symbols never seen
  u32 lo0 = ((u32)(255 - lane) << 8) | (u32)lane, lo1 = ((u32)(191 - lane) << 8) | (u32)(64 + lane);
  u32 lo2 = ((u32)(127 - lane) << 8) | (u32)(128 + lane), lo3 = ((u32)(63 - lane) << 8) | (u32)(192 + lane);
  u32 hi0 = 0, hi1 = 0, hi2 = 0, hi3 = 0;
  u32 flo = (255u << 8) | 0u, fhi = 0;      // authoritative copy of position 0 (synced into lane 0 on demand)
  u32 cur = (lane < n) ? (u32)s[lane] : 0u;
  for (int row = 0; row < n; row += 64) {
    const int cnt = min(64, n - row);
    const int nrow = row + 64;
    const u32 nxt = (nrow + lane < n) ? (u32)s[nrow + lane] : 0u;  // prefetch the next row
    uint64_t nz = kz_ballot(cur != 0 && lane < cnt);
    u32 outv = 0;
    int prev = -1;
    for (;;) {
      const int j = nz ? (int)__builtin_ctzll(nz) : cnt;
      const int zr = j - prev - 1;
      if (zr > 0) {
        // zero run [prev+1, j): the front symbol repeats; only its (q,p) change (SBRT.java:194-201)
        const u32 pk = flo >> 8;
        const u32 pold = (pk >= 256u) ? pk - 256u : 0u;
        const u32 pl = (u32)(row + prev + zr);                     // last index of the run
        const u32 pp = (zr >= 2) ? pl - 1u : pold;
        const u32 q = (mode == 2) ? ((pl + pp) >> 1) : ((mode == 1) ? pl : pp);
        fhi = q << 8;
        flo = ((pl + 256u) << 8) | (flo & 0xFFu);
        if (lane > prev && lane < j) outv = flo & 0xFFu;
      }
      if (j >= cnt) break;
      nz &= nz - 1;
      const int r = __builtin_amdgcn_readlane((int)cur, j);
      const int i = row + j;
      if (lane == 0) { lo0 = flo; hi0 = fhi; }                     // sync the cached front entry
      const int R = r >> 6, rl = r & 63;
      u32 clo;
      if (R == 0) clo = (u32)__builtin_amdgcn_readlane((int)lo0, rl);
      else if (R == 1) clo = (u32)__builtin_amdgcn_readlane((int)lo1, rl);
      else if (R == 2) clo = (u32)__builtin_amdgcn_readlane((int)lo2, rl);
      else clo = (u32)__builtin_amdgcn_readlane((int)lo3, rl);
      const u32 c = clo & 0xFFu;
      const u32 pk = clo >> 8;
      const u32 pc = (pk >= 256u) ? pk - 256u : 0u;
      const u32 qc = (mode == 2) ? (((u32)i + pc) >> 1) : ((mode == 1) ? (u32)i : pc);
      const u32 nlo = (((u32)i + 256u) << 8) | c, nhi = qc << 8;
      const u64 nk = ((u64)nhi << 32) | (u64)nlo;
      // new position = number of entries above the new key (entries below r are smaller than the old key)
      int rp = (int)__popcll(kz_ballot(KZ_K64(0) > nk));
      if (R >= 1) rp += (int)__popcll(kz_ballot(KZ_K64(1) > nk));
      if (R >= 2) rp += (int)__popcll(kz_ballot(KZ_K64(2) > nk));
      if (R >= 3) rp += (int)__popcll(kz_ballot(KZ_K64(3) > nk));
      const int RP = rp >> 6;
      // descending order: register k reads its carry from the still unmodified register k-1
      KZ_SBRT_SHIFT(3, 2, true)
      KZ_SBRT_SHIFT(2, 1, true)
      KZ_SBRT_SHIFT(1, 0, true)
      KZ_SBRT_SHIFT(0, 0, false)
      if (rp == 0) { flo = nlo; fhi = nhi; }                       // else position 0 is untouched
      if (lane == j) outv = c;
      prev = j;
    }
    if (lane < cnt) d[row + lane] = (u8)outv;
    cur = nxt;
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
    KZ_LAUNCH(ctx, KID_SBRT_REPLAY, k_sbrt_replay, dim3(T, B), dim3(64), src, dst, bt.stride, bt.d_len, tab, T, mode);
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
    KZ_LAUNCH(ctx, KID_SBRT_REPLAY, k_sbrt_replay, dim3(T, B), dim3(64), src, dst, stride, d_len, tab, T, mode);
  }
  KZ_HIP(hipGetLastError());
  return 0;
}

int kz_stage_sbrt_inverse(kz_ctx* ctx, kz_batch& bt, int mode) {
  const int B = bt.B;
  for (int b = 0; b < B; b++) if (bt.h_len[b] >= (1 << 24) - 256) {
    snprintf(ctx->err, sizeof(ctx->err), "sbrt_inverse: block of %d bytes exceeds the packed-key limit 2^24-256", bt.h_len[b]);
    return -KZ_ERR_BLOCK_SIZE;
  }
  hipStream_t st = ctx->stream;
  const u8* src = bt.buf[bt.cur];
  u8* dst = bt.buf[bt.cur ^ 1];
  KZ_LAUNCH(ctx, KID_SBRT_INVERSE, k_sbrt_inverse, dim3(B), dim3(64), src, dst, bt.stride, bt.d_len, mode);
  KZ_LAUNCH(ctx, KID_COPY_LEN, k_copy_len, dim3((B + 255) / 256), dim3(256), bt.d_len, bt.d_len2, bt.d_flag, B);
  KZ_HIP(hipGetLastError());
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}
