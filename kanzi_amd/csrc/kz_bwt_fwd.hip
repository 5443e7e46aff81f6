// kz_bwt_fwd.hip -- forward BWT for a batch of blocks on gfx950.
//
// Replaces K/transform/BWTBlockCodec.java:71-128 + K/transform/BWT.java:148-191 +
// K/transform/DivSufSort.java:204-327.  The BWT of a string is unique, so instead of
// imitating DivSufSort's induced sorting (a serial pointer-heavy CPU algorithm) the suffixes are
// sorted the GPU way: prefix doubling where every round is a batched LSD radix sort
// (LDS-staged 256-bin histograms, wave64 ballot match-any ranking, coalesced tile I/O) over the
// still-unsorted suffixes only.  Output convention and the 8 primary indexes follow
// DivSufSort.java:217-224 and :233-325 (primary[k] = ISA[k*step] + 1).
//
// Per block b (n bytes), arrays live in HBM with stride NS elements:
//   key[2] u64, val[2] u32 (suffix index), cpos[2] u32 (SA slot of the compact element),
//   head[2] u8, rank u32 (= ISA as "group head slot"), sa u32.
// Round r keeps only suffixes whose group is not yet a singleton ("compact" arrays of size m_b).
#include "kz_device.h"
#include "kz_internal.h"

#define RS_ITEMS 16
#define RS_TILE (KZ_WG * RS_ITEMS)   // 4096 elements per workgroup

typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint8_t u8;

struct BwtArrays {
  u64* key[2]; u32* val[2]; u32* cpos[2]; u8* head[2]; u8* nhead;
  u32* rank; u32* sa;
  u32* tileHist;     // [B][T][256]
  u32* digitBase;    // [B][256]
  u32* tileA;        // [B][T] scan temporaries
  u32* tileB;        // [B][T]
  int32_t* d_n;      // [B] block length
  int32_t* d_m;      // [B] compact size (current)
  int32_t* d_m2;     // [B] compact size (next)
  int32_t* d_g;      // [B] group count among compact (next)
  int64_t NS;        // element stride per block
  int T;             // tile stride per block
};

// ---------------------------------------------------------------------------------------------
// round 0 keys: 7 data bytes (zero padded) + min(n-i,7): ties between a truncated suffix and a
// longer one resolve "shorter first" exactly like plain string comparison.
__global__ void k_bwt_init(const u8* __restrict__ src, int64_t srcStride, u64* keyC, u32* valC, u32* cposC, u8* headC, BwtArrays A) {
  const int b = blockIdx.y;
  const int n = A.d_n[b];
  const u8* s = src + (int64_t)b * srcStride;
  u64* key = keyC + (int64_t)b * A.NS;
  u32* val = valC + (int64_t)b * A.NS;
  u32* cpos = cposC + (int64_t)b * A.NS;
  u8* head = headC + (int64_t)b * A.NS;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    u64 k = 0;
    const int rem = n - i;
#pragma unroll
    for (int j = 0; j < 7; j++) k = (k << 8) | (u64)((j < rem) ? s[i + j] : 0);
    k = (k << 8) | (u64)(rem < 7 ? rem : 7);
    key[i] = k; val[i] = (u32)i; cpos[i] = (u32)i; head[i] = (i == 0) ? 1 : 0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { A.d_m[b] = n; }
}

// ---------------------------------------------------------------------------------------------
// radix pass 1/3: per-tile digit histogram (LDS, wave-aggregated via ballot match-any)
__global__ __launch_bounds__(KZ_WG) void k_radix_hist(const u64* __restrict__ keyIn, BwtArrays A, int shift) {
  const int b = blockIdx.y;
  const int m = A.d_m[b];
  const int tile = blockIdx.x;
  if ((int64_t)tile * RS_TILE >= m) return;
  __shared__ u32 hist[256];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const u64* key = keyIn + (int64_t)b * A.NS;
  const int base = tile * RS_TILE;
  const int lane = kz_lane();
#pragma unroll 4
  for (int r = 0; r < RS_ITEMS; r++) {
    const int idx = base + r * KZ_WG + threadIdx.x;
    const bool valid = idx < m;
    const u32 d = valid ? (u32)((key[idx] >> shift) & 0xFF) : 0;
    const uint64_t peers = kz_match8(d, valid);
    if (valid && (peers & kz_lanemask_lt()) == 0) atomicAdd(&hist[d], (u32)__popcll(peers));
  }
  (void)lane;
  __syncthreads();
  A.tileHist[((int64_t)b * A.T + tile) * 256 + threadIdx.x] = hist[threadIdx.x];
}

// radix pass 2/3: per block, thread d walks the tiles (coalesced across d) -> exclusive tile
// offsets per digit, then an exclusive scan over digit totals.
__global__ __launch_bounds__(256) void k_radix_scan(BwtArrays A) {
  const int b = blockIdx.x;
  const int m = A.d_m[b];
  const int tiles = (m + RS_TILE - 1) / RS_TILE;
  __shared__ u32 lds[32];
  u32* h = A.tileHist + (int64_t)b * A.T * 256;
  u32 run = 0;
  for (int t = 0; t < tiles; t++) {
    u32 v = h[(int64_t)t * 256 + threadIdx.x];
    h[(int64_t)t * 256 + threadIdx.x] = run;
    run += v;
  }
  u32 total;
  u32 ex = kz_wg_excl_sum(run, lds, &total);
  A.digitBase[b * 256 + threadIdx.x] = ex;
}

// radix pass 3/3: stable scatter.  Wave w owns the contiguous sub-tile [w*1024, (w+1)*1024);
// rows of 64 keys are ranked with ballot match-any against per-wave LDS digit counters.  The tile is
// then reordered IN LDS (keys, then values through the same 32 KiB buffer) so that consecutive threads
// store consecutive elements of each digit run: a wave store touches a few 128 B lines instead of up
// to 64 scattered 8 B / 4 B segments (the pass is bound by memory transactions, not bytes).
__global__ __launch_bounds__(KZ_WG) void k_radix_scatter(const u64* __restrict__ keyIn, const u32* __restrict__ valIn,
                                                          u64* __restrict__ keyOut, u32* __restrict__ valOut,
                                                          BwtArrays A, int shift) {
  const int b = blockIdx.y;
  const int m = A.d_m[b];
  const int tile = blockIdx.x;
  if ((int64_t)tile * RS_TILE >= m) return;
  __shared__ u32 cnt[4][256];
  __shared__ u32 gdelta[256];        // global slot of the digit's first element of this tile - its tile-local slot
  __shared__ u32 scan[32];
  __shared__ u64 stage[RS_TILE];     // 32 KiB: keys, then values
  for (int i = threadIdx.x; i < 1024; i += KZ_WG) (&cnt[0][0])[i] = 0;
  __syncthreads();
  const int64_t off = (int64_t)b * A.NS;
  const int wave = threadIdx.x >> 6;
  const int lane = kz_lane();
  const int tbase = tile * RS_TILE;
  const int base = tbase + wave * (64 * RS_ITEMS);
  const int tcount = min(RS_TILE, m - tbase);
  const uint64_t lt = kz_lanemask_lt();
  u64 k[RS_ITEMS]; u32 v[RS_ITEMS]; u32 dr[RS_ITEMS];   // dr = digit | (rank<<8), later the tile-local slot
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const int idx = base + r * 64 + lane;
    const bool valid = idx < m;
    k[r] = valid ? keyIn[off + idx] : 0;
    v[r] = valid ? valIn[off + idx] : 0;
  }
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const int idx = base + r * 64 + lane;
    const bool valid = idx < m;
    const u32 d = (u32)((k[r] >> shift) & 0xFF);
    const uint64_t peers = kz_match8(d, valid);
    u32 pre = 0;
    if (valid) pre = cnt[wave][d];
    const u32 rnk = pre + (u32)__popcll(peers & lt);
    // the highest peer lane publishes the new count (all peers read `pre` before: same wave, in order)
    if (valid && (peers >> lane) == 1ULL) cnt[wave][d] = pre + (u32)__popcll(peers);
    dr[r] = d | (rnk << 8);
  }
  __syncthreads();
  {
    const int d = threadIdx.x;
    const u32 c0 = cnt[0][d], c1 = cnt[1][d], c2 = cnt[2][d], c3 = cnt[3][d];
    u32 total;
    const u32 ts = kz_wg_excl_sum(c0 + c1 + c2 + c3, scan, &total);
    gdelta[d] = A.digitBase[b * 256 + d] + A.tileHist[((int64_t)b * A.T + tile) * 256 + d] - ts;
    cnt[0][d] = ts; cnt[1][d] = ts + c0; cnt[2][d] = ts + c0 + c1; cnt[3][d] = ts + c0 + c1 + c2;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const int idx = base + r * 64 + lane;
    if (idx < m) { const u32 slot = cnt[wave][dr[r] & 0xFF] + (dr[r] >> 8); dr[r] = slot; stage[slot] = k[r]; }
  }
  __syncthreads();
  u32 gp[RS_ITEMS];
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const int slot = r * KZ_WG + threadIdx.x;
    if (slot < tcount) {
      const u64 kk = stage[slot];
      gp[r] = gdelta[(u32)((kk >> shift) & 0xFF)] + (u32)slot;
      keyOut[off + gp[r]] = kk;
    }
  }
  __syncthreads();
  u32* stageV = (u32*)stage;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const int idx = base + r * 64 + lane;
    if (idx < m) stageV[dr[r]] = v[r];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const int slot = r * KZ_WG + threadIdx.x;
    if (slot < tcount) valOut[off + gp[r]] = stageV[slot];
  }
}

// ---------------------------------------------------------------------------------------------
// new group heads after a sort: old head (positional) or key differs from the predecessor
__global__ void k_bwt_newhead(const u64* __restrict__ keyS, const u8* __restrict__ headOld, BwtArrays A) {
  const int b = blockIdx.y;
  const int m = A.d_m[b];
  const int64_t off = (int64_t)b * A.NS;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < m; c += gridDim.x * blockDim.x) {
    u8 h = headOld[off + c];
    if (!h && c > 0) h = (keyS[off + c] != keyS[off + c - 1]) ? 1 : 0;
    if (c == 0) h = 1;
    A.nhead[off + c] = h;
  }
}

// scan A (max): per tile, the last head slot (c+1) in the tile
__global__ __launch_bounds__(KZ_WG) void k_hp_reduce(BwtArrays A) {
  const int b = blockIdx.y;
  const int m = A.d_m[b];
  const int tile = blockIdx.x;
  if ((int64_t)tile * RS_TILE >= m) return;
  __shared__ u32 lds[32];
  const int64_t off = (int64_t)b * A.NS;
  const int base = tile * RS_TILE + threadIdx.x * RS_ITEMS;
  u32 mx = 0;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) { const int c = base + r; if (c < m && A.nhead[off + c]) mx = (u32)c + 1; }
  u32 total; kz_wg_incl_max(mx, lds, &total);
  if (threadIdx.x == 0) A.tileA[(int64_t)b * A.T + tile] = total;
}
// per block: exclusive max-scan over tiles (serial per block: <= 1024 tiles) -- one wave per block
__global__ void k_hp_scan(BwtArrays A) {
  const int b = blockIdx.x;
  const int m = A.d_m[b];
  const int tiles = (m + RS_TILE - 1) / RS_TILE;
  u32* t = A.tileA + (int64_t)b * A.T;
  u32 carry = 0;
  for (int base = 0; base < tiles; base += 64) {
    const int i = base + threadIdx.x;
    u32 v = (i < tiles) ? t[i] : 0;
    u32 inc = kz_wave_incl_max(v);
    u32 exc = __shfl_up(inc, 1, 64); if (threadIdx.x == 0) exc = 0;
    exc = exc > carry ? exc : carry;
    if (i < tiles) t[i] = exc;
    u32 last = __shfl(inc, 63, 64);
    carry = carry > last ? carry : last;
  }
}
// apply: rank[val[c]] = cpos[headslot]
__global__ __launch_bounds__(KZ_WG) void k_hp_apply(const u32* __restrict__ valS, const u32* __restrict__ cposOld, BwtArrays A) {
  const int b = blockIdx.y;
  const int m = A.d_m[b];
  const int tile = blockIdx.x;
  if ((int64_t)tile * RS_TILE >= m) return;
  __shared__ u32 lds[32];
  const int64_t off = (int64_t)b * A.NS;
  const int base = tile * RS_TILE + threadIdx.x * RS_ITEMS;
  u32 loc[RS_ITEMS];
  u32 mx = 0;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) { const int c = base + r; if (c < m && A.nhead[off + c]) mx = (u32)c + 1; loc[r] = mx; }
  u32 total;
  u32 inc = kz_wg_incl_max(mx, lds, &total);
  // exclusive prefix for this thread = max over previous threads
  u32 prevT = __shfl_up(inc, 1, 64);
  __shared__ u32 wlast[4];
  if (kz_lane() == 63) wlast[threadIdx.x >> 6] = inc;
  __syncthreads();
  if (kz_lane() == 0) prevT = (threadIdx.x >> 6) ? wlast[(threadIdx.x >> 6) - 1] : 0;
  const u32 carry = A.tileA[(int64_t)b * A.T + tile];
  u32 pre = prevT > carry ? prevT : carry;
  u32* rank = A.rank + off;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const int c = base + r;
    if (c < m) {
      u32 hs = loc[r] > pre ? loc[r] : pre;     // head slot + 1 (always >= 1 since c==0 is a head)
      rank[valS[off + c]] = cposOld[off + hs - 1];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// filter: keep[c] = element's new group is not a singleton.  Two counts are scanned together:
// kept elements and kept heads (-> group ordinal).  Dropped elements are final: write SA.
__device__ __forceinline__ bool kz_keep(const u8* nh, int c, int m) {
  const bool h0 = nh[c] != 0;
  const bool h1 = (c + 1 >= m) ? true : (nh[c + 1] != 0);
  return !(h0 && h1);
}
__global__ __launch_bounds__(KZ_WG) void k_flt_reduce(BwtArrays A) {
  const int b = blockIdx.y;
  const int m = A.d_m[b];
  const int tile = blockIdx.x;
  if ((int64_t)tile * RS_TILE >= m) return;
  __shared__ u32 lds[32];
  const u8* nh = A.nhead + (int64_t)b * A.NS;
  const int base = tile * RS_TILE + threadIdx.x * RS_ITEMS;
  u32 ck = 0, ch = 0;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const int c = base + r;
    if (c < m && kz_keep(nh, c, m)) { ck++; if (nh[c]) ch++; }
  }
  u32 tk, th;
  kz_wg_excl_sum(ck, lds, &tk);
  kz_wg_excl_sum(ch, lds, &th);
  if (threadIdx.x == 0) { A.tileA[(int64_t)b * A.T + tile] = tk; A.tileB[(int64_t)b * A.T + tile] = th; }
}
__global__ void k_flt_scan(BwtArrays A) {
  const int b = blockIdx.x;
  const int m = A.d_m[b];
  const int tiles = (m + RS_TILE - 1) / RS_TILE;
  u32* ta = A.tileA + (int64_t)b * A.T;
  u32* tb = A.tileB + (int64_t)b * A.T;
  u32 ca = 0, cb = 0;
  for (int base = 0; base < tiles; base += 64) {
    const int i = base + threadIdx.x;
    u32 va = (i < tiles) ? ta[i] : 0, vb = (i < tiles) ? tb[i] : 0;
    u32 ia = kz_wave_incl_sum(va), ib = kz_wave_incl_sum(vb);
    if (i < tiles) { ta[i] = ca + ia - va; tb[i] = cb + ib - vb; }
    ca += __shfl(ia, 63, 64); cb += __shfl(ib, 63, 64);
  }
  if (threadIdx.x == 0) { A.d_m2[b] = (int32_t)ca; A.d_g[b] = (int32_t)cb; }
}
// apply: compaction + next-round key gather (rank[] is final for this round: k_hp_apply ran before)
__global__ __launch_bounds__(KZ_WG) void k_flt_apply(const u32* __restrict__ valS, const u32* __restrict__ cposOld,
                                                      u64* __restrict__ keyN, u32* __restrict__ valN,
                                                      u32* __restrict__ cposN, u8* __restrict__ headN,
                                                      BwtArrays A, int hNext, int bitsR) {
  const int b = blockIdx.y;
  const int m = A.d_m[b];
  const int n = A.d_n[b];
  const int tile = blockIdx.x;
  if ((int64_t)tile * RS_TILE >= m) return;
  __shared__ u32 lds[32];
  const int64_t off = (int64_t)b * A.NS;
  const u8* nh = A.nhead + off;
  const int base = tile * RS_TILE + threadIdx.x * RS_ITEMS;
  u32 ck = 0, ch = 0;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const int c = base + r;
    if (c < m && kz_keep(nh, c, m)) { ck++; if (nh[c]) ch++; }
  }
  u32 tk, th;
  u32 ek = kz_wg_excl_sum(ck, lds, &tk);
  u32 eh = kz_wg_excl_sum(ch, lds, &th);
  ek += A.tileA[(int64_t)b * A.T + tile];
  eh += A.tileB[(int64_t)b * A.T + tile];
  const u32* rank = A.rank + off;
  u32* sa = A.sa + off;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const int c = base + r;
    if (c >= m) break;
    const u32 sv = valS[off + c];
    const u32 cp = cposOld[off + c];
    if (kz_keep(nh, c, m)) {
      const bool hd = nh[c] != 0;
      if (hd) eh++;
      const u32 gord = eh - 1;               // ordinal of this element's group among kept groups
      const int64_t j = (int64_t)sv + hNext;
      const u64 r2 = (j < n) ? (u64)rank[j] + 1ULL : 0ULL;
      keyN[off + ek] = ((u64)gord << bitsR) | r2;
      valN[off + ek] = sv;
      cposN[off + ek] = cp;
      headN[off + ek] = hd ? 1 : 0;
      ek++;
    } else {
      sa[cp] = sv;                           // singleton group: final position
    }
  }
}

// ---------------------------------------------------------------------------------------------
// emit: header + BWT bytes (BWTBlockCodec.java:90-126, DivSufSort.java:217-224)
__global__ void k_bwt_emit(const u8* __restrict__ src, int64_t srcStride, u8* __restrict__ dst, int64_t dstStride,
                           BwtArrays A, int32_t* d_lenOut, int32_t* d_flag) {
  const int b = blockIdx.y;
  const int n = A.d_n[b];
  const u8* s = src + (int64_t)b * srcStride;
  u8* d = dst + (int64_t)b * dstStride;
  const int64_t off = (int64_t)b * A.NS;
  if (n < 2) {   // n==1: pIndexSize==0 -> BWTBlockCodec declines (BWTBlockCodec.java:95-98); n==0 no-op
    if (blockIdx.x == 0 && threadIdx.x == 0) { d_flag[b] = 0; d_lenOut[b] = n; if (n == 1) d[0] = s[0]; }
    return;
  }
  int logBlockSize = kz_ilog2((u32)n);
  if ((n & (n - 1)) != 0) logBlockSize++;
  const int pIndexSize = (logBlockSize + 7) >> 3;
  const int chunks = (n < 256) ? 1 : 8;
  const int hdr = 1 + chunks * pIndexSize;
  const u32* rank = A.rank + off;
  const u32* sa = A.sa + off;
  const int p = (int)rank[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int logNbChunks = (chunks == 8) ? 3 : 0;
    d[0] = (u8)((logNbChunks << 2) | (pIndexSize - 1));
    const int st = n / chunks;
    const int step = (st * chunks != n) ? st + 1 : st;
    int idx = 1;
    for (int k = 0; k < chunks; k++) {
      const int pi = (int)rank[(int64_t)k * step];       // primary[k]-1 = ISA[k*step]
      for (int shift = (pIndexSize - 1) << 3; shift >= 0; shift -= 8) d[idx++] = (u8)(pi >> shift);
    }
    d[hdr] = s[n - 1];
    d_lenOut[b] = hdr + n;
    d_flag[b] = 1;
  }
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    if (j == p) continue;
    const u32 sv = sa[j];
    const u8 c = s[sv - 1];
    d[hdr + ((j < p) ? j + 1 : j)] = c;
  }
}

// ---------------------------------------------------------------------------------------------
size_t kz_bwt_forward_scratch(int B, int maxN) {
  const int64_t NS = (int64_t)kz_align((size_t)maxN, RS_TILE);
  const int T = (int)(NS / RS_TILE);
  size_t per = (size_t)NS * (8 * 2 + 4 * 2 + 4 * 2 + 1 * 3 + 4 + 4) + (size_t)T * (256 * 4 + 8) + 256 * 4 + 64;
  return kz_align(per * (size_t)B + 4096 * 16, 4096) + (1 << 20);
}

static inline int gridFor(int n, int per) { int g = (n + per - 1) / per; return g < 1 ? 1 : g; }

int kz_stage_bwt_forward(kz_ctx* ctx, kz_batch& bt) {
  const int B = bt.B;
  int maxN = 0;
  for (int b = 0; b < B; b++) if (bt.h_len[b] > maxN) maxN = bt.h_len[b];
  const int64_t NS = (int64_t)kz_align((size_t)(maxN > 0 ? maxN : 1), RS_TILE);
  const int T = (int)(NS / RS_TILE);
  BwtArrays A;
  A.NS = NS; A.T = T;
  for (int i = 0; i < 2; i++) {
    A.key[i] = (u64*)kz_arena_alloc(ctx, (size_t)NS * B * 8);
    A.val[i] = (u32*)kz_arena_alloc(ctx, (size_t)NS * B * 4);
    A.cpos[i] = (u32*)kz_arena_alloc(ctx, (size_t)NS * B * 4);
    A.head[i] = (u8*)kz_arena_alloc(ctx, (size_t)NS * B);
  }
  A.nhead = (u8*)kz_arena_alloc(ctx, (size_t)NS * B);
  A.rank = (u32*)kz_arena_alloc(ctx, (size_t)NS * B * 4);
  A.sa = (u32*)kz_arena_alloc(ctx, (size_t)NS * B * 4);
  A.tileHist = (u32*)kz_arena_alloc(ctx, (size_t)T * B * 256 * 4);
  A.digitBase = (u32*)kz_arena_alloc(ctx, (size_t)B * 256 * 4);
  A.tileA = (u32*)kz_arena_alloc(ctx, (size_t)T * B * 4);
  A.tileB = (u32*)kz_arena_alloc(ctx, (size_t)T * B * 4);
  A.d_m = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  A.d_m2 = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  A.d_g = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  if (!A.d_g || !A.tileB || !A.sa) { snprintf(ctx->err, sizeof(ctx->err), "bwt_forward: arena overflow"); return -KZ_ERR_DEVICE; }
  A.d_n = bt.d_len;
  hipStream_t st = ctx->stream;
  const u8* src = bt.buf[bt.cur];
  u8* dst = bt.buf[bt.cur ^ 1];

  int bitsR = 1;
  while ((1LL << bitsR) < (int64_t)maxN + 2) bitsR++;

  u64 *kC = A.key[0], *kF = A.key[1];
  u32 *vC = A.val[0], *vF = A.val[1];
  u32 *cposC = A.cpos[0], *cposF = A.cpos[1];
  u8 *headC = A.head[0], *headF = A.head[1];
  KZ_LAUNCH(ctx, KID_BWT_INIT, k_bwt_init, dim3(gridFor(maxN, 256 * 8), B), dim3(256), src, bt.stride, kC, vC, cposC, headC, A);
  int mMax = maxN, gMax = 1;
  int h = 0;
  for (int round = 0; round < 64 && mMax > 0; round++) {
    // ---- sort (kC,vC): LSD radix, 8-bit digits, ping-pong with the free pair ----
    const int nbits = (round == 0) ? 64 : bitsR + (gMax > 1 ? (32 - __builtin_clz((unsigned)(gMax - 1))) : 0);
    const int passes = (nbits + 7) / 8;
    const int tiles = gridFor(mMax, RS_TILE);
    for (int p = 0; p < passes; p++) {
      KZ_LAUNCH(ctx, KID_RADIX_HIST, k_radix_hist, dim3(tiles, B), dim3(KZ_WG), kC, A, p * 8);
      KZ_LAUNCH(ctx, KID_RADIX_SCAN, k_radix_scan, dim3(B), dim3(256), A);
      KZ_LAUNCH(ctx, KID_RADIX_SCATTER, k_radix_scatter, dim3(tiles, B), dim3(KZ_WG), kC, vC, kF, vF, A, p * 8);
      u64* tk = kC; kC = kF; kF = tk;
      u32* tv = vC; vC = vF; vF = tv;
    }
    // sorted data in (kC, vC); (kF, vF) is free
    KZ_LAUNCH(ctx, KID_BWT_NEWHEAD, k_bwt_newhead, dim3(gridFor(mMax, 256 * 8), B), dim3(256), kC, headC, A);
    KZ_LAUNCH(ctx, KID_HP_REDUCE, k_hp_reduce, dim3(tiles, B), dim3(KZ_WG), A);
    KZ_LAUNCH(ctx, KID_HP_SCAN, k_hp_scan, dim3(B), dim3(64), A);
    KZ_LAUNCH(ctx, KID_HP_APPLY, k_hp_apply, dim3(tiles, B), dim3(KZ_WG), vC, cposC, A);
    KZ_LAUNCH(ctx, KID_FLT_REDUCE, k_flt_reduce, dim3(tiles, B), dim3(KZ_WG), A);
    KZ_LAUNCH(ctx, KID_FLT_SCAN, k_flt_scan, dim3(B), dim3(64), A);
    h = (round == 0) ? 7 : h * 2;
    KZ_LAUNCH(ctx, KID_FLT_APPLY, k_flt_apply, dim3(tiles, B), dim3(KZ_WG), vC, cposC, kF, vF, cposF, headF, A, h, bitsR);
    { u64* tk = kC; kC = kF; kF = tk; u32* tv = vC; vC = vF; vF = tv; }
    { u32* tc = cposC; cposC = cposF; cposF = tc; u8* th = headC; headC = headF; headF = th; }
    // ---- read back sizes (next compact size / group count per block) ----
    KZ_HIP(hipMemcpyAsync(ctx->hpin, A.d_m2, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    KZ_HIP(hipMemcpyAsync(ctx->hpin + B, A.d_g, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    KZ_HIP(hipStreamSynchronize(st));
    mMax = 0; gMax = 1;
    for (int b = 0; b < B; b++) { if (ctx->hpin[b] > mMax) mMax = ctx->hpin[b]; if (ctx->hpin[B + b] > gMax) gMax = ctx->hpin[B + b]; }
    int32_t* tm = A.d_m; A.d_m = A.d_m2; A.d_m2 = tm;
  }
  if (mMax > 0) { snprintf(ctx->err, sizeof(ctx->err), "bwt_forward: suffix sort did not converge"); return -KZ_ERR_PROCESS_BLOCK; }
  KZ_LAUNCH(ctx, KID_BWT_EMIT, k_bwt_emit, dim3(gridFor(maxN, 256 * 8), B), dim3(256), src, bt.stride, dst, bt.stride, A, bt.d_len2, bt.d_flag);
  KZ_HIP(hipGetLastError());
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}
