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
//   key[2] u64, val[2] u32 (suffix index): the compact (still unsorted) suffixes, ping-pong for the radix passes
//   rank u32: ISA as "head slot of the suffix's group", bit 31 = LIVE (group not yet a singleton);  sa u32.
// A round with step h:
//   1. text order (coalesced): every LIVE suffix s emits key = (rank[s] << bitsR) | (rank[s+h] + 1), val = s.
//      Reading rank[s+h] in TEXT order makes both reads sequential; gathering it in SA order (the textbook
//      formulation, and the first version of this file) costs a 128 B line per 4-byte rank and was the largest
//      HBM consumer of the whole encoder (114 GB fetched per launch).  Scanning all n ranks each round costs less
//      than gathering for n/16 live suffixes.
//   2. radix sort of the compact pairs (old group = high key bits, so groups stay contiguous).
//   3. SA order: the members of an old group g occupy the slots g, g+1, ... in sorted order; a new group starts
//      wherever the key changes; rank[val] = new head slot | LIVE, and suffixes whose new group is a singleton are
//      final: sa[slot] = val.  Two max-scans (old-group start index, new-group start index) give the slots.
#include "kz_device.h"
#include "kz_internal.h"
#include <stdlib.h>
#include <algorithm>

#define RS_ITEMS 16
#define RS_TILE (KZ_WG * RS_ITEMS)   // 4096 elements per workgroup
#define BW_LIVE 0x80000000u
#define RSORT_TILE 16384               // keys per radix tile: a digit run of a tile averages 64 keys = 512 B
#define RSORT_ITEMS (RSORT_TILE / KZ_WG)

typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint8_t u8;
typedef u64 __attribute__((aligned(1))) bw_u64_unaligned;

struct BwtArrays {
  u64* key[2]; u32* val[2];
  u32* rank; u32* sa;
  u32* tileHist;     // [B][radix tiles][256 | MSD_BINS]
  u32* digitBase;    // [B][MSD_BINS]
  u32* bucketCnt;    // [B][MSD_BINS] elements per bucket of the current round (bit 31: too large for the LDS sort)
  int32_t* d_w;      // [B] window start: the LSD passes and the k_seg_* kernels work on [d_w, d_m) of the compact arrays
  int32_t* d_big;    // [B] d_m - d_w after k_msd_scan
  u32* tileA;        // [B][T] scan temporaries
  u32* tileB;        // [B][T]
  u32* tileLive;     // [B][T] live suffixes of the text tile in the previous round (0 stays 0: suffixes only become final)
  int32_t* d_n;      // [B] block length
  int32_t* d_m;      // [B] compact size (current)
  int32_t* d_m2;     // [B] compact size (next)
  int64_t NS;        // element stride per block
  int T;             // tile stride per block
  int64_t HS;        // tileHist stride per block: radix tiles x MSD_BINS
  const int32_t* act; // later rounds: the blocks that still have live suffixes, one grid row each (nullptr: row = block)
};
// A block whose suffixes are all final takes no part in the later rounds: their grids have one row per block that is still live
// (a mixed batch used to carry the uniform / geometric blocks' empty workgroups through every round).
__device__ __forceinline__ int bw_row_block(const BwtArrays& A, unsigned row) { return A.act ? A.act[row] : (int)row; }

// ---------------------------------------------------------------------------------------------
// round 0 keys: the first 7 bytes, zero padded at the end of the text.  A truncated suffix can therefore share a
// group with suffixes that continue with real zero bytes; it is a prefix of all of them, must sort first, and
// does: k_live_emit gives positions past the end of the text the smallest, distinct secondary keys.
__device__ __forceinline__ u64 bw_text_key(const u8* __restrict__ s, int i, int n, int K) {
  // first K bytes as a big-endian number: one unaligned 8-byte load (blocks have >= 4 KiB of slack behind them)
  u64 k = __builtin_bswap64(*(const bw_u64_unaligned*)(s + i)) >> (8 * (8 - K));
  const int rem = n - i;
  if (rem < K) k &= ~0ULL << (8 * (K - rem));                       // zero padding at the end of the text
  return k;
}
// The first LSD pass of round 0 reads the text itself (TextSrc below): the pairs (key, suffix) are never written in
// their initial order, which saves a 12-byte write and a 20-byte read per input byte.
struct TextSrc { const u8* src; int64_t stride; int K; };
__global__ void k_bwt_init(BwtArrays A, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) { A.d_m[b] = A.d_n[b]; A.d_w[b] = 0; }
}

// ---------------------------------------------------------------------------------------------
// radix pass 1/3: per-tile digit histogram (LDS, wave-aggregated via ballot match-any).  BINS = 256 for the LSD passes,
// MSD_BINS for the bucket partition of the later rounds (k_msd_*).
#define MSD_BINS 1024
template <int BINS, bool MSD, bool TEXT>
__device__ __forceinline__ void radix_hist_body(const u64* __restrict__ keyIn, const BwtArrays& A, int shift, TextSrc X) {
  const int b = bw_row_block(A, blockIdx.y);
  const int w0 = MSD ? 0 : A.d_w[b];
  const int m = A.d_m[b] - w0;
  const int tile = blockIdx.x;
  if ((int64_t)tile * RSORT_TILE >= m) return;
  __shared__ u32 hist[BINS];
  for (int i = threadIdx.x; i < BINS; i += KZ_WG) hist[i] = 0;
  __syncthreads();
  const u64* key = keyIn + (int64_t)b * A.NS + w0;
  const int base = tile * RSORT_TILE;
  const int lane = kz_lane();
#pragma unroll 4
  for (int r = 0; r < RSORT_ITEMS; r++) {
    const int idx = base + r * KZ_WG + threadIdx.x;
    const bool valid = idx < m;
    const u32 d = valid ? (u32)(((TEXT ? bw_text_key(X.src + (int64_t)b * X.stride, idx, m, X.K) : key[idx]) >> shift) & (BINS - 1)) : 0;
    // skewed digits (one value for the whole row: high key bytes, runs) would serialise 64 LDS atomics on one
    // address: those rows add once; rows with mixed digits use plain LDS atomics (conflicts only on equal digits)
    const uint64_t vm = kz_ballot(valid);
    const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)d);
    if (vm != 0 && kz_ballot(valid && d == d0) == vm) { if (lane == (int)__builtin_ctzll(vm)) atomicAdd(&hist[d0], (u32)__popcll(vm)); }
    else if (valid) atomicAdd(&hist[d], 1u);
  }
  __syncthreads();
  u32* th = A.tileHist + (int64_t)b * A.HS + (int64_t)tile * BINS;
  for (int i = threadIdx.x; i < BINS; i += KZ_WG) th[i] = hist[i];
}
__global__ __launch_bounds__(KZ_WG) void k_radix_hist(const u64* __restrict__ keyIn, BwtArrays A, int shift) { radix_hist_body<256, false, false>(keyIn, A, shift, TextSrc{nullptr, 0, 0}); }
__global__ __launch_bounds__(KZ_WG) void k_radix_hist0(BwtArrays A, TextSrc X) { radix_hist_body<256, false, true>(nullptr, A, 0, X); }
__global__ __launch_bounds__(KZ_WG) void k_msd_hist(const u64* __restrict__ keyIn, BwtArrays A, int shift) { radix_hist_body<MSD_BINS, true, false>(keyIn, A, shift, TextSrc{nullptr, 0, 0}); }

// radix pass 2/3: per block, thread d walks the tiles (coalesced across d) -> exclusive tile
// offsets per digit, then an exclusive scan over digit totals.
__global__ __launch_bounds__(256) void k_radix_scan(BwtArrays A) {
  const int b = bw_row_block(A, blockIdx.x);
  const int m = A.d_m[b] - A.d_w[b];
  const int tiles = (m + RSORT_TILE - 1) / RSORT_TILE;
  __shared__ u32 lds[32];
  u32* h = A.tileHist + (int64_t)b * A.HS;
  u32 run = 0;
  for (int t = 0; t < tiles; t++) {
    u32 v = h[(int64_t)t * 256 + threadIdx.x];
    h[(int64_t)t * 256 + threadIdx.x] = run;
    run += v;
  }
  u32 total;
  u32 ex = kz_wg_excl_sum(run, lds, &total);
  A.digitBase[b * MSD_BINS + threadIdx.x] = ex;
}

// Bucket partition of a later round (k_msd_hist / k_msd_scan / k_msd_scatter): the top bits of the OLD GROUP (a slot of
// the suffix array) split the compact list into buckets of BK_SLOTS slots; a bucket holds at most as many live suffixes
// as it has slots (plus the overhang of its last group), and old groups never span buckets, so a bucket that fits the
// LDS is sorted and applied by one workgroup (k_bucket_sort): one read of the pair instead of six LSD passes.
// Buckets above BK_CAP elements (long runs: one old group of many thousand suffixes) are placed BEHIND the small ones,
// in bucket order, and that window [d_w, d_m) goes through the LSD passes and k_seg_* as before.
#define BK_BITS 12
#define BK_CAP 6144            // 4096 slots + an overhang of 2048: 48 KiB of LDS, two workgroups per CU
#define BK_BIG 0x80000000u
__global__ __launch_bounds__(MSD_BINS) void k_msd_scan(BwtArrays A) {
  const int b = bw_row_block(A, blockIdx.x);
  const int m = A.d_m[b];
  const int tiles = (m + RSORT_TILE - 1) / RSORT_TILE;
  __shared__ u32 lds[32];
  u32* h = A.tileHist + (int64_t)b * A.HS;
  u32 run = 0;
  for (int t = 0; t < tiles; t++) {
    u32 v = h[(int64_t)t * MSD_BINS + threadIdx.x];
    h[(int64_t)t * MSD_BINS + threadIdx.x] = run;
    run += v;
  }
  const bool big = run > BK_CAP;
  u32 mSmall, mBig;
  const u32 exS = kz_wg_excl_sum(big ? 0u : run, lds, &mSmall);
  __syncthreads();
  const u32 exB = kz_wg_excl_sum(big ? run : 0u, lds, &mBig);
  A.digitBase[b * MSD_BINS + threadIdx.x] = big ? mSmall + exB : exS;
  A.bucketCnt[b * MSD_BINS + threadIdx.x] = big ? (run | BK_BIG) : run;
  if (threadIdx.x == 0) { A.d_w[b] = (int32_t)mSmall; A.d_big[b] = (int32_t)mBig; }
}

// radix pass 3/3: stable scatter.  Wave w owns the contiguous sub-tile [w*1024, (w+1)*1024);
// rows of 64 keys are ranked with ballot match-any against per-wave LDS digit counters.  The tile is
// then reordered IN LDS (keys, then values through the same 32 KiB buffer) so that consecutive threads
// store consecutive elements of each digit run: a wave store touches a few 128 B lines instead of up
// to 64 scattered 8 B / 4 B segments (the pass is bound by memory transactions, not bytes).
#define RSC_WAVES 16                      // scatter workgroup: 16 waves x 16 rows of 64 keys = one radix tile
#define RSC_ITEMS (RSORT_TILE / (64 * RSC_WAVES))
#define RSC_WG (64 * RSC_WAVES)
// lanes of the wave whose NBITS-bit digit equals this lane's (0 for a lane that is not valid).
// Per digit bit: x = the lane's bit spread over a word (0 / ~0), bal = the lanes whose bit is set; the lanes that agree with this lane
// in the bit are ~(bal ^ x).  Kept in 32-bit halves: the compiler then spends six VALU instructions per bit (shift, v_bfe_i32, v_cmp,
// two v_xor and half a v_or3 over the differences) instead of the nine of `m &= bit ? bal : ~bal` on 64-bit masks; the ranking phase of
// the LDS sorts is VALU bound (k_tr_sort 39.8 -> 36.8 ms, k_bucket_sort 16.9 -> 15.3 ms per 342 mixed blocks).  A hand-scheduled
// four-instruction form (v_bfe_i32, v_cmp into an SGPR pair, two v_bitop3_b32 reading it) is correct but SLOWER (50.7 ms): an
// instruction with an SGPR operand occupies the SIMD twice as long as a VGPR-only one (tools/ubench_valu_rate.hip) and a VALU read of an
// SGPR that a VALU compare just wrote stalls the wave for about 16 cycles.
template <int NBITS>
__device__ __forceinline__ uint64_t bw_match(u32 d, bool valid) {
  const uint64_t m0 = kz_ballot(valid);
  u32 mlo = (u32)m0, mhi = (u32)(m0 >> 32);
#pragma unroll
  for (int b = 0; b < NBITS; b++) {
    const u32 x = (u32)(((int32_t)(d << (31 - b))) >> 31);
    const uint64_t bal = kz_ballot(x != 0u);
    mlo &= ~((u32)bal ^ x);
    mhi &= ~((u32)(bal >> 32) ^ x);
  }
  return valid ? (((uint64_t)mhi << 32) | mlo) : 0ULL;
}
template <int BINS, int NBITS, bool MSD, typename CNT, bool TEXT>
__device__ __forceinline__ void radix_scatter_body(const u64* __restrict__ keyIn, const u32* __restrict__ valIn,
                                                   u64* __restrict__ keyOut, u32* __restrict__ valOut,
                                                   const BwtArrays& A, int shift, TextSrc X) {
  const int b = bw_row_block(A, blockIdx.y);
  const int w0 = MSD ? 0 : A.d_w[b];
  const int m = A.d_m[b] - w0;
  const int tile = blockIdx.x;
  if ((int64_t)tile * RSORT_TILE >= m) return;
  __shared__ CNT cnt[RSC_WAVES][BINS];
  __shared__ u32 gdelta[BINS];       // global slot of the digit's first element of this tile - its tile-local slot
  __shared__ u32 scan[32];
  __shared__ u64 stage[RSORT_TILE];  // 64 KiB: keys, then values
  for (int i = threadIdx.x; i < RSC_WAVES * BINS; i += RSC_WG) (&cnt[0][0])[i] = 0;
  __syncthreads();
  const int64_t off = (int64_t)b * A.NS + w0;
  const int wave = threadIdx.x >> 6;
  const int lane = kz_lane();
  const int tbase = tile * RSORT_TILE;
  const int base = tbase + wave * (64 * RSC_ITEMS);
  const int tcount = min(RSORT_TILE, m - tbase);
  const uint64_t lt = kz_lanemask_lt();
  u64 k[RSC_ITEMS]; u32 v[RSC_ITEMS]; u32 dr[RSC_ITEMS];   // dr = digit | (rank<<8), later the tile-local slot
#pragma unroll
  for (int r = 0; r < RSC_ITEMS; r++) {
    const int idx = base + r * 64 + lane;
    const bool valid = idx < m;
    if (TEXT) {
      k[r] = valid ? bw_text_key(X.src + (int64_t)b * X.stride, idx, m, X.K) : 0;
      v[r] = (u32)idx;
    } else {
      k[r] = valid ? keyIn[off + idx] : 0;
      v[r] = valid ? valIn[off + idx] : 0;
    }
  }
#pragma unroll
  for (int r = 0; r < RSC_ITEMS; r++) {
    const int idx = base + r * 64 + lane;
    const bool valid = idx < m;
    const u32 d = (u32)((k[r] >> shift) & (BINS - 1));
    // rows holding a single digit value (high key bytes, runs) skip the 8-ballot match-any
    const uint64_t vm = kz_ballot(valid);
    const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)d);
    const bool uni = (vm != 0) && (kz_ballot(valid && d == d0) == vm);
    const uint64_t peers = uni ? (valid ? vm : 0ULL) : bw_match<NBITS>(d, valid);
    u32 pre = 0;
    if (valid) pre = cnt[wave][d];
    const u32 rnk = pre + (u32)__popcll(peers & lt);
    // the highest peer lane publishes the new count (all peers read `pre` before: same wave, in order)
    if (valid && (peers >> lane) == 1ULL) cnt[wave][d] = (CNT)(pre + (u32)__popcll(peers));
    dr[r] = d | (rnk << 16);
  }
  __syncthreads();
  {
    u32 tot = 0;
    u32 c[RSC_WAVES];
    const int d = threadIdx.x & (BINS - 1);
#pragma unroll
    for (int w = 0; w < RSC_WAVES; w++) { c[w] = cnt[w][d]; tot += c[w]; }
    u32 total;
    // exclusive scan over the digits (the threads behind them carry zero and are ignored)
    const u32 ts = kz_wg_excl_sum(threadIdx.x < BINS ? tot : 0u, scan, &total);
    if (threadIdx.x < BINS) {
      gdelta[d] = A.digitBase[b * MSD_BINS + d] + A.tileHist[(int64_t)b * A.HS + (int64_t)tile * BINS + d] - ts;
      u32 run = ts;
#pragma unroll
      for (int w = 0; w < RSC_WAVES; w++) { cnt[w][d] = (CNT)run; run += c[w]; }
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RSC_ITEMS; r++) {
    const int idx = base + r * 64 + lane;
    if (idx < m) { const u32 slot = cnt[wave][dr[r] & 0xFFFF] + (dr[r] >> 16); dr[r] = slot; stage[slot] = k[r]; }
  }
  __syncthreads();
  u32 gp[RSC_ITEMS];
#pragma unroll
  for (int r = 0; r < RSC_ITEMS; r++) {
    const int slot = r * RSC_WG + threadIdx.x;
    if (slot < tcount) {
      const u64 kk = stage[slot];
      gp[r] = gdelta[(u32)((kk >> shift) & (BINS - 1))] + (u32)slot;
      keyOut[off + gp[r]] = kk;
    }
  }
  __syncthreads();
  u32* stageV = (u32*)stage;
#pragma unroll
  for (int r = 0; r < RSC_ITEMS; r++) {
    const int idx = base + r * 64 + lane;
    if (idx < m) stageV[dr[r]] = v[r];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RSC_ITEMS; r++) {
    const int slot = r * RSC_WG + threadIdx.x;
    if (slot < tcount) valOut[off + gp[r]] = stageV[slot];
  }
}
__global__ __launch_bounds__(RSC_WG) void k_radix_scatter(const u64* __restrict__ keyIn, const u32* __restrict__ valIn,
                                                           u64* __restrict__ keyOut, u32* __restrict__ valOut,
                                                           BwtArrays A, int shift) { radix_scatter_body<256, 8, false, u32, false>(keyIn, valIn, keyOut, valOut, A, shift, TextSrc{nullptr, 0, 0}); }
__global__ __launch_bounds__(RSC_WG) void k_radix_scatter0(u64* __restrict__ keyOut, u32* __restrict__ valOut, BwtArrays A, TextSrc X) {
  radix_scatter_body<256, 8, false, u32, true>(nullptr, nullptr, keyOut, valOut, A, 0, X);
}
// The bucket partition need not be stable (k_bucket_sort orders by the whole key, and suffixes with equal keys stay
// one group whatever their order), so the tile-local rank of an element is simply an LDS atomic on its bucket's counter.
__global__ __launch_bounds__(RSC_WG) void k_msd_scatter(const u64* __restrict__ keyIn, const u32* __restrict__ valIn,
                                                         u64* __restrict__ keyOut, u32* __restrict__ valOut,
                                                         BwtArrays A, int shift) {
  const int b = bw_row_block(A, blockIdx.y);
  const int m = A.d_m[b];
  const int tile = blockIdx.x;
  if ((int64_t)tile * RSORT_TILE >= m) return;
  __shared__ u32 cnt[MSD_BINS];
  __shared__ u32 gdelta[MSD_BINS];
  __shared__ u32 scan[32];
  __shared__ u64 stage[RSORT_TILE];  // 128 KiB: keys, then values
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t off = (int64_t)b * A.NS;
  const int lane = kz_lane();
  const int tbase = tile * RSORT_TILE;
  const int tcount = min(RSORT_TILE, m - tbase);
  const uint64_t lt = kz_lanemask_lt();
  u64 k[RSC_ITEMS]; u32 v[RSC_ITEMS]; u32 dr[RSC_ITEMS];
#pragma unroll
  for (int r = 0; r < RSC_ITEMS; r++) {
    const int idx = tbase + r * RSC_WG + threadIdx.x;
    const bool valid = idx < m;
    k[r] = valid ? keyIn[off + idx] : 0;
    v[r] = valid ? valIn[off + idx] : 0;
  }
#pragma unroll
  for (int r = 0; r < RSC_ITEMS; r++) {
    const int idx = tbase + r * RSC_WG + threadIdx.x;
    const bool valid = idx < m;
    const u32 d = (u32)(k[r] >> shift) & (MSD_BINS - 1);
    const uint64_t vm = kz_ballot(valid);
    const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)d);
    u32 pos = 0;
    if (vm != 0 && kz_ballot(valid && d == d0) == vm) {             // one bucket for the whole row: one atomic
      u32 basePos = 0;
      if (lane == (int)__builtin_ctzll(vm)) basePos = atomicAdd(&cnt[d0], (u32)__popcll(vm));
      basePos = (u32)__shfl((int)basePos, (int)__builtin_ctzll(vm), 64);
      pos = basePos + (u32)__popcll(vm & lt);
    } else if (valid) pos = atomicAdd(&cnt[d], 1u);
    dr[r] = pos;
  }
  __syncthreads();
  {
    const u32 tot = cnt[threadIdx.x];
    u32 total;
    const u32 ts = kz_wg_excl_sum(tot, scan, &total);
    gdelta[threadIdx.x] = A.digitBase[b * MSD_BINS + threadIdx.x] + A.tileHist[(int64_t)b * A.HS + (int64_t)tile * MSD_BINS + threadIdx.x] - ts;
    cnt[threadIdx.x] = ts;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RSC_ITEMS; r++) {
    const int idx = tbase + r * RSC_WG + threadIdx.x;
    if (idx < m) { const u32 slot = cnt[(u32)(k[r] >> shift) & (MSD_BINS - 1)] + dr[r]; dr[r] = slot; stage[slot] = k[r]; }
  }
  __syncthreads();
  u32 gp[RSC_ITEMS];
#pragma unroll
  for (int r = 0; r < RSC_ITEMS; r++) {
    const int slot = r * RSC_WG + threadIdx.x;
    if (slot < tcount) {
      const u64 kk = stage[slot];
      gp[r] = gdelta[(u32)(kk >> shift) & (MSD_BINS - 1)] + (u32)slot;
      keyOut[off + gp[r]] = kk;
    }
  }
  __syncthreads();
  u32* stageV = (u32*)stage;
#pragma unroll
  for (int r = 0; r < RSC_ITEMS; r++) {
    const int idx = tbase + r * RSC_WG + threadIdx.x;
    if (idx < m) stageV[dr[r]] = v[r];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RSC_ITEMS; r++) {
    const int slot = r * RSC_WG + threadIdx.x;
    if (slot < tcount) valOut[off + gp[r]] = stageV[slot];
  }
}

// ---------------------------------------------------------------------------------------------
// step 3 (SA order, after the sort).  g(c) = old group head slot = key >> gshift (gshift >= 64: one group, slot 0).
__device__ __forceinline__ u32 bw_group(u64 k, int gshift) { return gshift >= 64 ? 0u : (u32)(k >> gshift); }

// Wave w of a tile owns the 1024 consecutive sorted elements [w*1024, (w+1)*1024) as 16 rows of 64 (coalesced).
// Loads the keys, returns per row the ballots "key differs from its predecessor" (hb) and "old group differs" (sb).
__device__ __forceinline__ void bw_row_flags(const u64* __restrict__ key, int base, int m, int gshift, int lane,
                                             u64 (&k)[RS_ITEMS], uint64_t (&hb)[RS_ITEMS], uint64_t (&sb)[RS_ITEMS]) {
  u64 prevRowLast = (base > 0 && base <= m) ? key[base - 1] : 0;      // uniform
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const int c = base + r * 64 + lane;
    k[r] = (c < m) ? key[c] : 0;
    u64 prev = __shfl_up(k[r], 1, 64);
    if (lane == 0) prev = prevRowLast;
    const bool valid = c < m;
    const bool head = valid && (c == 0 || k[r] != prev);
    const bool seg = head && (c == 0 || bw_group(k[r], gshift) != bw_group(prev, gshift));
    hb[r] = kz_ballot(head);
    sb[r] = kz_ballot(seg);
    prevRowLast = __shfl(k[r], 63, 64);
  }
}

// per tile: (last index+1 where the old group changes, last index+1 where the key changes)
__global__ __launch_bounds__(KZ_WG) void k_seg_reduce(const u64* __restrict__ keyS, BwtArrays A, int gshift) {
  const int b = bw_row_block(A, blockIdx.y);
  const int w0 = A.d_w[b];
  const int m = A.d_m[b] - w0;
  const int tile = blockIdx.x;
  if ((int64_t)tile * RS_TILE >= m) return;
  __shared__ u32 wS[4], wH[4];
  const u64* key = keyS + (int64_t)b * A.NS + w0;
  const int wave = threadIdx.x >> 6, lane = kz_lane();
  const int base = tile * RS_TILE + wave * (64 * RS_ITEMS);
  u64 k[RS_ITEMS]; uint64_t hb[RS_ITEMS], sb[RS_ITEMS];
  bw_row_flags(key, base, m, gshift, lane, k, hb, sb);
  u32 ms = 0, mh = 0;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    if (hb[r]) mh = (u32)(base + r * 64 + 63 - (int)__builtin_clzll(hb[r])) + 1;
    if (sb[r]) ms = (u32)(base + r * 64 + 63 - (int)__builtin_clzll(sb[r])) + 1;
  }
  if (lane == 0) { wS[wave] = ms; wH[wave] = mh; }
  __syncthreads();
  if (threadIdx.x == 0) {
    A.tileA[(int64_t)b * A.T + tile] = max(max(wS[0], wS[1]), max(wS[2], wS[3]));
    A.tileB[(int64_t)b * A.T + tile] = max(max(wH[0], wH[1]), max(wH[2], wH[3]));
  }
}
// per block: exclusive max-scans over the tiles -- one wave per block
__global__ void k_seg_scan(BwtArrays A) {
  const int b = bw_row_block(A, blockIdx.x);
  const int m = A.d_m[b] - A.d_w[b];
  const int tiles = (m + RS_TILE - 1) / RS_TILE;
  u32* ta = A.tileA + (int64_t)b * A.T;
  u32* tb = A.tileB + (int64_t)b * A.T;
  u32 ca = 0, cb = 0;
  for (int base = 0; base < tiles; base += 64) {
    const int i = base + threadIdx.x;
    const u32 va = (i < tiles) ? ta[i] : 0, vb = (i < tiles) ? tb[i] : 0;
    const u32 ia = kz_wave_incl_max(va), ib = kz_wave_incl_max(vb);
    u32 ea = __shfl_up(ia, 1, 64), eb = __shfl_up(ib, 1, 64);
    if (threadIdx.x == 0) { ea = 0; eb = 0; }
    ea = ea > ca ? ea : ca; eb = eb > cb ? eb : cb;
    if (i < tiles) { ta[i] = ea; tb[i] = eb; }
    const u32 la = __shfl(ia, 63, 64), lb = __shfl(ib, 63, 64);
    ca = ca > la ? ca : la; cb = cb > lb ? cb : lb;
  }
}
// apply: slot = g + (c - segment start); new rank = g + (new-group start - segment start); LIVE unless the new
// group is a singleton, in which case the suffix is final
__global__ __launch_bounds__(KZ_WG) void k_seg_apply(const u64* __restrict__ keyS, const u32* __restrict__ valS, BwtArrays A, int gshift) {
  const int b = bw_row_block(A, blockIdx.y);
  const int w0 = A.d_w[b];
  const int m = A.d_m[b] - w0;
  const int tile = blockIdx.x;
  if ((int64_t)tile * RS_TILE >= m) return;
  __shared__ u32 wS[4], wH[4];
  const int64_t off = (int64_t)b * A.NS;
  const u64* key = keyS + off + w0;
  const int wave = threadIdx.x >> 6, lane = kz_lane();
  const int base = tile * RS_TILE + wave * (64 * RS_ITEMS);
  u64 k[RS_ITEMS]; uint64_t hb[RS_ITEMS], sb[RS_ITEMS];
  bw_row_flags(key, base, m, gshift, lane, k, hb, sb);
  u32 ms = 0, mh = 0;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    if (hb[r]) mh = (u32)(base + r * 64 + 63 - (int)__builtin_clzll(hb[r])) + 1;
    if (sb[r]) ms = (u32)(base + r * 64 + 63 - (int)__builtin_clzll(sb[r])) + 1;
  }
  if (lane == 0) { wS[wave] = ms; wH[wave] = mh; }
  __syncthreads();
  // carry into this wave: the tile's carry and the previous waves of the tile ("index + 1", 0 = none)
  u32 carS = A.tileA[(int64_t)b * A.T + tile], carH = A.tileB[(int64_t)b * A.T + tile];
  for (int w = 0; w < wave; w++) { carS = max(carS, wS[w]); carH = max(carH, wH[w]); }
  u32* rank = A.rank + off;
  u32* sa = A.sa + off;
  const u32* val = valS + off + w0;
  const uint64_t le = kz_lanemask_lt() | (1ULL << lane);
  const u64 afterLast = (base + 64 * RS_ITEMS < m) ? key[base + 64 * RS_ITEMS] : 0;   // uniform
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const int rowBase = base + r * 64;
    const int c = rowBase + lane;
    const uint64_t hbl = hb[r] & le, sbl = sb[r] & le;
    const u32 hh = hbl ? (u32)(rowBase + 63 - (int)__builtin_clzll(hbl)) : carH - 1;   // index of the new group's first element
    const u32 ss = sbl ? (u32)(rowBase + 63 - (int)__builtin_clzll(sbl)) : carS - 1;   // index of the old group's first element
    // key of the next element: next lane / first lane of the next row / first key behind the wave's chunk
    u64 nextk = __shfl_down(k[r], 1, 64);
    const u64 nextRowFirst = (r + 1 < RS_ITEMS) ? __shfl(k[(r + 1 < RS_ITEMS) ? r + 1 : r], 0, 64) : afterLast;
    if (lane == 63) nextk = nextRowFirst;
    if (c < m) {
      const u32 g = bw_group(k[r], gshift);
      const bool headC = hh == (u32)c;
      const bool headN = (c + 1 >= m) || (nextk != k[r]);
      const u32 sv = val[c];
      const bool live = !(headC && headN);
      // the first new group of an old group keeps the old head slot: while it stays LIVE its members' ranks (g | LIVE since the
      // round before) do not change, and the random 4-byte store is saved (not in the first round: ranks are unwritten then)
      if (!(live && hh == ss && gshift < 64)) rank[sv] = (g + (hh - ss)) | (live ? BW_LIVE : 0u);
      if (!live) sa[g + ((u32)c - ss)] = sv;
    }
    if (hb[r]) carH = (u32)(rowBase + 63 - (int)__builtin_clzll(hb[r])) + 1;
    if (sb[r]) carS = (u32)(rowBase + 63 - (int)__builtin_clzll(sb[r])) + 1;
  }
}

// ---------------------------------------------------------------------------------------------
// One bucket of a later round whose longest old group is above BK_GMAX (flagged by k_bucket_count*), start to finish in one
// workgroup: load the bucket's pairs, LSD radix sort them in LDS by (old group, secondary key), then do what k_seg_reduce /
// k_seg_apply do for the sorted run.  An element is one u64: (group - bucket base : BK_BITS | r2 : bitsR | suffix : bitsG),
// 58 bits for 4 MiB blocks.  Wave w owns the contiguous rows [w*R, (w+1)*R) of 64 elements; between passes the elements
// live in registers, the LDS buffer is only the exchange (48 KiB + 16 KiB of u16 counters, 64 VGPRs: two workgroups per CU).
#define BK_DBITS 9             // LDS sort digit: 36 key bits of a 4 MiB block in four passes
#define BK_DBINS (1 << BK_DBITS)
#define BK_SMALL 1024
#define BK_SORT 0x40000000u
static_assert(RSC_WG == MSD_BINS, "k_msd_scatter geometry");

template <int WAVES, int ROWS>
__device__ __forceinline__ void bucket_body(const u64* __restrict__ keyS, const u32* __restrict__ valS, const BwtArrays& A, int bitsR, int bitsG) {
  const int b = bw_row_block(A, blockIdx.y);
  const u32 d = blockIdx.x;
  const u32 bc0 = A.bucketCnt[b * MSD_BINS + d];
  if ((bc0 & (BK_SORT | BK_BIG)) != BK_SORT) return;          // only the buckets k_bucket_count left behind
  const u32 bc = bc0 & ~BK_SORT;
  const int cnt = (int)bc;
  const int wave = threadIdx.x >> 6, lane = kz_lane();
  __shared__ u64 buf[WAVES * ROWS * 64];
  __shared__ uint16_t cw[WAVES][BK_DBINS];                     // per-wave digit counts, then bucket-local slots
  __shared__ u32 wsum[BK_DBINS / 64];
  __shared__ u32 wS[WAVES], wH[WAVES];
  __shared__ int skip[8];
  const int64_t off = (int64_t)b * A.NS;
  const u32 bo = A.digitBase[b * MSD_BINS + d];
  const int rows = (cnt + 63) >> 6;
  const int R = (rows + WAVES - 1) / WAVES;                    // rows per wave, 1..ROWS (uniform)
  const int base = wave * R * 64;
  const uint64_t lt = kz_lanemask_lt();
  const u64 kbase = (u64)d << (bitsR + BK_BITS);
  u64 k[ROWS]; u32 dr[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    const int idx = base + r * 64 + lane;
    k[r] = ~0ULL;
    if (r < R && idx < cnt) k[r] = ((keyS[off + bo + idx] - kbase) << bitsG) | (u64)valS[off + bo + idx];
  }
  {
    if (threadIdx.x < 8) skip[threadIdx.x] = 0;
    const int keyBits = bitsR + BK_BITS;
    const int passes = (keyBits + BK_DBITS - 1) / BK_DBITS;
    for (int p = 0; p < passes; p++) {
      const int shift = bitsG + BK_DBITS * p;
      for (int i = threadIdx.x; i < WAVES * BK_DBINS / 2; i += WAVES * 64) ((u32*)&cw[0][0])[i] = 0;
      __syncthreads();
#pragma unroll
      for (int r = 0; r < ROWS; r++) {
        if (r < R) {
          const int idx = base + r * 64 + lane;
          const bool valid = idx < cnt;
          const u32 dg = (u32)(k[r] >> shift) & (BK_DBINS - 1);
          const uint64_t vm = kz_ballot(valid);
          if (vm == 0) continue;                               // rows behind the bucket's last element (uniform)
          const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)dg);
          const bool uni = kz_ballot(valid && dg == d0) == vm;
          const uint64_t peers = uni ? (valid ? vm : 0ULL) : bw_match<BK_DBITS>(dg, valid);
          u32 pre = 0;
          if (valid) pre = cw[wave][dg];
          const u32 rnk = pre + (u32)__popcll(peers & lt);
          if (valid && (peers >> lane) == 1ULL) cw[wave][dg] = (uint16_t)(pre + (u32)__popcll(peers));
          dr[r] = dg | (rnk << 16);
        }
      }
      __syncthreads();
      // digit totals and their exclusive scan: thread t owns the digits t, t + WAVES*64, ...
      constexpr int DPT = (BK_DBINS + WAVES * 64 - 1) / (WAVES * 64);
      u32 c[DPT][WAVES]; u32 tot[DPT]; u32 mine = 0;
#pragma unroll
      for (int q = 0; q < DPT; q++) {
        const int dg = threadIdx.x * DPT + q;                  // consecutive digits per thread: a plain scan order
        tot[q] = 0;
        if (dg < BK_DBINS) {
#pragma unroll
          for (int w = 0; w < WAVES; w++) { c[q][w] = cw[w][dg]; tot[q] += c[q][w]; }
          if (tot[q] == (u32)cnt) skip[p] = 1;                 // one digit value for the whole bucket: nothing moves
        }
        mine += tot[q];
      }
      const u32 inc = kz_wave_incl_sum(mine);
      if (lane == 63 && wave < BK_DBINS / 64) wsum[wave] = inc;
      __syncthreads();
      if (skip[p]) continue;                                   // uniform
      {
        u32 run = inc - mine;
        for (int w = 0; w < wave && w < BK_DBINS / 64; w++) run += wsum[w];
#pragma unroll
        for (int q = 0; q < DPT; q++) {
          const int dg = threadIdx.x * DPT + q;
          if (dg < BK_DBINS) {
#pragma unroll
            for (int w = 0; w < WAVES; w++) { cw[w][dg] = (uint16_t)run; run += c[q][w]; }
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < ROWS; r++) {
        if (r < R) {
          const int idx = base + r * 64 + lane;
          if (idx < cnt) buf[cw[wave][dr[r] & 0xFFFF] + (dr[r] >> 16)] = k[r];
        }
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < ROWS; r++) {
        if (r < R) {
          const int idx = base + r * 64 + lane;
          if (idx < cnt) k[r] = buf[idx];
        }
      }
      __syncthreads();
    }
    // sorted: neighbours through the LDS buffer
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
      if (r < R) {
        const int idx = base + r * 64 + lane;
        if (idx < cnt) buf[idx] = k[r];
      }
    }
    __syncthreads();
  }
  uint64_t hb[ROWS], sb[ROWS];
  u32 ms = 0, mh = 0;
  const int gsh = bitsG + bitsR;
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    hb[r] = 0; sb[r] = 0;
    if (r < R) {
      const int idx = base + r * 64 + lane;
      const bool valid = idx < cnt;
      const u64 prev = (valid && idx > 0) ? buf[idx - 1] : 0;
      const bool head = valid && (idx == 0 || (k[r] >> bitsG) != (prev >> bitsG));
      const bool seg = head && (idx == 0 || (k[r] >> gsh) != (prev >> gsh));
      hb[r] = kz_ballot(head);
      sb[r] = kz_ballot(seg);
      if (hb[r]) mh = (u32)(base + r * 64 + 63 - (int)__builtin_clzll(hb[r])) + 1;
      if (sb[r]) ms = (u32)(base + r * 64 + 63 - (int)__builtin_clzll(sb[r])) + 1;
    }
  }
  u32 carS = 0, carH = 0;
  {
    if (lane == 0) { wS[wave] = ms; wH[wave] = mh; }
    __syncthreads();
    for (int w = 0; w < wave; w++) { carS = max(carS, wS[w]); carH = max(carH, wH[w]); }
  }
  u32* rank = A.rank + off;
  u32* sa = A.sa + off;
  const uint64_t le = lt | (1ULL << lane);
  const u64 vmask = (1ULL << bitsG) - 1ULL;
  const u32 gbase = d << BK_BITS;
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    if (r < R) {
      const int rowBase = base + r * 64;
      const int idx = rowBase + lane;
      const uint64_t hbl = hb[r] & le, sbl = sb[r] & le;
      const u32 hh = hbl ? (u32)(rowBase + 63 - (int)__builtin_clzll(hbl)) : carH - 1;
      const u32 ss = sbl ? (u32)(rowBase + 63 - (int)__builtin_clzll(sbl)) : carS - 1;
      if (idx < cnt) {
        const u64 nextk = (idx + 1 < cnt) ? buf[idx + 1] : 0;
        const bool headC = hh == (u32)idx;
        const bool headN = (idx + 1 >= cnt) || ((nextk >> bitsG) != (k[r] >> bitsG));
        const bool live = !(headC && headN);
        const u32 g = gbase + (u32)(k[r] >> gsh);
        const u32 sv = (u32)(k[r] & vmask);
        if (!(live && hh == ss)) rank[sv] = (g + (hh - ss)) | (live ? BW_LIVE : 0u);     // see k_seg_apply
        if (!live) sa[g + ((u32)idx - ss)] = sv;
      }
      if (hb[r]) carH = (u32)(rowBase + 63 - (int)__builtin_clzll(hb[r])) + 1;
      if (sb[r]) carS = (u32)(rowBase + 63 - (int)__builtin_clzll(sb[r])) + 1;
    }
  }
}
__global__ __launch_bounds__(1024, 8) void k_bucket_sort(const u64* __restrict__ keyS, const u32* __restrict__ valS, BwtArrays A, int bitsR, int bitsG) {
  bucket_body<16, BK_CAP / 1024>(keyS, valS, A, bitsR, bitsG);
}

// ---------------------------------------------------------------------------------------------
// The same job without sorting.  What the apply step needs of a suffix is only, among the members of its old group,
// how many have a smaller secondary key (rk) and how many the same one (eq): the new group starts rk slots behind the
// old head, it is a singleton iff eq == 1, and then the suffix's final slot is head + rk.  So the bucket is only
// PARTITIONED by old group (LDS atomics on one counter per slot of the bucket: order inside a group is irrelevant), and
// every element then walks the members of its group.  Groups are small where this is used (a handful of suffixes
// sharing 7, 14, ... bytes); a bucket holding a group above BK_GMAX is left to k_bucket_sort (flag BK_SORT).
#define BK_GMAX 256
#define BK_SLOTS (1 << BK_BITS)
template <int WAVES, int ROWS, int LO, int HI, typename CNT>
__device__ __forceinline__ void bucket_count_body(const u64* __restrict__ keyS, const u32* __restrict__ valS, const BwtArrays& A, int bitsR, int bitsG, u32 gmax) {
  const int b = bw_row_block(A, blockIdx.y);
  const u32 d = blockIdx.x;
  const u32 bc = A.bucketCnt[b * MSD_BINS + d];
  if (bc <= (u32)LO || bc > (u32)HI) return;
  const int cnt = (int)bc;
  const int wave = threadIdx.x >> 6, lane = kz_lane();
  const bool tiny = LO == 0 && cnt <= 64;
  if (tiny && wave != 0) return;
  const int64_t off = (int64_t)b * A.NS;
  const u32 bo = A.digitBase[b * MSD_BINS + d];
  u32* rank = A.rank + off;
  u32* sa = A.sa + off;
  const u32 gbase = d << BK_BITS;
  const u32 rmask = (1u << bitsR) - 1u;
  constexpr int WG = WAVES * 64;
  if (tiny) {
    const bool valid = lane < cnt;
    const u64 key = valid ? keyS[off + bo + lane] : ~0ULL;
    const u32 sv = valid ? valS[off + bo + lane] : 0u;
    const u32 gv = (u32)(key >> bitsR), rv = (u32)key & rmask;
    u32 rk = 0, eq = 0;
    for (int j = 0; j < cnt; j++) {
      const u32 gj = (u32)__builtin_amdgcn_readlane((int)gv, j), rj = (u32)__builtin_amdgcn_readlane((int)rv, j);
      if (gj == gv) { rk += (rj < rv) ? 1u : 0u; eq += (rj == rv) ? 1u : 0u; }
    }
    if (valid) {
      const bool live = eq > 1;
      if (!(live && rk == 0)) rank[sv] = (gv + rk) | (live ? BW_LIVE : 0u);               // see k_seg_apply
      if (!live) sa[gv + rk] = sv;
    }
    return;
  }
  __shared__ u64 buf[WAVES * ROWS * 64];
  __shared__ CNT start[BK_SLOTS];
  __shared__ u32 wsum[WAVES];
  __shared__ u32 gmaxS;
  for (int i = threadIdx.x; i < BK_SLOTS; i += WG) start[i] = 0;
  if (threadIdx.x == 0) gmaxS = 0;
  __syncthreads();
  const u64 kbase = (u64)d << (bitsR + BK_BITS);
  u64 k[ROWS]; u32 pos[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    const int idx = r * WG + threadIdx.x;
    k[r] = 0; pos[r] = 0;
    if (idx < cnt) {
      k[r] = ((keyS[off + bo + idx] - kbase) << bitsG) | (u64)valS[off + bo + idx];
      pos[r] = (u32)atomicAdd(&start[(u32)(k[r] >> (bitsG + bitsR))], (CNT)1);
    }
  }
  __syncthreads();
  {                                                            // exclusive scan over the slots: BK_SLOTS / WG consecutive per thread
    constexpr int SPT = BK_SLOTS / WG;
    u32 c[SPT]; u32 mine = 0, gm = 0;
#pragma unroll
    for (int q = 0; q < SPT; q++) { c[q] = start[threadIdx.x * SPT + q]; mine += c[q]; gm = max(gm, c[q]); }
    const u32 inc = kz_wave_incl_sum(mine);
    if (lane == 63) wsum[wave] = inc;
    if (gm > gmax) atomicMax(&gmaxS, gm);
    __syncthreads();
    u32 run = inc - mine;
    for (int w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
    for (int q = 0; q < SPT; q++) { start[threadIdx.x * SPT + q] = (CNT)run; run += c[q]; }
  }
  __syncthreads();
  if (gmaxS > gmax) {                                        // a long group: the sorting kernel takes this bucket
    if (threadIdx.x == 0) A.bucketCnt[b * MSD_BINS + d] = bc | BK_SORT;
    return;
  }
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    const int idx = r * WG + threadIdx.x;
    if (idx < cnt) buf[(u32)start[(u32)(k[r] >> (bitsG + bitsR))] + pos[r]] = k[r];
  }
  __syncthreads();
  const u64 vmask = (1ULL << bitsG) - 1ULL;
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    const int idx = r * WG + threadIdx.x;                      // position in group order: a wave reads a few neighbouring groups
    if (idx < cnt) {
      const u64 e = buf[idx];
      const u32 grel = (u32)(e >> (bitsG + bitsR));
      const u32 rv = (u32)(e >> bitsG) & rmask;
      const int gs = (int)start[grel];
      const int ge = (grel + 1 < (u32)BK_SLOTS) ? (int)start[grel + 1] : cnt;
      u32 rk = 0, eq = 0;
      int j = gs;
      for (; j + 4 <= ge; j += 4) {                            // four independent LDS reads in flight
        const u64 e0 = buf[j], e1 = buf[j + 1], e2 = buf[j + 2], e3 = buf[j + 3];
        const u32 r0 = (u32)(e0 >> bitsG) & rmask, r1 = (u32)(e1 >> bitsG) & rmask, r2_ = (u32)(e2 >> bitsG) & rmask, r3 = (u32)(e3 >> bitsG) & rmask;
        rk += (r0 < rv) + (r1 < rv) + (r2_ < rv) + (r3 < rv);
        eq += (r0 == rv) + (r1 == rv) + (r2_ == rv) + (r3 == rv);
      }
      for (; j < ge; j++) {
        const u32 rj = (u32)(buf[j] >> bitsG) & rmask;
        rk += (rj < rv) ? 1u : 0u; eq += (rj == rv) ? 1u : 0u;
      }
      const u32 sv = (u32)(e & vmask);
      const u32 g = gbase + grel;
      const bool live = eq > 1;
      if (!(live && rk == 0)) rank[sv] = (g + rk) | (live ? BW_LIVE : 0u);
      if (!live) sa[g + rk] = sv;
    }
  }
}
__global__ __launch_bounds__(256) void k_bucket_count_s(const u64* __restrict__ keyS, const u32* __restrict__ valS, BwtArrays A, int bitsR, int bitsG, u32 gmax) {
  bucket_count_body<4, BK_SMALL / 256, 0, BK_SMALL, u32>(keyS, valS, A, bitsR, bitsG, gmax);
}
__global__ __launch_bounds__(1024) void k_bucket_count(const u64* __restrict__ keyS, const u32* __restrict__ valS, BwtArrays A, int bitsR, int bitsG, u32 gmax) {
  bucket_count_body<16, BK_CAP / 1024, BK_SMALL, BK_CAP, u32>(keyS, valS, A, bitsR, bitsG, gmax);
}

// (the trie rounds' per-block counters, defined further down; k_live_emit reads two of them)
#define TR_META 32               // ints per block: [0] nodes, [1] buckets, [2 + L] first node of depth L, [10 + L] end, [18] error, [20] lazy ranks: live suffixes found
// ---------------------------------------------------------------------------------------------
// step 1 (text order): compact the LIVE suffixes and build their keys from sequential rank reads
// The compact list need not be in text order (its only readers partition it by group: k_msd_* / the key trie round), so a tile
// reserves its slice with ONE returning atomic on the block's counter instead of a count kernel + a scan kernel + a second read of
// every rank (k_live_count / k_live_scan until round 4: 36 ms and 122 GB of rank reads per 8 GiB).  Inside a tile the order stays
// the text order.  tileLive[tile] = live suffixes of the tile after the previous round (0 stays 0: suffixes only become final).
__global__ __launch_bounds__(KZ_WG) void k_live_emit(u64* __restrict__ keyN, u32* __restrict__ valN, BwtArrays A, int h, int bitsR, int first, const int32_t* __restrict__ lazyMeta) {
  const int b = bw_row_block(A, blockIdx.y);
  const int n = A.d_n[b];
  const int tile = blockIdx.x;
  if ((int64_t)tile * RS_TILE >= n) return;
  if (lazyMeta && lazyMeta[(int64_t)b * TR_META] == 256 && lazyMeta[(int64_t)b * TR_META + 18] == 0 && lazyMeta[(int64_t)b * TR_META + 20] == 0) {
    // round 0 left no live suffix in this block and stored no ranks (k_tr_sort, lazy ranks): nothing to read, nothing to emit
    if (threadIdx.x == 0) A.tileLive[(int64_t)b * A.T + tile] = 0;
    return;
  }
  if (!first && A.tileLive[(int64_t)b * A.T + tile] == 0) return;     // nothing left here: neither read nor written
  __shared__ u32 rowCnt[RS_ITEMS * 4 + 1];   // live suffixes per (wave, row), then exclusive prefix
  __shared__ u32 tileBase;
  const int64_t off = (int64_t)b * A.NS;
  const u32* rank = A.rank + off;
  const int wave = threadIdx.x >> 6, lane = kz_lane();
  // wave w owns the 1024 consecutive suffixes [w*1024, (w+1)*1024): rows of 64 -> every access is coalesced
  const int base = tile * RS_TILE + wave * (64 * RS_ITEMS);
  u32 rk[RS_ITEMS];
  uint64_t bal[RS_ITEMS];
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const int s = base + r * 64 + lane;
    rk[r] = (s < n) ? rank[s] : 0u;
    bal[r] = kz_ballot((rk[r] & BW_LIVE) != 0);
    if (lane == 0) rowCnt[wave * RS_ITEMS + r] = (u32)__popcll(bal[r]);
  }
  __syncthreads();
  if (threadIdx.x < 64) {                      // exclusive scan of the 64 (wave,row) counts by one wave
    const u32 v = rowCnt[threadIdx.x];
    const u32 inc = kz_wave_incl_sum(v);
    rowCnt[threadIdx.x] = inc - v;
    if (threadIdx.x == 63) {
      const u32 tot = inc;
      A.tileLive[(int64_t)b * A.T + tile] = tot;
      tileBase = tot ? (u32)atomicAdd(&A.d_m2[b], (int32_t)tot) : 0u;
    }
  }
  __syncthreads();
  const u32 tbase = tileBase;
  const uint64_t lt = kz_lanemask_lt();
  const u32 hcap = (u32)min(h, n);
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    if (rk[r] & BW_LIVE) {
      const int s = base + r * 64 + lane;
      const int64_t j = (int64_t)s + h;
      // positions past the end (s + h >= n): n - s in [1, min(h, n)], smaller for the shorter suffix and below every real rank
      const u64 r2 = (j < n) ? (u64)(rank[j] & ~BW_LIVE) + (u64)hcap + 1ULL : (u64)(n - s);
      const u32 pos = tbase + rowCnt[wave * RS_ITEMS + r] + (u32)__popcll(bal[r] & lt);
      keyN[off + pos] = ((u64)(rk[r] & ~BW_LIVE) << bitsR) | r2;
      valN[off + pos] = (u32)s;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Round 0 as a TRIE round: count first, move once.
// The seven LSD passes of the old round 0 moved every suffix seven times (24 B per pass) to learn, for three of
// the five data classes, that half of the suffixes sit in groups of many thousand equal 7-byte prefixes which no
// amount of sorting can split.  Here the text is only COUNTED, level by level, until every prefix class is either
// small enough to be finished in LDS or known to be one large group:
//   node   = a prefix of d bytes shared by more than TR_CAP suffixes (depth-1 nodes = the 256 first bytes, always
//            expanded); its 256 children are counted by the next byte (k_tr_hist16 for depth 1: LDS histogram of byte
//            pairs; k_tr_count for the deeper levels: LDS counters of the level's nodes, one info lookup per suffix);
//   k_tr_assign turns the counts of a level into child ranges of the suffix array (slots) and classifies each child:
//            EXPANDED (count > TR_CAP and depth < Dmax: a node of the next level), TERMINAL (count > TR_CAP at depth
//            Dmax: one group, nothing to sort), or SMALL: consecutive small siblings are merged greedily into buckets
//            of at most TR_CAP suffixes (a bucket is a contiguous slot range);
//   k_tr_scatter moves every suffix ONCE: terminal ones just get their group's head slot as rank (a coalesced store in
//            text order, they are never moved at all); the others go to their bucket as one 8-byte element
//            (the 64 - bitsG key bits that follow the bucket's common prefix | suffix index), staged per tile in LDS
//            so that a bucket's elements leave in runs;
//   k_tr_sort sorts one bucket in LDS (LSD radix over the bits that differ inside the bucket only), finds the groups
//            of equal keys and writes ranks and final suffixes like k_seg_apply.
// Depth: a bucket whose common prefix has d >= 1 bytes is sorted to d + (64 - bitsG) / 8 >= 6 bytes (zero padded at the
// end of the text like the old keys), terminal groups share Dmax >= 6 bytes.  The ranks are therefore a refinement of
// the 6-byte order and a coarsening of the suffix order, which is all the doubling rounds (h = 6, 12, ...) need: equal
// rank => equal 6-byte prefix; different rank => the true order.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "the trie rounds are sized for gfx950's 160 KiB of LDS per workgroup (k_tr_hist16 128 KiB, k_tr_scatter / k_trk_scatter 152 KiB, the bucket sorts 77 KiB)"
#endif
#define TR_CAP 7680u             // LDS sort capacity (60 KiB of elements + 16 KiB of counters: two workgroups per CU)
#define TR_MERGE (TR_CAP / 2)    // siblings above this are a bucket of their own
#define TR_K_SMALL 1u
#define TR_K_EXP 2u
#define TR_K_TERM 3u
#define TR_UNCHANGED (1u << 29)     // key rounds, terminal child: the new group keeps the old group's head slot (its members' ranks stay)
#define TR_NODECHUNK 128         // nodes whose counters fit the LDS of k_tr_count
struct TrieArrays {
  u32* cnt;        // [B][MN][256] children counts
  u32* info;       // [B][MN][256] kind << 30 | payload (SMALL: skip << 16 | bucket; EXP: node; TERM: head slot)
  u32* nodeStart;  // [B][MN] first slot of the node's range
  u32* bStart;     // [B][MB]
  u32* bCount;     // [B][MB]
  u32* bFill;      // [B][MB] elements placed so far (k_tr_scatter)
  u32* bAux;       // [B][MB] key rounds: skip >= 3: slot of the bucket's first element (g + bStart - group start); skip < 3: the bucket's prefix bytes
  u32* bG;         // [B][MB] key rounds, skip >= 3: the old group g all elements of the bucket belong to
  u32* nodePfx;    // [B][MN] key rounds: the node's prefix bytes (depth < 3) or its old group g (depth >= 3)
  u32* nodeGS;     // [B][MN] key rounds, depth >= 3: position of the old group's first element in the sorted window
  int32_t* meta;   // [B][TR_META]
  int32_t* err;    // [1] set when a table overflows (cannot happen for blocks the host admits: see tr_max_nodes / tr_max_buckets)
  int MN, MB;
};

__device__ __forceinline__ u32 tr_byte(const u8* __restrict__ s, int i, int n) { return i < n ? (u32)s[i] : 0u; }

// level 1: histogram of byte pairs (x[i], x[i+1]), x[n] = 0, one half of the 65536 bins per workgroup
__global__ __launch_bounds__(1024) void k_tr_hist16(const u8* __restrict__ srcAll, int64_t stride, BwtArrays A, TrieArrays T) {
  const int b = blockIdx.y;
  const int n = A.d_n[b];
  const u32 half = blockIdx.x;
  __shared__ u32 hist[32768];
  for (int i = threadIdx.x; i < 32768; i += 1024) hist[i] = 0;
  __syncthreads();
  const u8* s = srcAll + (int64_t)b * stride;
  const int lane = kz_lane();
  for (int base = 0; base < n; base += 1024 * 16) {
    const int p = base + threadIdx.x * 16;
    uint4 q = make_uint4(0, 0, 0, 0);
    if (p < n) q = *(const uint4*)(s + p);                               // blocks are 256-byte aligned with >= 4 KiB of slack
    u32 nxt = (u32)__shfl_down((int)(q.x & 0xFFu), 1, 64);
    if (lane == 63) nxt = (p + 16 < n) ? (u32)s[p + 16] : 0u;
    const u32 w[5] = {q.x, q.y, q.z, q.w, nxt};
    u32 runKey = 0xFFFFFFFFu, runCnt = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int i = p + j;
      const u32 c0 = (w[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
      const u32 c1r = (w[(j + 1) >> 2] >> (((j + 1) & 3) * 8)) & 0xFFu;
      const u32 c1 = (i + 1 < n) ? c1r : 0u;
      const u32 key = (c0 << 8) | c1;
      const bool mine = i < n && (key >> 15) == half;
      if (mine && key == runKey) runCnt++;
      else {
        if (runCnt) atomicAdd(&hist[runKey & 32767u], runCnt);
        runKey = mine ? key : 0xFFFFFFFFu; runCnt = mine ? 1u : 0u;
      }
    }
    if (runCnt) atomicAdd(&hist[runKey & 32767u], runCnt);
  }
  __syncthreads();
  u32* out = T.cnt + (int64_t)b * T.MN * 256 + half * 32768;
  for (int i = threadIdx.x; i < 32768; i += 1024) out[i] = hist[i];
}

// classify the children of the nodes of depth L (thread = node, its 256 children in order)
__global__ __launch_bounds__(256) void k_tr_assign(BwtArrays A, TrieArrays T, int L, int Dmax, int keys) {
  const int b = bw_row_block(A, blockIdx.x);
  int32_t* meta = T.meta + (int64_t)b * TR_META;
  __shared__ u32 sNodes, sBuckets, sErr;
  __shared__ u32 scan[32];
  const int lo = (L == 1) ? 0 : meta[2 + L], hi = (L == 1) ? 256 : meta[10 + L];
  if (threadIdx.x == 0) { sNodes = (L == 1) ? 256u : (u32)meta[0]; sBuckets = (L == 1) ? 0u : (u32)meta[1]; sErr = 0; }
  __syncthreads();
  const u32 firstNew = sNodes;
  u32* cnt = T.cnt + (int64_t)b * T.MN * 256;
  u32* info = T.info + (int64_t)b * T.MN * 256;
  u32* nodeStart = T.nodeStart + (int64_t)b * T.MN;
  u32* bStart = T.bStart + (int64_t)b * T.MB;
  u32* bCount = T.bCount + (int64_t)b * T.MB;
  u32* bAux = T.bAux + (int64_t)b * T.MB;
  u32* bG = T.bG + (int64_t)b * T.MB;
  u32* nodePfx = T.nodePfx + (int64_t)b * T.MN;
  u32* nodeGS = T.nodeGS + (int64_t)b * T.MN;
  const u32 skipTag = (u32)L << 16;                                      // packed next to a bucket's count (k_tr_sort reads it)
  for (int base = lo; base < hi; base += 256) {                          // uniform
    const int node = base + threadIdx.x;
    const bool valid = node < hi;
    u32 start = 0;
    if (L == 1) {                                                        // depth-1 nodes: their ranges from the byte totals
      u32 tot = 0;
      for (int c = 0; c < 256; c += 4) {
        const uint4 v = *(const uint4*)(cnt + (int64_t)node * 256 + c);
        tot += v.x + v.y + v.z + v.w;
      }
      u32 total;
      start = kz_wg_excl_sum(tot, scan, &total);
      nodeStart[node] = start;
      if (keys) nodePfx[node] = (u32)node;
    } else if (valid) start = nodeStart[node];
    if (!valid) continue;
    u32 run = start, curCnt = 0;
    int curB = -1;
    // key rounds (see k_trk_*): the first three key bytes are the old group g; a node of depth >= 3 knows g and where g's elements start
    const u32 pfxN = keys ? nodePfx[node] : 0u;
    const u32 gsN = (keys && L >= 3) ? nodeGS[node] : 0u;
    const u32 auxSmall = !keys ? 0u : (L >= 3 ? pfxN - gsN : pfxN);      // skip >= 3: slot = aux + position; skip < 3: the prefix bytes
    for (int c4 = 0; c4 < 256; c4 += 4) {
      const uint4 v4 = *(const uint4*)(cnt + (int64_t)node * 256 + c4);
      const u32 cc4[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const u32 cc = cc4[j];
        if (cc == 0) continue;
        const u32 s0 = run;
        run += cc;
        u32 e;
        const u32 cbyte = (u32)(c4 + j);
        if (cc > TR_CAP) {
          if (curB >= 0) { bCount[curB] = curCnt | skipTag; curB = -1; }
          if (L + 1 < Dmax) {
            const u32 M = atomicAdd(&sNodes, 1u);
            if (M < (u32)T.MN) {
              nodeStart[M] = s0; e = (TR_K_EXP << 30) | M;
              uint4* z = (uint4*)(cnt + (int64_t)M * 256);                // the new node's counters start at zero (no memset of the whole table)
              for (int q = 0; q < 64; q++) z[q] = make_uint4(0, 0, 0, 0);
              if (keys) { nodePfx[M] = (L < 3) ? ((pfxN << 8) | cbyte) : pfxN; nodeGS[M] = (L + 1 == 3) ? s0 : gsN; }
            } else { sErr = 1; e = (TR_K_TERM << 30) | s0; }
          } else if (keys) {                                              // all key bytes equal: one new group; its head slot = g + (s0 - group start)
            const u32 head = pfxN + (s0 - gsN);
            e = (TR_K_TERM << 30) | (s0 == gsN ? TR_UNCHANGED : 0u) | (head & 0xFFFFFFu);
          } else e = (TR_K_TERM << 30) | s0;
        } else if (cc > TR_MERGE) {
          if (curB >= 0) { bCount[curB] = curCnt | skipTag; curB = -1; }
          const u32 id = atomicAdd(&sBuckets, 1u);
          if (id < (u32)T.MB) { bStart[id] = s0; bCount[id] = cc | skipTag; bAux[id] = (keys && L >= 3) ? auxSmall + s0 : auxSmall; bG[id] = pfxN; } else sErr = 1;
          e = (TR_K_SMALL << 30) | ((u32)L << 16) | (id & 0xFFFFu);
        } else {
          if (curB >= 0 && curCnt + cc <= TR_CAP) curCnt += cc;
          else {
            if (curB >= 0) bCount[curB] = curCnt | skipTag;
            const u32 id = atomicAdd(&sBuckets, 1u);
            if (id < (u32)T.MB) { bStart[id] = s0; curB = (int)id; bAux[id] = (keys && L >= 3) ? auxSmall + s0 : auxSmall; bG[id] = pfxN; } else { sErr = 1; curB = -1; }
            curCnt = cc;
          }
          e = (TR_K_SMALL << 30) | ((u32)L << 16) | ((u32)curB & 0xFFFFu);
        }
        info[(int64_t)node * 256 + c4 + j] = e;
      }
    }
    if (curB >= 0) bCount[curB] = curCnt | skipTag;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    meta[0] = (int32_t)min(sNodes, (u32)T.MN); meta[1] = (int32_t)min(sBuckets, (u32)T.MB);
    meta[2 + L + 1] = (int32_t)firstNew; meta[10 + L + 1] = (int32_t)min(sNodes, (u32)T.MN);
    if (sErr) { meta[18] = 1; atomicOr(T.err, 1); }
  }
}

// level L >= 2: every suffix that is still inside an expanded node of depth L-1 looks its child up; children that were
// expanded (nodes of depth L) count the suffix's byte L in LDS.  state[i] = the suffix's leaf, or node | depth << 16.
#define TRC_U 4                  // quads of suffixes a thread keeps in flight
__global__ __launch_bounds__(1024) void k_tr_count(const u8* __restrict__ srcAll, int64_t stride, u32* __restrict__ stateAll, BwtArrays A, TrieArrays T, int L) {
  const int b = blockIdx.y;
  const int32_t* meta = T.meta + (int64_t)b * TR_META;
  const int lo = meta[2 + L], hi = meta[10 + L];
  if (hi <= lo) return;                                                  // no node of this depth: the states stay as they are
  const int n = A.d_n[b];
  __shared__ u32 lds[TR_NODECHUNK * 256];
  const u8* s = srcAll + (int64_t)b * stride;
  u32* state = stateAll + (int64_t)b * A.NS;
  const u32* info = T.info + (int64_t)b * T.MN * 256;
  u32* cnt = T.cnt + (int64_t)b * T.MN * 256;
  const int P = gridDim.x;
  const int per = (((n + P - 1) / P) + 1023) & ~1023;
  const int pbeg = blockIdx.x * per, pend = min(n, pbeg + per);
  for (int chunk = 0; lo + chunk * TR_NODECHUNK < hi; chunk++) {
    for (int i = threadIdx.x; i < TR_NODECHUNK * 256; i += 1024) lds[i] = 0;
    __syncthreads();
    const int nlo = lo + chunk * TR_NODECHUNK;
    // a thread takes four CONSECUTIVE suffixes: one 16-byte state access and one 8-byte text window serve all four, equal
    // targets in a row (runs) are added once; two such quads per thread are in flight (the chain state -> info -> counter is latency)
    for (int i0 = pbeg; i0 < pend; i0 += TRC_U * 4096) {
      uint4 st4[TRC_U]; u64 win[TRC_U]; bool any[TRC_U];
#pragma unroll
      for (int u = 0; u < TRC_U; u++) {
        const int i = i0 + u * 4096 + threadIdx.x * 4;
        any[u] = i < pend;
        st4[u] = make_uint4(0xC0000000u, 0xC0000000u, 0xC0000000u, 0xC0000000u);
        win[u] = 0;
        if (any[u]) {
          if (!(L == 2 && chunk == 0)) st4[u] = *(const uint4*)(state + i);
          const int at = i + L - 2;                                        // bytes at .. at + 5: previous byte, child byte, counted byte of the four
          u64 k = __builtin_bswap64(*(const bw_u64_unaligned*)(s + at));
          const int rem = n - at;
          if (rem < 8) k = rem <= 0 ? 0ULL : (k & (~0ULL << (8 * (8 - rem))));
          win[u] = k;
        }
      }
      // the info lookups of ALL the quads are issued before any of them is used: one workgroup per CU (128 KiB of counters)
      // hides the chain state -> info -> counter only by what a thread keeps in flight
      u32 stqA[TRC_U][4]; u32 eA[TRC_U][4]; bool lookA[TRC_U][4];
#pragma unroll
      for (int u = 0; u < TRC_U; u++) {
        const int i = i0 + u * 4096 + threadIdx.x * 4;
        stqA[u][0] = st4[u].x; stqA[u][1] = st4[u].y; stqA[u][2] = st4[u].z; stqA[u][3] = st4[u].w;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (L == 2 && chunk == 0) stqA[u][q] = (any[u] && i + q < pend) ? ((1u << 16) | (u32)((win[u] >> (56 - 8 * q)) & 0xFFu)) : 0xC0000000u;
          lookA[u][q] = any[u] && (i + q < pend) && (stqA[u][q] >> 30) == 0 && ((stqA[u][q] >> 16) & 7u) != (u32)L;
          eA[u][q] = 0;
          if (lookA[u][q]) eA[u][q] = info[(stqA[u][q] & 0xFFFFu) * 256 + (u32)((win[u] >> (48 - 8 * q)) & 0xFFu)];
        }
      }
#pragma unroll
      for (int u = 0; u < TRC_U; u++) {
        if (!any[u]) continue;
        const int i = i0 + u * 4096 + threadIdx.x * 4;
        u32 (&stq)[4] = stqA[u];
        u32 (&e)[4] = eA[u]; bool (&look)[4] = lookA[u];
        bool changed = false;
        u32 runT = 0xFFFFFFFFu, runC = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          u32 tgt = 0xFFFFFFFFu;
          if ((i + q < pend) && (stq[q] >> 30) == 0) {
            if (look[q]) {
              stq[q] = ((e[q] >> 30) == TR_K_EXP) ? ((e[q] & 0xFFFFu) | ((u32)L << 16)) : e[q];
              changed = true;
            }
            if ((stq[q] >> 30) == 0) {                                       // inside a node of depth L now
              const int k = (int)(stq[q] & 0xFFFFu) - nlo;
              if (k >= 0 && k < TR_NODECHUNK) tgt = (u32)k * 256 + (u32)((win[u] >> (40 - 8 * q)) & 0xFFu);
            }
          }
          if (tgt == runT) runC++;
          else {
            if (runC && runT != 0xFFFFFFFFu) atomicAdd(&lds[runT], runC);
            runT = tgt; runC = 1;
          }
        }
        if (runC && runT != 0xFFFFFFFFu) atomicAdd(&lds[runT], runC);
        if (changed) *(uint4*)(state + i) = make_uint4(stq[0], stq[1], stq[2], stq[3]);
      }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < TR_NODECHUNK * 256; j += 1024) {
      const u32 v = lds[j];
      if (v && nlo + (j >> 8) < hi) atomicAdd(&cnt[(int64_t)(nlo + (j >> 8)) * 256 + (j & 255)], v);
    }
    __syncthreads();
  }
}

// move every suffix once (see above).  Tile = 8192 suffixes in text order.  The tile's count per bucket lives in
// 16-bit halves of LDS words (a tile holds at most 8192 suffixes), so that TR_MAXB buckets fit next to the staging buffer.
#define TRS_TILE 8192
#define TRS_ITEMS (TRS_TILE / 1024)
#define TR_MAXB 12288
// Two sizes of the same kernel (round 5): blocks with at most TRS_FASTB buckets (every block measured: 456 .. 790) take tiles of 4096
// suffixes with 52 KiB of LDS -- three workgroups per CU to hide the state -> info -> text chain -- the others the full tables
// (152 KiB, one workgroup per CU).  Both are launched; a block's workgroups of the form that is not its own return at once.
#define TRS_FASTB 2048
template <int TILE_, int MAXB_, bool FAST>
__device__ __forceinline__ void tr_scatter_body(const u8* __restrict__ srcAll, int64_t stride, const u32* __restrict__ stateAll,
                                                u64* __restrict__ elemAll, const BwtArrays& A, const TrieArrays& T, int bitsG) {
  constexpr int ITEMS_ = TILE_ / 1024;
  if ((T.meta[(int64_t)blockIdx.y * TR_META + 1] <= TRS_FASTB) != FAST) return;
  const int b = blockIdx.y;
  const int n = A.d_n[b];
  const int tile = blockIdx.x;
  const int tbase = tile * TILE_;
  if (tbase >= n) return;
  __shared__ u32 tc2[MAXB_ / 2];                                       // two 16-bit counters per word
  __shared__ u32 gdelta[MAXB_];
  __shared__ u64 stage[TILE_];
  __shared__ uint16_t stageB[TILE_];
  __shared__ u32 scan[32];
  const int32_t* meta = T.meta + (int64_t)b * TR_META;
  const int nB = meta[1];
  const bool hasState = meta[10 + 2] > meta[2 + 2];                      // a level-2 count pass wrote the states
  for (int i = threadIdx.x; i < (nB + 1) / 2; i += 1024) tc2[i] = 0;
  __syncthreads();
  const u8* s = srcAll + (int64_t)b * stride;
  const u32* state = stateAll + (int64_t)b * A.NS;
  const u32* info = T.info + (int64_t)b * T.MN * 256;
  u32* rank = A.rank + (int64_t)b * A.NS;
  const u64 lowMask = (1ULL << bitsG) - 1ULL;
  u64 el[ITEMS_]; u32 bp[ITEMS_];                                  // element, bucket | position in (tile, bucket) << 16
  // three rounds of loads for the tile's eight items per thread, each round issued for all items before its results are used
  // (state -> info -> text is a dependent chain; one item after the other was 24 memory round trips per thread, the kernel's time)
  u32 stv[ITEMS_];
#pragma unroll
  for (int r = 0; r < ITEMS_; r++) {
    const int i = tbase + r * 1024 + threadIdx.x;
    stv[r] = 0xC0000000u;
    if (i < n) stv[r] = hasState ? state[i] : ((u32)s[i] | (1u << 16));
  }
  u32 cb[ITEMS_];
#pragma unroll
  for (int r = 0; r < ITEMS_; r++) {
    const int i = tbase + r * 1024 + threadIdx.x;
    cb[r] = 0;
    if (i < n && (stv[r] >> 30) == 0) cb[r] = tr_byte(s, i + (int)((stv[r] >> 16) & 7u), n);
  }
#pragma unroll
  for (int r = 0; r < ITEMS_; r++) {
    const int i = tbase + r * 1024 + threadIdx.x;
    if (i < n && (stv[r] >> 30) == 0) stv[r] = info[(stv[r] & 0xFFFFu) * 256 + cb[r]];
  }
  u64 kv[ITEMS_];
#pragma unroll
  for (int r = 0; r < ITEMS_; r++) {
    const int i = tbase + r * 1024 + threadIdx.x;
    kv[r] = 0;
    if (i < n && (stv[r] >> 30) != TR_K_TERM) {
      const int at = i + (int)((stv[r] >> 16) & 7u);
      u64 k = __builtin_bswap64(*(const bw_u64_unaligned*)(s + at));
      const int rem = n - at;
      if (rem < 8) k = rem <= 0 ? 0ULL : (k & (~0ULL << (8 * (8 - rem))));
      kv[r] = k;
    }
  }
#pragma unroll
  for (int r = 0; r < ITEMS_; r++) {
    const int i = tbase + r * 1024 + threadIdx.x;
    bp[r] = 0xFFFFFFFFu;
    el[r] = 0;
    if (i < n) {
      const u32 st = stv[r];
      if ((st >> 30) == TR_K_TERM) rank[i] = (st & 0xFFFFFFu) | BW_LIVE;
      else {
        const u32 bk = st & 0xFFFFu;
        el[r] = (kv[r] & ~lowMask) | (u64)(u32)i;
        const u32 old = atomicAdd(&tc2[bk >> 1], (bk & 1u) ? 65536u : 1u);
        const u32 pos = (bk & 1u) ? (old >> 16) : (old & 0xFFFFu);
        bp[r] = bk | (pos << 16);
      }
    }
  }
  __syncthreads();
  // exclusive scan of the tile's bucket counts (a thread owns whole words); buckets present reserve their run in the bucket's slot range
  {
    const int per = 2 * ((nB + 2047) / 2048);
    const int j0 = threadIdx.x * per;
    u32 mine = 0;
    for (int j = j0; j < j0 + per && j < nB; j += 2) { const u32 w = tc2[j >> 1]; mine += (w & 0xFFFFu) + (w >> 16); }
    u32 total;
    u32 run = kz_wg_excl_sum(mine, scan, &total);
    const u32* bStart = T.bStart + (int64_t)b * T.MB;
    u32* bFill = T.bFill + (int64_t)b * T.MB;
    for (int j = j0; j < j0 + per && j < nB; j += 2) {
      const u32 w = tc2[j >> 1];
      const u32 c0 = w & 0xFFFFu, c1 = w >> 16;
      if (c0) gdelta[j] = bStart[j] + atomicAdd(&bFill[j], c0) - run;
      const u32 r1 = run + c0;
      if (c1) gdelta[j + 1] = bStart[j + 1] + atomicAdd(&bFill[j + 1], c1) - r1;
      tc2[j >> 1] = run | (r1 << 16);                                    // tile-local start of both buckets (< 8192 unless the bucket is empty)
      run = r1 + c1;
    }
    if (threadIdx.x == 0) scan[31] = total;
  }
  __syncthreads();
  const u32 total = scan[31];
#pragma unroll
  for (int r = 0; r < ITEMS_; r++) {
    if (bp[r] != 0xFFFFFFFFu) {
      const u32 bk = bp[r] & 0xFFFFu;
      const u32 w = tc2[bk >> 1];
      const u32 slot = ((bk & 1u) ? (w >> 16) : (w & 0xFFFFu)) + (bp[r] >> 16);
      stage[slot] = el[r]; stageB[slot] = (uint16_t)bk;
    }
  }
  __syncthreads();
  u64* elem = elemAll + (int64_t)b * A.NS;
#pragma unroll
  for (int r = 0; r < ITEMS_; r++) {
    const u32 slot = (u32)r * 1024 + threadIdx.x;
    if (slot < total) elem[gdelta[stageB[slot]] + slot] = stage[slot];
  }
}

__global__ __launch_bounds__(1024) void k_tr_scatter(const u8* __restrict__ srcAll, int64_t stride, const u32* __restrict__ stateAll,
                                                      u64* __restrict__ elemAll, BwtArrays A, TrieArrays T, int bitsG) { tr_scatter_body<TRS_TILE, TR_MAXB, false>(srcAll, stride, stateAll, elemAll, A, T, bitsG); }
__global__ __launch_bounds__(1024) void k_tr_scatter_f(const u8* __restrict__ srcAll, int64_t stride, const u32* __restrict__ stateAll,
                                                        u64* __restrict__ elemAll, BwtArrays A, TrieArrays T, int bitsG) { tr_scatter_body<TRS_TILE / 2, TRS_FASTB, true>(srcAll, stride, stateAll, elemAll, A, T, bitsG); }

// round 0: a block whose trie has only the 256 depth-1 nodes (see the lazy ranks of k_tr_sort)
__device__ __forceinline__ bool tr_lazy_candidate(const int32_t* meta) { return meta[0] == 256 && meta[18] == 0; }
// one bucket at a time in LDS: sort by the key bits that differ, groups of equal keys, ranks and final suffixes
#define TRQ_WAVES 16
#define TRQ_ROWS 8
template <bool KEYS>
__device__ __forceinline__ void tr_sort_body(const u64* __restrict__ elemAll, const BwtArrays& A, const TrieArrays& T, int bitsG, int lazyMode) {
  const int b = bw_row_block(A, blockIdx.y);
  const int nB = T.meta[(int64_t)b * TR_META + 1];
  // LAZY RANKS (round 0 only).  A block without a single expanded node (no 2-byte prefix above TR_CAP suffixes: high-entropy data)
  // almost always leaves round 0 with every suffix final, and then nobody reads its ranks except k_bwt_emit (the 8 primary indexes):
  // the scattered 4-byte rank stores are a quarter of this kernel's time on such blocks.  Pass 1 (lazyMode 1) sorts such a block,
  // writes its final suffixes and the 8 ranks k_bwt_emit reads, and COUNTS the live suffixes in meta[20]; pass 2 (lazyMode 2) runs the
  // blocks whose count is not zero again, storing every rank (the elements are untouched by pass 1; the sa stores repeat).
  // k_live_emit skips the blocks pass 1 finished.
  const bool lazyCand = !KEYS && lazyMode != 0 && tr_lazy_candidate(T.meta + (int64_t)b * TR_META);
  if (!KEYS && lazyMode == 2 && !(lazyCand && T.meta[(int64_t)b * TR_META + 20] > 0)) return;
  const bool lazy = lazyCand && lazyMode == 1;
  u32 pstep = 0;
  if (lazy) { const int n = A.d_n[b]; const int st = n >> 3; pstep = (u32)((st * 8 != n) ? st + 1 : st); }   // k_bwt_emit: primary index k = rank[k * step]
  u32 liveMine = 0;
  __shared__ u64 buf[TR_CAP];
  __shared__ uint16_t cw[TRQ_WAVES][BK_DBINS];
  __shared__ u32 wsum[BK_DBINS / 64];
  __shared__ u32 wH[TRQ_WAVES], wS[TRQ_WAVES];
  __shared__ u64 redO[TRQ_WAVES], redA[TRQ_WAVES];
  __shared__ int skip[8];
  const int wave = threadIdx.x >> 6, lane = kz_lane();
  const int64_t off = (int64_t)b * A.NS;
  const u64* elem = elemAll + off;
  u32* rank = A.rank + off;
  u32* sa = A.sa + off;
  const uint64_t lt = kz_lanemask_lt();
  const uint64_t le = lt | (1ULL << lane);
  const u64 vmask = (1ULL << bitsG) - 1ULL;
  for (int d = blockIdx.x; d < nB; d += gridDim.x) {
    const u32 bcs = T.bCount[(int64_t)b * T.MB + d];
    const int cnt = (int)(bcs & 0xFFFFu);
    const int skipB = (int)(bcs >> 16);                                  // bytes of common prefix in front of the bucket's key fragment
    const u32 bo = T.bStart[(int64_t)b * T.MB + d];
    const u32 aux = KEYS ? T.bAux[(int64_t)b * T.MB + d] : 0u;
    const u32 gOne = KEYS ? T.bG[(int64_t)b * T.MB + d] : 0u;
    const int rows = (cnt + 63) >> 6;
    const int R = (rows + TRQ_WAVES - 1) / TRQ_WAVES;                    // rows per wave, 1..TRQ_ROWS (uniform)
    const int base = wave * R * 64;
    u64 k[TRQ_ROWS]; u32 dr[TRQ_ROWS];
    u64 vo = 0, va = ~0ULL;
#pragma unroll
    for (int r = 0; r < TRQ_ROWS; r++) {
      const int idx = base + r * 64 + lane;
      k[r] = ~0ULL;
      if (r < R && idx < cnt) { k[r] = elem[bo + idx]; vo |= k[r]; va &= k[r]; }
    }
    for (int dd = 32; dd > 0; dd >>= 1) { vo |= __shfl_xor(vo, dd, 64); va &= __shfl_xor(va, dd, 64); }
    if (threadIdx.x < 8) skip[threadIdx.x] = 0;
    if (lane == 0) { redO[wave] = vo; redA[wave] = va; }
    __syncthreads();
    vo = 0; va = ~0ULL;
#pragma unroll
    for (int w = 0; w < TRQ_WAVES; w++) { vo |= redO[w]; va &= redA[w]; }
    const u64 diff = (vo ^ va) >> bitsG;                                 // key bits that differ inside the bucket
    int passes = 0, lowBit = 0;
    if (diff) {
      lowBit = (int)__builtin_ctzll(diff);
      const int hiBit = 63 - (int)__builtin_clzll(diff);
      passes = (hiBit - lowBit + BK_DBITS) / BK_DBITS;
    }
    for (int p = 0; p < passes; p++) {
      const int shift = bitsG + lowBit + BK_DBITS * p;
      for (int i = threadIdx.x; i < TRQ_WAVES * BK_DBINS / 2; i += TRQ_WAVES * 64) ((u32*)&cw[0][0])[i] = 0;
      __syncthreads();
#pragma unroll
      for (int r = 0; r < TRQ_ROWS; r++) {
        if (r < R) {
          const int idx = base + r * 64 + lane;
          const bool valid = idx < cnt;
          const u32 dg = (u32)(k[r] >> shift) & (BK_DBINS - 1);
          const uint64_t vm = kz_ballot(valid);
          if (vm == 0) continue;
          const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)dg);
          const bool uni = kz_ballot(valid && dg == d0) == vm;
          const uint64_t peers = uni ? (valid ? vm : 0ULL) : bw_match<BK_DBITS>(dg, valid);
          u32 pre = 0;
          if (valid) pre = cw[wave][dg];
          const u32 rnk = pre + (u32)__popcll(peers & lt);
          if (valid && (peers >> lane) == 1ULL) cw[wave][dg] = (uint16_t)(pre + (u32)__popcll(peers));
          dr[r] = dg | (rnk << 16);
        }
      }
      __syncthreads();
      static_assert(BK_DBINS <= TRQ_WAVES * 64, "one digit per thread");
      u32 mine = 0;                                                      // thread = digit (the threads behind carry zero)
      if (threadIdx.x < BK_DBINS) {
#pragma unroll
        for (int w = 0; w < TRQ_WAVES; w++) mine += cw[w][threadIdx.x];
        if (mine == (u32)cnt) skip[p] = 1;
      }
      const u32 inc = kz_wave_incl_sum(mine);
      if (lane == 63 && wave < BK_DBINS / 64) wsum[wave] = inc;
      __syncthreads();
      if (skip[p]) continue;                                             // uniform: one digit value for the whole bucket, nothing moves
      if (threadIdx.x < BK_DBINS) {
        u32 run = inc - mine;
        for (int w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
        for (int w = 0; w < TRQ_WAVES; w++) { const u32 cv = cw[w][threadIdx.x]; cw[w][threadIdx.x] = (uint16_t)run; run += cv; }
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < TRQ_ROWS; r++) {
        if (r < R) {
          const int idx = base + r * 64 + lane;
          if (idx < cnt) buf[cw[wave][dr[r] & 0xFFFF] + (dr[r] >> 16)] = k[r];
        }
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < TRQ_ROWS; r++) {
        if (r < R) {
          const int idx = base + r * 64 + lane;
          if (idx < cnt) k[r] = buf[idx];
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < TRQ_ROWS; r++) {
      if (r < R) {
        const int idx = base + r * 64 + lane;
        if (idx < cnt) buf[idx] = k[r];
      }
    }
    __syncthreads();
    // groups of equal keys; key rounds with skip < 3 also need the OLD groups (the first 3 - skip key bytes): an element's slot
    // is g + its index inside its old group, a new group's rank g + the index of its first element inside the old group
    const bool segs = KEYS && skipB < 3;
    const int gsh = 64 - 8 * (3 - skipB);                                 // element >> gsh = the bytes of g the fragment holds
    u32 mh = 0, ms = 0;
#pragma unroll
    for (int r = 0; r < TRQ_ROWS; r++) {
      if (r < R) {
        const int idx = base + r * 64 + lane;
        const bool valid = idx < cnt;
        const u64 prev = (valid && idx > 0) ? buf[idx - 1] : 0;
        const bool head = valid && (idx == 0 || (k[r] >> bitsG) != (prev >> bitsG));
        const uint64_t hbr = kz_ballot(head);
        if (hbr) mh = (u32)(base + r * 64 + 63 - (int)__builtin_clzll(hbr)) + 1;
        if (segs) {
          const uint64_t sbr = kz_ballot(head && (idx == 0 || (k[r] >> gsh) != (prev >> gsh)));
          if (sbr) ms = (u32)(base + r * 64 + 63 - (int)__builtin_clzll(sbr)) + 1;
        }
      }
    }
    if (lane == 0) { wH[wave] = mh; wS[wave] = ms; }
    __syncthreads();
    u32 carH = 0, carS = 0;
    for (int w = 0; w < wave; w++) { carH = max(carH, wH[w]); carS = max(carS, wS[w]); }
#pragma unroll
    for (int r = 0; r < TRQ_ROWS; r++) {
      if (r < R) {
        const int rowBase = base + r * 64;
        const int idx = rowBase + lane;
        const bool valid = idx < cnt;
        const u64 prev = (valid && idx > 0) ? buf[idx - 1] : 0;
        const bool head = valid && (idx == 0 || (k[r] >> bitsG) != (prev >> bitsG));
        const uint64_t hbr = kz_ballot(head);
        const uint64_t hbl = hbr & le;
        const u32 hh = hbl ? (u32)(rowBase + 63 - (int)__builtin_clzll(hbl)) : carH - 1;
        uint64_t sbr = 0;
        u32 ss = 0;
        if (segs) {
          sbr = kz_ballot(head && (idx == 0 || (k[r] >> gsh) != (prev >> gsh)));
          const uint64_t sbl = sbr & le;
          ss = sbl ? (u32)(rowBase + 63 - (int)__builtin_clzll(sbl)) : carS - 1;
        }
        if (valid) {
          const u64 nextk = (idx + 1 < cnt) ? buf[idx + 1] : 0;
          const bool headN = (idx + 1 >= cnt) || ((nextk >> bitsG) != (k[r] >> bitsG));
          const bool live = !(hh == (u32)idx && headN);
          const u32 sv = (u32)(k[r] & vmask);
          if (!KEYS) {
            if (!lazy) rank[sv] = (bo + hh) | (live ? BW_LIVE : 0u);
            else {
              const u32 q = sv / pstep;
              if (q * pstep == sv && q < 8u) rank[sv] = (bo + hh) | (live ? BW_LIVE : 0u);
              liveMine += live ? 1u : 0u;
            }
            if (!live) sa[bo + (u32)idx] = sv;
          } else {
            u32 g, headSlot, slot;
            if (segs) {
              g = (aux << (8 * (3 - skipB))) | (u32)(k[r] >> gsh);
              headSlot = g + (hh - ss); slot = g + ((u32)idx - ss);
            } else { g = gOne; headSlot = aux + hh; slot = aux + (u32)idx; }
            // the first new group of an old group keeps the old head slot: while it stays LIVE its members' ranks do not change
            if (!(live && headSlot == g)) rank[sv] = headSlot | (live ? BW_LIVE : 0u);
            if (!live) sa[slot] = sv;
          }
        }
        if (hbr) carH = (u32)(rowBase + 63 - (int)__builtin_clzll(hbr)) + 1;
        if (sbr) carS = (u32)(rowBase + 63 - (int)__builtin_clzll(sbr)) + 1;
      }
    }
    __syncthreads();
  }
  if (!KEYS && lazy) {
    if (kz_ballot(liveMine != 0) != 0) {
      for (int dd = 32; dd > 0; dd >>= 1) liveMine += __shfl_xor(liveMine, dd, 64);
      if (lane == 0) atomicAdd(&T.meta[(int64_t)b * TR_META + 20], (int32_t)liveMine);
    }
  }
}
// (two workgroups per CU at 64 VGPRs with 16 registers spilled beat one workgroup without spills: 45 vs 50 ms per 357 uniform blocks)
__global__ __launch_bounds__(1024, 8) void k_tr_sort(const u64* __restrict__ elemAll, BwtArrays A, TrieArrays T, int bitsG, int lazyMode) { tr_sort_body<false>(elemAll, A, T, bitsG, lazyMode); }
__global__ __launch_bounds__(1024, 8) void k_trk_sort(const u64* __restrict__ elemAll, BwtArrays A, TrieArrays T, int bitsG) { tr_sort_body<true>(elemAll, A, T, bitsG, 0); }

// ---------------------------------------------------------------------------------------------
// KEY rounds: the same trie round over the WINDOW of a doubling round -- the compact pairs (old group g << bitsR | secondary key r2,
// suffix) of the buckets too large for the LDS kernels (groups of many thousand suffixes: six full LSD passes over HBM and the
// k_seg_* kernels until round 3).  The digits are the six bytes of K48 = g : 24 | r2 : 24; the first three bytes are the old
// group, so a child of depth 3 IS an old group and its start in the sorted window is where the group's elements begin: an
// element at sorted position pos of group g takes slot g + (pos - group start).  A child that is still large after all six
// bytes holds equal keys only: one new group, no sorting (terminal).  Buckets of merged small siblings above depth 3 hold whole
// old groups (k_trk_sort finds their boundaries in the sorted fragment), buckets below depth 3 lie inside one group.
struct KeySrc { const u64* key; const u32* val; int bitsR; };
__device__ __forceinline__ u64 trk_k48(u64 key, int bitsR) { return ((key >> bitsR) << 24) | (key & ((1ULL << bitsR) - 1ULL)); }

__global__ __launch_bounds__(1024) void k_trk_hist16(KeySrc X, BwtArrays A, TrieArrays T) {
  const int b = bw_row_block(A, blockIdx.y);
  const int w0 = A.d_w[b];
  const int W = max(0, A.d_m[b] - w0);                                  // an empty window still writes its (zero) counts: k_tr_assign reads them
  const u32 half = blockIdx.x;
  __shared__ u32 hist[32768];
  for (int i = threadIdx.x; i < 32768; i += 1024) hist[i] = 0;
  __syncthreads();
  const u64* key = X.key + (int64_t)b * A.NS + w0;
  const int lane = kz_lane();
  for (int j0 = 0; j0 < W; j0 += 1024) {
    const int j = j0 + threadIdx.x;
    u32 tgt = 0xFFFFFFFFu;
    if (j < W) {
      const u32 pair = (u32)(trk_k48(key[j], X.bitsR) >> 32);
      if ((pair >> 15) == half) tgt = pair & 32767u;
    }
    const uint64_t am = kz_ballot(tgt != 0xFFFFFFFFu);                   // the window is grouped by bucket: most rows hit one counter
    if (am) {
      const int l0 = (int)__builtin_ctzll(am);
      const u32 t0 = (u32)__shfl((int)tgt, l0, 64);
      const uint64_t same = kz_ballot(tgt == t0);
      if (lane == l0) atomicAdd(&hist[t0], (u32)__popcll(same));
      else if (tgt != 0xFFFFFFFFu && tgt != t0) atomicAdd(&hist[tgt], 1u);
    }
  }
  __syncthreads();
  u32* out = T.cnt + (int64_t)b * T.MN * 256 + half * 32768;
  for (int i = threadIdx.x; i < 32768; i += 1024) out[i] = hist[i];
}

__global__ __launch_bounds__(1024) void k_trk_count(KeySrc X, u32* __restrict__ stateAll, BwtArrays A, TrieArrays T, int L) {
  const int b = bw_row_block(A, blockIdx.y);
  const int32_t* meta = T.meta + (int64_t)b * TR_META;
  const int lo = meta[2 + L], hi = meta[10 + L];
  if (hi <= lo) return;
  const int w0 = A.d_w[b];
  const int W = A.d_m[b] - w0;
  __shared__ u32 lds[TR_NODECHUNK * 256];
  const u64* key = X.key + (int64_t)b * A.NS + w0;
  u32* state = stateAll + (int64_t)b * A.NS;
  const u32* info = T.info + (int64_t)b * T.MN * 256;
  u32* cnt = T.cnt + (int64_t)b * T.MN * 256;
  const int P = gridDim.x;
  const int per = (((W + P - 1) / P) + 1023) & ~1023;
  const int pbeg = blockIdx.x * per, pend = min(W, pbeg + per);
  for (int chunk = 0; lo + chunk * TR_NODECHUNK < hi; chunk++) {
    for (int i = threadIdx.x; i < TR_NODECHUNK * 256; i += 1024) lds[i] = 0;
    __syncthreads();
    const int nlo = lo + chunk * TR_NODECHUNK;
    // a thread takes four CONSECUTIVE items (two 16-byte key loads, one 16-byte state access); equal targets in a row are added once;
    // TRC_U such quads are in flight per thread: all their loads, then all their info lookups, then the counters
    for (int i0 = pbeg; i0 < pend; i0 += TRC_U * 4096) {
      u64 kq[TRC_U][4]; u32 stq[TRC_U][4]; bool act[TRC_U];
#pragma unroll
      for (int u = 0; u < TRC_U; u++) {
        const int i = i0 + u * 4096 + (int)threadIdx.x * 4;
        act[u] = i < pend;
#pragma unroll
        for (int q = 0; q < 4; q++) { stq[u][q] = 0xC0000000u; kq[u][q] = 0; }
        if (!act[u]) continue;
        if (!(L == 2 && chunk == 0)) {                                     // settled items (most of a late level's window) cost one state read
          const uint4 sv = *(const uint4*)(state + i);
          stq[u][0] = sv.x; stq[u][1] = sv.y; stq[u][2] = sv.z; stq[u][3] = sv.w;
          if ((stq[u][0] >> 30) && (stq[u][1] >> 30) && (stq[u][2] >> 30) && (stq[u][3] >> 30)) { act[u] = false; continue; }
        }
        if (i + 4 <= pend && ((((uintptr_t)(key + i)) & 15) == 0)) {
          const uint4 a = *(const uint4*)(key + i), c = *(const uint4*)(key + i + 2);
          kq[u][0] = ((u64)a.y << 32) | a.x; kq[u][1] = ((u64)a.w << 32) | a.z; kq[u][2] = ((u64)c.y << 32) | c.x; kq[u][3] = ((u64)c.w << 32) | c.z;
        } else {
#pragma unroll
          for (int q = 0; q < 4; q++) kq[u][q] = (i + q < pend) ? key[i + q] : 0ULL;
        }
      }
      u32 e[TRC_U][4]; bool look[TRC_U][4]; u64 k48[TRC_U][4];
#pragma unroll
      for (int u = 0; u < TRC_U; u++) {
        const int i = i0 + u * 4096 + (int)threadIdx.x * 4;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          k48[u][q] = trk_k48(kq[u][q], X.bitsR);
          if (L == 2 && chunk == 0) stq[u][q] = (act[u] && i + q < pend) ? ((1u << 16) | (u32)(k48[u][q] >> 40)) : 0xC0000000u;
          look[u][q] = act[u] && (i + q < pend) && (stq[u][q] >> 30) == 0 && ((stq[u][q] >> 16) & 7u) != (u32)L;
          e[u][q] = 0;
          if (look[u][q]) e[u][q] = info[(stq[u][q] & 0xFFFFu) * 256 + (u32)((k48[u][q] >> (40 - 8 * (L - 1))) & 0xFFu)];
        }
      }
#pragma unroll
      for (int u = 0; u < TRC_U; u++) {
        if (!act[u]) continue;
        const int i = i0 + u * 4096 + (int)threadIdx.x * 4;
        bool changed = false;
        u32 runT = 0xFFFFFFFFu, runC = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          u32 tgt = 0xFFFFFFFFu;
          if ((i + q < pend) && (stq[u][q] >> 30) == 0) {
            if (look[u][q]) { stq[u][q] = ((e[u][q] >> 30) == TR_K_EXP) ? ((e[u][q] & 0xFFFFu) | ((u32)L << 16)) : e[u][q]; changed = true; }
            if ((stq[u][q] >> 30) == 0) {
              const int k = (int)(stq[u][q] & 0xFFFFu) - nlo;
              if (k >= 0 && k < TR_NODECHUNK) tgt = (u32)k * 256 + (u32)((k48[u][q] >> (40 - 8 * L)) & 0xFFu);
            }
          }
          if (tgt == runT) runC++;
          else { if (runC && runT != 0xFFFFFFFFu) atomicAdd(&lds[runT], runC); runT = tgt; runC = 1; }
        }
        if (runC && runT != 0xFFFFFFFFu) atomicAdd(&lds[runT], runC);
        if (changed) *(uint4*)(state + i) = make_uint4(stq[u][0], stq[u][1], stq[u][2], stq[u][3]);
      }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < TR_NODECHUNK * 256; j += 1024) {
      const u32 v = lds[j];
      if (v && nlo + (j >> 8) < hi) atomicAdd(&cnt[(int64_t)(nlo + (j >> 8)) * 256 + (j & 255)], v);
    }
    __syncthreads();
  }
}

template <int TILE_, int MAXB_, bool FAST>
__device__ __forceinline__ void trk_scatter_body(const KeySrc& X, const u32* __restrict__ stateAll, u64* __restrict__ elemAll, const BwtArrays& A, const TrieArrays& T, int bitsG) {
  constexpr int ITEMS_ = TILE_ / 1024;
  const int b = bw_row_block(A, blockIdx.y);
  if ((T.meta[(int64_t)b * TR_META + 1] <= TRS_FASTB) != FAST) return;
  const int w0 = A.d_w[b];
  const int W = A.d_m[b] - w0;
  const int tbase = blockIdx.x * TILE_;
  if (tbase >= W) return;
  __shared__ u32 tc2[MAXB_ / 2];
  __shared__ u32 gdelta[MAXB_];
  __shared__ u64 stage[TILE_];
  __shared__ uint16_t stageB[TILE_];
  __shared__ u32 scan[32];
  const int32_t* meta = T.meta + (int64_t)b * TR_META;
  const int nB = meta[1];
  const bool hasState = meta[10 + 2] > meta[2 + 2];
  for (int i = threadIdx.x; i < (nB + 1) / 2; i += 1024) tc2[i] = 0;
  __syncthreads();
  const u64* key = X.key + (int64_t)b * A.NS + w0;
  const u32* val = X.val + (int64_t)b * A.NS + w0;
  const u32* state = stateAll + (int64_t)b * A.NS;
  const u32* info = T.info + (int64_t)b * T.MN * 256;
  u32* rank = A.rank + (int64_t)b * A.NS;
  const u64 lowMask = (1ULL << bitsG) - 1ULL;
  u64 el[ITEMS_]; u32 bp[ITEMS_];
  // loads in rounds, as in k_tr_scatter: keys / suffixes / states of all items, then all info lookups
  u64 k48v[ITEMS_]; u32 svv[ITEMS_], stv[ITEMS_];
#pragma unroll
  for (int r = 0; r < ITEMS_; r++) {
    const int i = tbase + r * 1024 + threadIdx.x;
    k48v[r] = 0; svv[r] = 0; stv[r] = 0xC0000000u;
    if (i < W) {
      k48v[r] = trk_k48(key[i], X.bitsR);
      svv[r] = val[i];
      stv[r] = hasState ? state[i] : ((1u << 16) | (u32)(k48v[r] >> 40));
    }
  }
#pragma unroll
  for (int r = 0; r < ITEMS_; r++) {
    const int i = tbase + r * 1024 + threadIdx.x;
    if (i < W && (stv[r] >> 30) == 0) stv[r] = info[(stv[r] & 0xFFFFu) * 256 + (u32)((k48v[r] >> (40 - 8 * (int)((stv[r] >> 16) & 7u))) & 0xFFu)];
  }
#pragma unroll
  for (int r = 0; r < ITEMS_; r++) {
    const int i = tbase + r * 1024 + threadIdx.x;
    bp[r] = 0xFFFFFFFFu;
    el[r] = 0;
    if (i < W) {
      const u64 k48 = k48v[r];
      const u32 sv = svv[r];
      const u32 st = stv[r];
      if ((st >> 30) == TR_K_TERM) {                                       // a new group of equal keys: its members' ranks, nothing to sort
        if (!(st & TR_UNCHANGED)) rank[sv] = (st & 0xFFFFFFu) | BW_LIVE;
      } else {
        const u32 bk = st & 0xFFFFu;
        const int skip = (int)((st >> 16) & 7u);
        el[r] = (((k48 << 16) << (8 * skip)) & ~lowMask) | (u64)sv;
        const u32 old = atomicAdd(&tc2[bk >> 1], (bk & 1u) ? 65536u : 1u);
        bp[r] = bk | (((bk & 1u) ? (old >> 16) : (old & 0xFFFFu)) << 16);
      }
    }
  }
  __syncthreads();
  {
    const int per = 2 * ((nB + 2047) / 2048);
    const int j0 = threadIdx.x * per;
    u32 mine = 0;
    for (int j = j0; j < j0 + per && j < nB; j += 2) { const u32 w = tc2[j >> 1]; mine += (w & 0xFFFFu) + (w >> 16); }
    u32 total;
    u32 run = kz_wg_excl_sum(mine, scan, &total);
    const u32* bStart = T.bStart + (int64_t)b * T.MB;
    u32* bFill = T.bFill + (int64_t)b * T.MB;
    for (int j = j0; j < j0 + per && j < nB; j += 2) {
      const u32 w = tc2[j >> 1];
      const u32 c0 = w & 0xFFFFu, c1 = w >> 16;
      if (c0) gdelta[j] = bStart[j] + atomicAdd(&bFill[j], c0) - run;
      const u32 r1 = run + c0;
      if (c1) gdelta[j + 1] = bStart[j + 1] + atomicAdd(&bFill[j + 1], c1) - r1;
      tc2[j >> 1] = run | (r1 << 16);
      run = r1 + c1;
    }
    if (threadIdx.x == 0) scan[31] = total;
  }
  __syncthreads();
  const u32 total = scan[31];
#pragma unroll
  for (int r = 0; r < ITEMS_; r++) {
    if (bp[r] != 0xFFFFFFFFu) {
      const u32 bk = bp[r] & 0xFFFFu;
      const u32 w = tc2[bk >> 1];
      const u32 slot = ((bk & 1u) ? (w >> 16) : (w & 0xFFFFu)) + (bp[r] >> 16);
      stage[slot] = el[r]; stageB[slot] = (uint16_t)bk;
    }
  }
  __syncthreads();
  u64* elem = elemAll + (int64_t)b * A.NS;
#pragma unroll
  for (int r = 0; r < ITEMS_; r++) {
    const u32 slot = (u32)r * 1024 + threadIdx.x;
    if (slot < total) elem[gdelta[stageB[slot]] + slot] = stage[slot];
  }
}

__global__ __launch_bounds__(1024) void k_trk_scatter(KeySrc X, const u32* __restrict__ stateAll, u64* __restrict__ elemAll, BwtArrays A, TrieArrays T, int bitsG) { trk_scatter_body<TRS_TILE, TR_MAXB, false>(X, stateAll, elemAll, A, T, bitsG); }
__global__ __launch_bounds__(1024) void k_trk_scatter_f(KeySrc X, const u32* __restrict__ stateAll, u64* __restrict__ elemAll, BwtArrays A, TrieArrays T, int bitsG) { trk_scatter_body<TRS_TILE / 2, TRS_FASTB, true>(X, stateAll, elemAll, A, T, bitsG); }

// ---------------------------------------------------------------------------------------------
// emit: header + BWT bytes (BWTBlockCodec.java:90-126, DivSufSort.java:217-224)
// XCD-aware mapping: the gather s[sa[j] - 1] hits random lines of the block's text, which fits ONE XCD's L2 (4 MiB) but is evicted
// when every XCD works on every block.  Workgroup w runs on XCD w % 8 (observed dispatch order, a speed matter only), so the
// workgroups of a super-group of 8 blocks are dealt out block = w % 8: each XCD gathers from one block's text at a time.
__global__ void k_bwt_emit(const u8* __restrict__ src, int64_t srcStride, u8* __restrict__ dst, int64_t dstStride,
                           BwtArrays A, int32_t* d_lenOut, int32_t* d_flag, int B) {
  const int lin = blockIdx.y * gridDim.x + blockIdx.x;
  const int sg = lin / (8 * (int)gridDim.x), r8 = lin % (8 * (int)gridDim.x);
  const int b = B < 0 ? (int)blockIdx.y : sg * 8 + (r8 & 7);               // B < 0: plain mapping (a grid of B rounded up to 8 rows would not fit)
  const int chunk = B < 0 ? (int)blockIdx.x : r8 >> 3;
  if (B >= 0 && b >= B) return;
  const int n = A.d_n[b];
  const u8* s = src + (int64_t)b * srcStride;
  u8* d = dst + (int64_t)b * dstStride;
  const int64_t off = (int64_t)b * A.NS;
  if (n < 2) {   // n==1: pIndexSize==0 -> BWTBlockCodec declines (BWTBlockCodec.java:95-98); n==0 no-op
    if (chunk == 0 && threadIdx.x == 0) { d_flag[b] = 0; d_lenOut[b] = n; if (n == 1) d[0] = s[0]; }
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
  if (chunk == 0 && threadIdx.x == 0) {
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
  for (int j = chunk * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    if (j == p) continue;
    const u32 sv = sa[j];
    const u8 c = s[sv - 1];
    d[hdr + ((j < p) ? j + 1 : j)] = c;
  }
}

// ---------------------------------------------------------------------------------------------
// The trie round is used for blocks of 64 KiB .. 4 MiB (+ slack).  Table bounds (n = block length, C = TR_CAP):
//   nodes: 256 of depth 1 + at most n / (C + 1) expanded children per level, levels 2 .. TR_DMAX_LIMIT - 1;
//   buckets: children above TR_MERGE (<= n / (TR_MERGE + 1)) + merged ones: two neighbours of a run hold more than C suffixes
//   together (<= 2n / C + 1 per run) and a run starts at a node or behind a breaker (nodes + children above TR_MERGE + terminals).
#define TR_DMAX_LIMIT 7
#define TR_MIN_N 65536
#define TR_MAX_N ((4 << 20) + 65536)
static inline int tr_max_nodes(int maxN) { return 257 + (TR_DMAX_LIMIT - 2) * (maxN / (int)(TR_CAP + 1) + 1); }
static inline int tr_max_buckets(int maxN) {
  const int med = maxN / (int)(TR_MERGE + 1) + 1, big = maxN / (int)(TR_CAP + 1) + 1;
  return med + 2 * (maxN / (int)TR_CAP) + 2 + tr_max_nodes(maxN) + med + (TR_DMAX_LIMIT - 2) * big + big;
}
static inline bool tr_applies(int maxN) {                          // (KZ_BWT_TRIE=0 switches the trie rounds off at the call: ctx->sw.bwtTrie)
  return maxN >= TR_MIN_N && maxN <= TR_MAX_N && tr_max_buckets(maxN) <= TR_MAXB && tr_max_nodes(maxN) <= 65535;
}

size_t kz_bwt_forward_scratch(int B, int maxN) {
  const int64_t NS = (int64_t)kz_align((size_t)maxN, RS_TILE);
  const int T = (int)(NS / RS_TILE);
  size_t per = (size_t)NS * (8 * 2 + 4 * 2 + 4 + 4) + (size_t)(T + 4) * (256 * 4 + 12) + 2 * MSD_BINS * 4 + 64 + 11 * 256;
  if (tr_applies(maxN)) per += (size_t)tr_max_nodes(maxN) * (2 * 1024 + 12) + (size_t)TR_MAXB * 20 + TR_META * 4 + 9 * 256;
  return kz_align(per * (size_t)B + 4096 * 16, 4096) + (1 << 20);
}

static inline int gridFor(int n, int per) { int g = (n + per - 1) / per; return g < 1 ? 1 : g; }

// one pass of the stage; *trieOverflow is set (and nothing of the batch changed) when a trie table overflowed in any round
static int bwt_forward_run(kz_ctx* ctx, kz_batch& bt, bool allowTrie, bool* trieOverflow) {
  const int B = bt.B;
  int maxN = 0;
  for (int b = 0; b < B; b++) if (bt.h_len[b] > maxN) maxN = bt.h_len[b];
  const int64_t NS = (int64_t)kz_align((size_t)(maxN > 0 ? maxN : 1), RS_TILE);
  const int T = (int)(NS / RS_TILE);
  BwtArrays A;
  A.NS = NS; A.T = T; A.HS = (int64_t)((NS + RSORT_TILE - 1) / RSORT_TILE) * MSD_BINS;
  for (int i = 0; i < 2; i++) {
    A.key[i] = (u64*)kz_arena_alloc(ctx, (size_t)NS * B * 8);
    A.val[i] = (u32*)kz_arena_alloc(ctx, (size_t)NS * B * 4);
  }
  A.rank = (u32*)kz_arena_alloc(ctx, (size_t)NS * B * 4);
  A.sa = (u32*)kz_arena_alloc(ctx, (size_t)NS * B * 4);
  A.tileHist = (u32*)kz_arena_alloc(ctx, (size_t)A.HS * B * 4);
  A.digitBase = (u32*)kz_arena_alloc(ctx, (size_t)B * MSD_BINS * 4);
  A.bucketCnt = (u32*)kz_arena_alloc(ctx, (size_t)B * MSD_BINS * 4);
  A.d_w = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  A.d_big = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  A.tileA = (u32*)kz_arena_alloc(ctx, (size_t)T * B * 4);
  A.tileB = (u32*)kz_arena_alloc(ctx, (size_t)T * B * 4);
  A.tileLive = (u32*)kz_arena_alloc(ctx, (size_t)T * B * 4);
  A.d_m = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  A.d_m2 = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  int32_t* d_act = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  A.act = nullptr;
  if (!d_act || !A.d_m2 || !A.tileB || !A.tileLive || !A.sa || !A.d_big || !A.bucketCnt) { snprintf(ctx->err, sizeof(ctx->err), "bwt_forward: arena overflow"); return -KZ_ERR_DEVICE; }
  bool useTrie = allowTrie && ctx->sw.bwtTrie != 0 && tr_applies(maxN);
  TrieArrays TR;
  memset(&TR, 0, sizeof(TR));
  const size_t markTrie = ctx->arenaTop;
  if (useTrie) {
    TR.MN = tr_max_nodes(maxN); TR.MB = TR_MAXB;
    TR.cnt = (u32*)kz_arena_alloc(ctx, (size_t)B * TR.MN * 1024);
    TR.info = (u32*)kz_arena_alloc(ctx, (size_t)B * TR.MN * 1024);
    TR.nodeStart = (u32*)kz_arena_alloc(ctx, (size_t)B * TR.MN * 4);
    TR.bStart = (u32*)kz_arena_alloc(ctx, (size_t)B * TR.MB * 4);
    TR.bCount = (u32*)kz_arena_alloc(ctx, (size_t)B * TR.MB * 4);
    TR.bFill = (u32*)kz_arena_alloc(ctx, (size_t)B * TR.MB * 4);
    TR.bAux = (u32*)kz_arena_alloc(ctx, (size_t)B * TR.MB * 4);
    TR.bG = (u32*)kz_arena_alloc(ctx, (size_t)B * TR.MB * 4);
    TR.nodePfx = (u32*)kz_arena_alloc(ctx, (size_t)B * TR.MN * 4);
    TR.nodeGS = (u32*)kz_arena_alloc(ctx, (size_t)B * TR.MN * 4);
    TR.meta = (int32_t*)kz_arena_alloc(ctx, (size_t)B * TR_META * 4);
    TR.err = (int32_t*)kz_arena_alloc(ctx, 256);
    if (!TR.cnt || !TR.info || !TR.nodeStart || !TR.bStart || !TR.bCount || !TR.bFill || !TR.bAux || !TR.bG || !TR.nodePfx || !TR.nodeGS || !TR.meta || !TR.err) {
      // the caller sized the arena for the longest block the CHAIN can hand over, which may lie outside the trie rounds' range while
      // this batch's blocks are inside: the LSD rounds need no tables
      ctx->arenaTop = markTrie;
      useTrie = false;
    }
  }
  A.d_n = bt.d_len;
  hipStream_t st = ctx->stream;
  const u8* src = bt.buf[bt.cur];
  u8* dst = bt.buf[bt.cur ^ 1];

  int bitsR = 1;
  while ((1LL << bitsR) < 2 * (int64_t)maxN + 2) bitsR++;         // secondary key < n + h + 1 <= 2n + 1
  int bitsG = 1;
  while ((1LL << bitsG) < (int64_t)maxN) bitsG++;

  const int K0 = 7;                                                // bytes of the first-round key
  u64 *kC = A.key[0], *kF = A.key[1];
  u32 *vC = A.val[0], *vF = A.val[1];
  KZ_LAUNCH(ctx, KID_BWT_INIT, k_bwt_init, dim3((B + 255) / 256), dim3(256), A, B);
  const TextSrc X0 = {src, bt.stride, K0};
  int mMax = maxN;
  int h = 0;
  // bucket path of the later rounds: the packed LDS element needs BK_BITS + bitsR + bitsG <= 64
  const bool useBuckets = (BK_BITS + bitsR + bitsG <= 64) && (1 << (bitsG > BK_BITS ? bitsG - BK_BITS : 0)) <= MSD_BINS && ctx->sw.bwtBuckets != 0;
  const int nBuckets = 1 << (bitsG > BK_BITS ? bitsG - BK_BITS : 0);
  const int bucketMin = 4096;                                      // below: the whole list is a handful of LSD tiles
  bool windowed = false;
  const u32 gmax = (u32)BK_GMAX;
  const int tilesN = gridFor(maxN, RS_TILE);
  // the context's switches (kz_switches: read when the context was created)
  const int Dmax = (ctx->sw.bwtDmax >= 6 && ctx->sw.bwtDmax <= TR_DMAX_LIMIT) ? ctx->sw.bwtDmax : 6;
  const bool noRetire = ctx->sw.bwtRetire == 0;                      // A/B switch: every block rides through every round, as before
  const bool trieWinOff = ctx->sw.bwtTrieWin == 0;
  const bool lazyRank = ctx->sw.bwtLazyRank != 0;
  const bool trace = ctx->sw.bwtTrace != 0;
  const int forceOvfRound = ctx->sw.bwtTestTrieOverflow;             // tests: pretend the tables overflowed in round <digit>
  if (useTrie) KZ_HIP(hipMemsetAsync(TR.err, 0, 4, st));
  int nAct = B;                                                      // grid rows of the later rounds: blocks with live suffixes (A.act)
  int32_t* const h_act = ctx->hpin + B + 16;                         // their indices, host side (pipe_setup reserves 9 B + 64 ints)
  for (int round = 0; round < 64 && mMax > 0; round++) {
    // ---- sort (kC,vC) and apply.  Later rounds of blocks up to 4 MiB: bucket partition + one LDS sort per bucket, the
    //      LSD passes only for the window of oversized buckets; otherwise LSD radix over the whole compact list ----
    const int nbits = (round == 0) ? 8 * K0 : bitsR + bitsG;
    const int passes = (nbits + 7) / 8;
    const int gshift = (round == 0) ? 64 : bitsR;
    const bool buckets = useBuckets && round > 0 && mMax >= bucketMin;
    int wMax = mMax;                                               // largest LSD window of the batch
    if (round == 0 && useTrie) {
      // ---- the trie round: count level by level, move once, finish the buckets in LDS (elements in key[0], states in val[0]) ----
      KZ_HIP(hipMemsetAsync(TR.bFill, 0, (size_t)B * TR.MB * 4, st));
      KZ_HIP(hipMemsetAsync(TR.meta, 0, (size_t)B * TR_META * 4, st));
      KZ_LAUNCH(ctx, KID_TR_HIST16, k_tr_hist16, dim3(2, B), dim3(1024), src, bt.stride, A, TR);
      KZ_LAUNCH(ctx, KID_TR_ASSIGN, k_tr_assign, dim3(B), dim3(256), A, TR, 1, Dmax, 0);
      const int P = B >= 2048 ? 1 : (B >= 1024 ? 2 : 8);
      for (int L = 2; L < Dmax; L++) {
        KZ_LAUNCH(ctx, KID_TR_COUNT, k_tr_count, dim3(P, B), dim3(1024), src, bt.stride, A.val[0], A, TR, L);
        KZ_LAUNCH(ctx, KID_TR_ASSIGN, k_tr_assign, dim3(B), dim3(256), A, TR, L, Dmax, 0);
      }
      KZ_LAUNCH(ctx, KID_TR_SCATTER, k_tr_scatter_f, dim3(gridFor(maxN, TRS_TILE / 2), B), dim3(1024), src, bt.stride, A.val[0], A.key[0], A, TR, bitsG);
      KZ_LAUNCH(ctx, KID_TR_SCATTER, k_tr_scatter, dim3(gridFor(maxN, TRS_TILE), B), dim3(1024), src, bt.stride, A.val[0], A.key[0], A, TR, bitsG);
      const int G = std::max(16, std::min(1024, 8192 / B));
      KZ_LAUNCH(ctx, KID_TR_SORT, k_tr_sort, dim3(G, B), dim3(1024), A.key[0], A, TR, bitsG, lazyRank ? 1 : 0);
      if (lazyRank) KZ_LAUNCH(ctx, KID_TR_SORT, k_tr_sort, dim3(G, B), dim3(1024), A.key[0], A, TR, bitsG, 2);   // the blocks pass 1 guessed wrong (workgroups of the others leave at once)
      wMax = 0;
    } else
    if (buckets) {
      const int rt = gridFor(mMax, RSORT_TILE);
      const int sh = bitsR + BK_BITS;
      KZ_LAUNCH(ctx, KID_MSD_HIST, k_msd_hist, dim3(rt, nAct), dim3(KZ_WG), kC, A, sh);
      KZ_LAUNCH(ctx, KID_MSD_SCAN, k_msd_scan, dim3(nAct), dim3(MSD_BINS), A);
      KZ_HIP(hipMemcpyAsync(ctx->hpin, A.d_big, (size_t)B * 4, hipMemcpyDeviceToHost, st));
      KZ_LAUNCH(ctx, KID_MSD_SCATTER, k_msd_scatter, dim3(rt, nAct), dim3(RSC_WG), kC, vC, kF, vF, A, sh);
      { u64* tk = kC; kC = kF; kF = tk; u32* tv = vC; vC = vF; vF = tv; }
      KZ_LAUNCH(ctx, KID_BUCKET_COUNT_S, k_bucket_count_s, dim3(nBuckets, nAct), dim3(256), kC, vC, A, bitsR, bitsG, gmax);
      KZ_LAUNCH(ctx, KID_BUCKET_COUNT, k_bucket_count, dim3(nBuckets, nAct), dim3(1024), kC, vC, A, bitsR, bitsG, gmax);
      KZ_LAUNCH(ctx, KID_BUCKET_SORT, k_bucket_sort, dim3(nBuckets, nAct), dim3(1024), kC, vC, A, bitsR, bitsG);
      KZ_HIP(kz_stream_sync(ctx, st));
      wMax = 0;                                                      // (a retired block's d_big is stale: only the live rows count)
      for (int r = 0; r < nAct; r++) { const int b = A.act ? h_act[r] : r; if (ctx->hpin[b] > wMax) wMax = ctx->hpin[b]; }
      windowed = true;
      if (trace) {
        long long tot = 0; for (int r = 0; r < nAct; r++) tot += ctx->hpin[A.act ? h_act[r] : r];
        fprintf(stderr, "[bwt] round %d: %lld suffixes in oversized buckets, max per block %d\n", round, tot, wMax);
      }
    } else if (windowed) {
      KZ_HIP(hipMemsetAsync(A.d_w, 0, (size_t)B * 4, st));
      windowed = false;
    }
    const bool trieWindow = useTrie && buckets && wMax > 0 && bitsR <= 24 && bitsG <= 24 && !trieWinOff;
    if (trieWindow) {
      // ---- the window of oversized buckets through a KEY trie round (k_trk_*): count by key byte, move once, finish in LDS ----
      const KeySrc XK = {kC, vC, bitsR};
      KZ_HIP(hipMemsetAsync(TR.bFill, 0, (size_t)B * TR.MB * 4, st));
      KZ_HIP(hipMemsetAsync(TR.meta, 0, (size_t)B * TR_META * 4, st));
      KZ_LAUNCH(ctx, KID_TR_HIST16, k_trk_hist16, dim3(2, nAct), dim3(1024), XK, A, TR);
      KZ_LAUNCH(ctx, KID_TR_ASSIGN, k_tr_assign, dim3(nAct), dim3(256), A, TR, 1, 6, 1);
      const int PW = nAct >= 1024 ? 1 : (nAct >= 256 ? 2 : 4);
      for (int L = 2; L < 6; L++) {
        KZ_LAUNCH(ctx, KID_TR_COUNT, k_trk_count, dim3(PW, nAct), dim3(1024), XK, vF, A, TR, L);
        KZ_LAUNCH(ctx, KID_TR_ASSIGN, k_tr_assign, dim3(nAct), dim3(256), A, TR, L, 6, 1);
      }
      KZ_LAUNCH(ctx, KID_TR_SCATTER, k_trk_scatter_f, dim3(gridFor(wMax, TRS_TILE / 2), nAct), dim3(1024), XK, vF, kF, A, TR, bitsG);
      KZ_LAUNCH(ctx, KID_TR_SCATTER, k_trk_scatter, dim3(gridFor(wMax, TRS_TILE), nAct), dim3(1024), XK, vF, kF, A, TR, bitsG);
      const int G = std::max(16, std::min(1024, 8192 / nAct));
      KZ_LAUNCH(ctx, KID_TR_SORT, k_trk_sort, dim3(G, nAct), dim3(1024), kF, A, TR, bitsG);
    } else
    if (wMax > 0) {
      const int tiles = gridFor(wMax, RS_TILE);
      const int rtiles = gridFor(wMax, RSORT_TILE);
      for (int p = 0; p < passes; p++) {
        if (round == 0 && p == 0) {                                  // keys straight from the text
          KZ_LAUNCH(ctx, KID_RADIX_HIST, k_radix_hist0, dim3(rtiles, nAct), dim3(KZ_WG), A, X0);
          KZ_LAUNCH(ctx, KID_RADIX_SCAN, k_radix_scan, dim3(nAct), dim3(256), A);
          KZ_LAUNCH(ctx, KID_RADIX_SCATTER, k_radix_scatter0, dim3(rtiles, nAct), dim3(RSC_WG), kF, vF, A, X0);
        } else {
        KZ_LAUNCH(ctx, KID_RADIX_HIST, k_radix_hist, dim3(rtiles, nAct), dim3(KZ_WG), kC, A, p * 8);
        KZ_LAUNCH(ctx, KID_RADIX_SCAN, k_radix_scan, dim3(nAct), dim3(256), A);
        KZ_LAUNCH(ctx, KID_RADIX_SCATTER, k_radix_scatter, dim3(rtiles, nAct), dim3(RSC_WG), kC, vC, kF, vF, A, p * 8);
        }
        u64* tk = kC; kC = kF; kF = tk;
        u32* tv = vC; vC = vF; vF = tv;
      }
      // ---- SA order: new groups, ranks, final suffixes ----
      KZ_LAUNCH(ctx, KID_SEG_REDUCE, k_seg_reduce, dim3(tiles, nAct), dim3(KZ_WG), kC, A, gshift);
      KZ_LAUNCH(ctx, KID_SEG_SCAN, k_seg_scan, dim3(nAct), dim3(64), A);
      KZ_LAUNCH(ctx, KID_SEG_APPLY, k_seg_apply, dim3(tiles, nAct), dim3(KZ_WG), kC, vC, A, gshift);
    }
    // ---- text order: compact the live suffixes, keys for the next round ----
    h = (round == 0) ? (useTrie ? 6 : K0) : h * 2;
    KZ_HIP(hipMemsetAsync(A.d_m2, 0, (size_t)B * 4, st));
    KZ_LAUNCH(ctx, KID_LIVE_EMIT, k_live_emit, dim3(tilesN, nAct), dim3(KZ_WG), kF, vF, A, h, bitsR, round == 0 ? 1 : 0, (round == 0 && useTrie && lazyRank) ? TR.meta : (const int32_t*)nullptr);
    { u64* tk = kC; kC = kF; kF = tk; u32* tv = vC; vC = vF; vF = tv; }
    // ---- read back the next compact sizes ----
    KZ_HIP(hipMemcpyAsync(ctx->hpin, A.d_m2, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    // the trie tables' overflow flag rides along, every round that ran a trie (round 0 and the key rounds of the window): an
    // overflow dropped buckets, the ranks of this pass are void (the bounds of tr_max_nodes / tr_max_buckets say it cannot happen)
    ctx->hpin[B] = 0;
    if (useTrie) KZ_HIP(hipMemcpyAsync(ctx->hpin + B, TR.err, 4, hipMemcpyDeviceToHost, st));
    KZ_HIP(kz_stream_sync(ctx, st));
    mMax = 0;
    for (int b = 0; b < B; b++) if (ctx->hpin[b] > mMax) mMax = ctx->hpin[b];
    if (useTrie && (ctx->hpin[B] != 0 || round == forceOvfRound)) { *trieOverflow = true; return 0; }
    if (trace) {                                   // diagnostic: live suffixes left after every doubling round
      long long tot = 0, totN = 0; for (int b = 0; b < B; b++) { tot += ctx->hpin[b]; totN += bt.h_len[b]; }
      fprintf(stderr, "[bwt] round %d h=%d: live %lld of %lld (%.1f%%), max per block %d\n", round, h, tot, totN, 100.0 * tot / (totN ? totN : 1), mMax);
    }
    int32_t* tm = A.d_m; A.d_m = A.d_m2; A.d_m2 = tm;
    // ---- retire the blocks that are done: the next round's grids get one row per block that still has live suffixes ----
    if (mMax > 0 && !noRetire) {
      int cnt = 0;
      for (int b = 0; b < B; b++) if (ctx->hpin[b] > 0) h_act[cnt++] = b;     // (hpin[0 .. B) is read before the next copy lands: in-order stream)
      if (cnt < B) {
        KZ_HIP(hipMemcpyAsync(d_act, h_act, (size_t)cnt * 4, hipMemcpyHostToDevice, st));
        A.act = d_act;
        nAct = cnt;
      }
    }
  }
  if (mMax > 0) { snprintf(ctx->err, sizeof(ctx->err), "bwt_forward: suffix sort did not converge"); return -KZ_ERR_PROCESS_BLOCK; }
  {
    const int rows8 = (B + 7) / 8 * 8;
    const bool xcd = rows8 <= KZ_MAX_BATCH;
    KZ_LAUNCH(ctx, KID_BWT_EMIT, k_bwt_emit, dim3(gridFor(maxN, 256 * 8), xcd ? rows8 : B), dim3(256), src, bt.stride, dst, bt.stride, A, bt.d_len2, bt.d_flag, xcd ? B : -1);
  }
  KZ_HIP(hipGetLastError());
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}

int kz_stage_bwt_forward(kz_ctx* ctx, kz_batch& bt) {
  const size_t mark = ctx->arenaTop;
  bool overflow = false;
  int rc = bwt_forward_run(ctx, bt, true, &overflow);
  if (rc == 0 && overflow) {                                       // redo the batch on the LSD rounds (no tables to overflow); the input buffer is untouched
    ctx->arenaTop = mark;
    overflow = false;
    rc = bwt_forward_run(ctx, bt, false, &overflow);
  }
  return rc;
}
