// kz_srt.hip -- Sorted Rank Transform (level-6 chain) on gfx950.
//
// Replaces K/transform/SRT.java:66-168 (forward), :171-257 (inverse), :259-302 (preprocess: symbols by
// (frequency desc, symbol asc) -- a total order, so any sort gives the reference's shell-sort result),
// :304-346 (header = 256 LEB128-style frequencies).
//
// forward = three data-parallel pieces: (1) byte histogram; (2) move-to-front ranks -- SRT's list starts
//   as the order of first appearance, so rank(c) = #symbols whose last occurrence is later than c's
//   (never seen: number of symbols seen so far); zero inside a run falls out naturally.  This is the
//   SBRT tile machinery with key 0 for never-seen symbols (kz_sbrt.hip mode 4);  (3) a STABLE scatter of
//   the ranks into per-symbol buckets laid out in (freq desc, symbol asc) order: LDS tile histograms,
//   per-symbol scan over tiles, ballot match-any ranking.
// inverse: the list is ordered by NEXT occurrence and every step needs the next rank of the symbol just
//   emitted (a dependent load), so it is serial per block: one wave per block, list in one VGPR,
//   64-byte windows of the current symbol's bucket fetched per event, zero ranks (runs) consumed in O(1).
#include "kz_device.h"
#include "kz_internal.h"

typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint8_t u8;

#define SR_ITEMS 16
#define SR_TILE (KZ_WG * SR_ITEMS)

struct SrtFwd {
  u32* tileHist;    // [B][T][256]
  u32* bucket;      // [B][256] bucket start (in output order) per symbol
  int32_t* hdrLen;  // [B]
  u8* ranks;        // [B][stride] MTF ranks
  int T;
};

__global__ __launch_bounds__(KZ_WG) void k_srt_hist(const u8* __restrict__ src, int64_t stride, const int32_t* __restrict__ d_len, SrtFwd S) {
  const int b = blockIdx.y, tile = blockIdx.x;
  const int n = d_len[b];
  if ((int64_t)tile * SR_TILE >= n) return;
  __shared__ u32 hist[256];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const u8* s = src + (int64_t)b * stride;
  const int base = tile * SR_TILE;
#pragma unroll 4
  for (int r = 0; r < SR_ITEMS; r++) {
    const int idx = base + r * KZ_WG + threadIdx.x;
    const bool valid = idx < n;
    const u32 d = valid ? s[idx] : 0;
    const uint64_t peers = kz_match8(d, valid);
    if (valid && (peers & kz_lanemask_lt()) == 0) atomicAdd(&hist[d], (u32)__popcll(peers));
  }
  __syncthreads();
  S.tileHist[((int64_t)b * S.T + tile) * 256 + threadIdx.x] = hist[threadIdx.x];
}

// per block: per-symbol tile offsets, frequencies, bucket order, header (thread = symbol)
__global__ __launch_bounds__(256) void k_srt_prep(const int32_t* __restrict__ d_len, SrtFwd S, u8* __restrict__ dst, int64_t stride) {
  const int b = blockIdx.x;
  const int n = d_len[b];
  if (n <= 0) { if (threadIdx.x == 0) S.hdrLen[b] = 0; return; }
  const int tiles = (n + SR_TILE - 1) / SR_TILE;
  __shared__ u32 freq[256];
  __shared__ u32 start[256];
  __shared__ u8 order[256];
  u32* h = S.tileHist + (int64_t)b * S.T * 256;
  const int sym = threadIdx.x;
  u32 run = 0;
  for (int t = 0; t < tiles; t++) { const u32 v = h[(int64_t)t * 256 + sym]; h[(int64_t)t * 256 + sym] = run; run += v; }
  freq[sym] = run;
  __syncthreads();
  // position in (freq desc, symbol asc) order (SRT.java:259-302)
  u32 pos = 0;
  for (int t = 0; t < 256; t++) { const u32 ft = freq[t]; if (ft > run || (ft == run && t < sym)) pos++; }
  order[pos] = (u8)sym;
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 acc = 0;
    for (int i = 0; i < 256; i++) { const int c = order[i]; start[c] = acc; acc += freq[c]; }
    // header (SRT.java:304-319)
    u8* d = dst + (int64_t)b * stride;
    int hl = 0;
    for (int i = 0; i < 256; i++) { u32 f = freq[i]; while (f >= 128) { d[hl++] = (u8)(0x80 | f); f >>= 7; } d[hl++] = (u8)f; }
    S.hdrLen[b] = hl;
  }
  __syncthreads();
  S.bucket[b * 256 + sym] = start[sym];
}

// stable scatter of rank bytes into the symbol buckets
__global__ __launch_bounds__(KZ_WG) void k_srt_scatter(const u8* __restrict__ src, u8* __restrict__ dst, int64_t stride,
                                                        const int32_t* __restrict__ d_len, SrtFwd S) {
  const int b = blockIdx.y, tile = blockIdx.x;
  const int n = d_len[b];
  if ((int64_t)tile * SR_TILE >= n) return;
  __shared__ u32 cnt[4][256];
  for (int i = threadIdx.x; i < 1024; i += KZ_WG) (&cnt[0][0])[i] = 0;
  __syncthreads();
  const u8* s = src + (int64_t)b * stride;
  const u8* rk = S.ranks + (int64_t)b * stride;
  u8* d = dst + (int64_t)b * stride + S.hdrLen[b];
  const int wave = threadIdx.x >> 6, lane = kz_lane();
  const int base = tile * SR_TILE + wave * (64 * SR_ITEMS);
  const uint64_t lt = kz_lanemask_lt();
  u32 dr[SR_ITEMS];
#pragma unroll
  for (int r = 0; r < SR_ITEMS; r++) {
    const int idx = base + r * 64 + lane;
    const bool valid = idx < n;
    const u32 c = valid ? s[idx] : 0;
    const uint64_t peers = kz_match8(c, valid);
    u32 pre = 0;
    if (valid) pre = cnt[wave][c];
    const u32 rnk = pre + (u32)__popcll(peers & lt);
    if (valid && (peers >> lane) == 1ULL) cnt[wave][c] = pre + (u32)__popcll(peers);
    dr[r] = c | (rnk << 8);
  }
  __syncthreads();
  {
    const int c = threadIdx.x;
    u32 runv = S.bucket[b * 256 + c] + S.tileHist[((int64_t)b * S.T + tile) * 256 + c];
#pragma unroll
    for (int w = 0; w < 4; w++) { const u32 t = cnt[w][c]; cnt[w][c] = runv; runv += t; }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < SR_ITEMS; r++) {
    const int i = base + r * 64 + lane;
    if (i < n) d[cnt[wave][dr[r] & 0xFF] + (dr[r] >> 8)] = rk[i];
  }
}

__global__ void k_srt_ffin(const int32_t* __restrict__ d_len, int32_t* __restrict__ d_len2, int32_t* __restrict__ d_flag, SrtFwd S, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int n = d_len[b];
  d_len2[b] = (n > 0) ? n + S.hdrLen[b] : 0;
  d_flag[b] = 1;
}

// =================================================================================================
// inverse: one wave per block
#define KZ_DPP_SHL1(x) ((u32)__builtin_amdgcn_update_dpp(0, (int)(x), 0x130 /*wave_shl:1*/, 0xF, 0xF, true))   /* lane 63 gets 0 */

// per-wave working set of k_srt_inv (up to 8 waves = 8 blocks per workgroup, see kz_place_blocks)
struct SrtInvLds {
  union { int32_t freq[256]; int32_t wbase[256]; };   // freq: header and bucket ranges only; wbase: the general loop's window bases
  int32_t bstart[256]; int32_t bend[256];
  u8 order[256]; u8 r2s0[256];
  union {
    u8 win[256][32];                 // general loop: next 32 ranks of every symbol
    u8 ring[256][64];                // fast loop: slot t & 63 holds rank t of the symbol's bucket, valid from the cursor to the next multiple of 64
  };
  int h, bad;
};
#define SRT_SYNC() __builtin_amdgcn_wave_barrier()   /* one wave per block: program order suffices, keep the compiler in line */

#define SRT_INV_LOOP(GENERAL) \
  while (i < count) { \
    const int32_t cur = bstart[c], end = bend[c]; \
    const int32_t lim = GENERAL ? min(end, count) : end; \
    int32_t wb = wbase[c]; \
    int vc = min(lim, wb + 32) - cur; /* ranks of c available in the cached window */ \
    if (vc <= 0 && cur < lim) { /* window used up: fetch the next 32 ranks */ \
      wb = cur; \
      if (lane < 32) win[c][lane] = (cur + lane < lim) ? s[cur + lane] : (u8)0; \
      if (lane == 0) wbase[c] = wb; \
      vc = min(32, lim - cur); \
    } \
    if (GENERAL) { if (vc < 0) vc = 0; if (vc == 0 && cur < end) { bad = 1; break; } }   /* the next rank of c lies past the payload */ \
    const u32 v = (lane < vc) ? (u32)win[c][((u32)(cur - wb) + (u32)lane) & 31u] : 0u; \
    const uint64_t nz = kz_ballot(v != 0 && lane < vc); \
    const int z = nz ? (int)__builtin_ctzll(nz) : vc; /* leading zero ranks = c repeats */ \
    int r = 0; \
    int emit, consumed; \
    bool moveC = false, removeC = false; \
    if (nz) { emit = z + 1; consumed = z + 1; r = __builtin_amdgcn_readlane((int)v, z); moveC = true; } \
    else if (cur + vc >= end) { emit = vc + 1; consumed = vc; removeC = true; } /* bucket exhausted (:239-248) */ \
    else { emit = vc; consumed = vc; } /* only zeros in the window, more to come */ \
    if (emit > count - i) { emit = count - i; moveC = false; removeC = false; } \
    if (lane < emit) o[i + lane] = (u8)c; \
    i += emit; \
    if (lane == 0) bstart[c] = cur + consumed; \
    if (moveC || (removeC && nbSymbols > 1)) { \
 /* moveC:   positions 0..r-1 <- 1..r, position r <- c                        (SRT.java:233-237) */ \
 /* removeC: positions 0..nbSymbols-1 <- 1..nbSymbols, the rest stays          (:242-248) */ \
      if (removeC) { nbSymbols--; r = nbSymbols; } \
      const u32 nxt = KZ_DPP_SHL1(list); \
      const u32 shifted = (list >> 8) | (nxt << 24); \
      const int jb = 4 * lane; \
      const int e = r - jb; /* bytes with position < r in this lane */ \
      const u32 mask = (e <= 0) ? 0u : ((e >= 4) ? 0xFFFFFFFFu : (0xFFFFFFFFu >> (8 * (4 - e)))); \
      u32 nl = (shifted & mask) | (list & ~mask); \
      if (moveC && lane == (r >> 2)) { const int sh = 8 * (r & 3); nl = (nl & ~(0xFFu << sh)) | ((u32)c << sh); } \
      list = nl; \
      c = (int)(__builtin_amdgcn_readfirstlane((int)list) & 0xFF); \
    } else if (removeC) { \
 /* single symbol left with an empty bucket: it repeats to the end (:239-240) */ \
      for (int k = i + lane; k < count; k += 64) o[k] = (u8)c; \
      i = count; \
    } \
  }

// One wave per block replays SRT.java:178-257 step by step.  It follows the reference on ANY input, not only on
// well-formed ones: frequencies are Java ints (a 5-byte varint can wrap negative; only freq > 0 counts, :268-273), the
// three tables start zeroed like the fields of a fresh SRT (absent symbols have an empty bucket at 0), a removal shifts
// only the first nbSymbols entries of r2s (:245-246), and a bucket running past the payload fails at the step that
// would read there (the oracle's reading of the out-of-range access).
__global__ __launch_bounds__(512) void k_srt_inv(const u8* __restrict__ src, u8* __restrict__ dst, int64_t stride,
                                                  const int32_t* __restrict__ d_len, int32_t* __restrict__ d_len2, int32_t* __restrict__ d_flag,
                                                  const int32_t* __restrict__ orderIdx, int wavesPerGroup) {
  __shared__ SrtInvLds LDS[8];
  const int wv = (int)(threadIdx.x >> 6);
  const int b = __builtin_amdgcn_readfirstlane(orderIdx[blockIdx.x * wavesPerGroup + __builtin_amdgcn_readfirstlane(wv)]);   // uniform on purpose: scalar loads and scalar control flow
  if (b < 0) return;
  SrtInvLds& L = LDS[wv];
  int32_t* freq = L.freq; int32_t* bstart = L.bstart; int32_t* bend = L.bend; int32_t* wbase = L.wbase;
  u8* order = L.order; u8* r2s0 = L.r2s0;
  u8 (*win)[32] = L.win;
  const int length = d_len[b];
  const int lane = kz_lane();
  const u8* in = src + (int64_t)b * stride;
  u8* o = dst + (int64_t)b * stride;
  if (length <= 0) { if (lane == 0) { d_len2[b] = 0; d_flag[b] = 1; } return; }
  // ---- header (SRT.java:321-346) ----
  if (lane == 0) {
    int h = 0, bad = 0;
    for (int i = 0; i < 256 && !bad; i++) {
      if (h >= length) { bad = 1; break; }
      u32 val = in[h++];
      u32 res = val & 0x7F; int shift = 7;
      while (val >= 128) {
        if (h >= length) { bad = 1; break; }
        val = in[h++];
        res |= ((val & 0x7F) << shift);                               // wraps like the Java int
        if (shift > 21) break;
        shift += 7;
      }
      freq[i] = (int32_t)res;
    }
    L.h = h; L.bad = bad;
  }
  for (int i = lane; i < 256; i += 64) { r2s0[i] = 0; bstart[i] = 0; bend[i] = 0; }
  SRT_SYNC();
  const int H = __builtin_amdgcn_readfirstlane(L.h);               // (LDS values the whole wave agrees on: scalar from here)
  const int count = length - H;
  if (__builtin_amdgcn_readfirstlane(L.bad) || count < 0) { if (lane == 0) { d_len2[b] = 0; d_flag[b] = 0; } return; }
  const u8* s = in + H;
  // ---- bucket order of the present symbols (freq desc, symbol asc :259-302), bucket ranges ----
  int nbSymbols = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int sym = q * 64 + lane;
    const int32_t f = freq[sym];
    u32 pos = 0;
    for (int t = 0; t < 256; t++) { const int32_t ft = freq[t]; if (ft > 0 && (ft > f || (ft == f && t < sym))) pos++; }
    if (f > 0) order[pos] = (u8)sym;
    nbSymbols += (int)__popcll(kz_ballot(f > 0));
  }
  SRT_SYNC();
  int bad = 0;
  bool general = false;
  {
    int32_t acc = 0;
    for (int i = 0; i < nbSymbols; i++) {                           // :204-215 (uniform across lanes)
      const int c = order[i];
      const int32_t at = (int32_t)((u32)H + (u32)acc);               // srcIdx + bucketPos, Java int arithmetic
      if (at < 0 || at >= length) { bad = 1; break; }
      const int first = in[at];
      if (lane == 0) { r2s0[first] = (u8)c; bstart[c] = acc + 1; }
      acc = (int32_t)((u32)acc + (u32)freq[c]);
      if (lane == 0) bend[c] = acc;
      if (acc < 0) general = true;                                 // Java int wrap
    }
    if (acc != count) general = true;
  }
  SRT_SYNC();
  if (bad) { if (lane == 0) { d_len2[b] = 0; d_flag[b] = 0; } return; }
  // Every step needs the next rank of the symbol that just became current: a dependent load.  Each symbol's
  // next 32 ranks are therefore cached in LDS (8 KiB per block): a step is an LDS read, global memory is touched once
  // per 32 ranks of a symbol.  Ranks are readable below `count` only (s[-H..-1] is the header, still inside the block).
  // list: position j -> lane j>>2, byte j&3
  u32 list = (u32)r2s0[4 * lane] | ((u32)r2s0[4 * lane + 1] << 8) | ((u32)r2s0[4 * lane + 2] << 16) | ((u32)r2s0[4 * lane + 3] << 24);
  int i = 0;
  int c = (int)(__builtin_amdgcn_readfirstlane((int)list) & 0xFF);
  if (!general) {
    // ---- well-formed input (every bucket ends inside the payload): the fast loop (round 3).  A step is ONE LDS round trip: the
    //      symbol's cursor / end and its 64-slot ring are read side by side (both addresses depend on c only), the zero run and the
    //      next non-zero rank come out of a ballot rotated by the cursor on the scalar unit.  A ring is refilled when the cursor
    //      reaches a multiple of 64: the load is issued at once and written to LDS one step later (or when its symbol comes
    //      up again first), so the memory latency is off the dependent chain. ----
    u8 (*ring)[64] = L.ring;
    for (int sym = 0; sym < 256; sym++) {
      const int32_t bs = bstart[sym];
      const int32_t t = (bs & ~63) + lane;                         // slot `lane` holds rank t
      if (t >= bs && t < bend[sym]) ring[sym][lane] = s[t];
    }
    SRT_SYNC();
    int pendSym = -1, pendCnt = 0, pendAge = 0;
    u32 vpend = 0;
#define SRT_FLUSH() do { if (lane < pendCnt) ring[pendSym][lane] = (u8)vpend; pendSym = -1; SRT_SYNC(); } while (0)
    while (i < count) {
    // Hot loop: while at least 66 outputs remain no step can run past the end (a step emits at most 65), so the common step -- a
    // non-zero rank inside the ring -- is straight-line code without clamps.  It is software pipelined: a non-zero rank moves c
    // behind position 1, so the NEXT current symbol is the list's position 1 and is known when the step starts; its cursor, end and
    // ring are requested from LDS at once and arrive while this step's ballot / scalar arithmetic / list move run: no LDS round
    // trip on the dependent chain.  Everything else (only zeros up to the ring's end, exhausted bucket, the block's last outputs)
    // leaves to the complete step below, one step at a time.
    if (i + 66 <= count) {
      if (pendSym >= 0) SRT_FLUSH();
      int32_t cur = __builtin_amdgcn_readfirstlane(bstart[c]), end = __builtin_amdgcn_readfirstlane(bend[c]);
      u32 v = ring[c][lane];
      for (;;) {
        const int c2 = (int)(((u32)__builtin_amdgcn_readfirstlane((int)list) >> 8) & 0xFF);
        // two rare cases leave the hot loop for one complete step (which re-reads everything; a pending refill is then written on
        // the way back in): c2's ring is still being refilled, or -- damaged input only -- the list holds the same symbol twice, so
        // that c2's state changes in this very step
        if (__builtin_expect(min((u32)(c2 ^ c), (u32)(c2 ^ pendSym)) == 0u, 0)) break;
        const int32_t cur2 = bstart[c2], end2 = bend[c2];             // (wave-uniform LDS reads; made scalar when they are used)
        const u32 v2 = ring[c2][lane];
        const int sh = cur & 63;
        const int avail = min(end - cur, 64 - sh);                    // valid ranks in the ring
        const uint64_t nzm = kz_ballot(v != 0);
        const uint64_t rot = nzm >> sh;                               // bit j: slot of rank cur + j is not zero (the ranks of the ring that
        const int z = rot ? (int)__builtin_ctzll(rot) : 64;           //   count sit in slots sh..63: avail <= 64 - sh, nothing wraps)
        if (__builtin_expect(z >= avail, 0)) break;                   // the first hit is valid iff it lies below avail
        const int r = __builtin_amdgcn_readlane((int)v, (cur + z) & 63);
        // all 64 lanes store: the bytes behind the z + 1 this step owns are written again by the steps that own them (a wave's stores
        // to one address land in program order, and at least 66 outputs remain), and the cursor goes out without an EXEC round trip:
        // lane 0 writes bstart[c], the other lanes a slot of their own in the table only the general loop uses
        o[(u32)i + (u32)lane] = (u8)c;
        i += z + 1;
        const int32_t ncur = cur + z + 1;
        *((lane == 0) ? &bstart[c] : &wbase[lane]) = ncur;
        if (__builtin_expect((ncur & 63) == 0, 0)) {                  // ring used up (one step in 64: one test on the common path)
          if (ncur < end) {                                           //   and the bucket goes on: fetch the next 64 ranks
            if (pendSym >= 0) SRT_FLUSH();
            pendCnt = min(64, end - ncur);
            vpend = (lane < pendCnt) ? (u32)s[ncur + lane] : 0u;
            pendSym = c;
          }
        }
        const u32 nxt = KZ_DPP_SHL1(list);                            // positions 0..r-1 <- 1..r, position r <- c (SRT.java:233-237)
        const u32 shifted = (list >> 8) | (nxt << 24);
        const int ec = min(max(r - 4 * lane, 0), 4);                  // bytes with position < r in this lane, branch-free
        const u32 mask = (u32)((0xFFFFFFFFull << (8 * ec)) >> 32);    // ec low bytes set
        const u32 nl = (shifted & mask) | (list & ~mask);
        const int shb = 8 * (r & 3);
        const u32 put = (nl & ~(0xFFu << shb)) | ((u32)c << shb);
        list = (lane == (r >> 2)) ? put : nl;
        c = c2;
        cur = __builtin_amdgcn_readfirstlane(cur2); end = __builtin_amdgcn_readfirstlane(end2); v = v2;
        if (__builtin_expect(i + 66 > count, 0)) break;
      }
    }
    pendAge = 1;
    if (i < count) {                                                  // ONE complete step, then back to the hot loop
      if (pendSym == c) SRT_FLUSH();
      // (LDS reads at a uniform address: readfirstlane tells the compiler so, and the step's arithmetic and branches stay scalar)
      const int32_t cur = __builtin_amdgcn_readfirstlane(bstart[c]), end = __builtin_amdgcn_readfirstlane(bend[c]);
      const u32 v = ring[c][lane];
      const int32_t fend = min(end, (cur & ~63) + 64);
      const int avail = fend - cur;                                  // valid ranks in the ring: 0..64
      const uint64_t nzm = kz_ballot(v != 0);
      const int sh = cur & 63;
      uint64_t rot = sh ? ((nzm >> sh) | (nzm << (64 - sh))) : nzm;  // bit j: rank cur + j is not zero
      if (avail < 64) rot &= ((1ull << avail) - 1ull);
      const int z = rot ? (int)__builtin_ctzll(rot) : avail;         // leading zero ranks = c repeats
      int r = 0, emit, consumed;
      bool moveC = false, removeC = false;
      if (rot) { emit = z + 1; consumed = z + 1; r = __builtin_amdgcn_readlane((int)v, (cur + z) & 63); moveC = true; }
      else if (fend >= end) { emit = avail + 1; consumed = avail; removeC = true; }   // bucket exhausted (:239-248)
      else { emit = avail; consumed = avail; }                        // only zeros up to the ring's end, more to come
      if (emit > count - i) { emit = count - i; moveC = false; removeC = false; }
      if (lane < emit) o[i + lane] = (u8)c;
      if (emit > 64 && lane == 0) o[i + 64] = (u8)c;
      i += emit;
      const int32_t ncur = cur + consumed;
      if (lane == 0) bstart[c] = ncur;
      if (consumed > 0 && ncur == fend && ncur < end) {               // ring used up, bucket goes on: fetch the next 64 ranks
        if (pendSym >= 0) SRT_FLUSH();
        pendCnt = min(64, end - ncur);
        vpend = (lane < pendCnt) ? (u32)s[ncur + lane] : 0u;
        pendSym = c; pendAge = 0;
      }
      if (moveC || (removeC && nbSymbols > 1)) {
        if (removeC) { nbSymbols--; r = nbSymbols; }
        const u32 nxt = KZ_DPP_SHL1(list);
        const u32 shifted = (list >> 8) | (nxt << 24);
        const int e = r - 4 * lane;                                   // bytes with position < r in this lane
        const u32 mask = (e <= 0) ? 0u : ((e >= 4) ? 0xFFFFFFFFu : (0xFFFFFFFFu >> (8 * (4 - e))));
        u32 nl = (shifted & mask) | (list & ~mask);
        if (moveC && lane == (r >> 2)) { const int shb = 8 * (r & 3); nl = (nl & ~(0xFFu << shb)) | ((u32)c << shb); }
        list = nl;
        c = (int)(__builtin_amdgcn_readfirstlane((int)list) & 0xFF);
      } else if (removeC) {
        for (int k = i + lane; k < count; k += 64) o[k] = (u8)c;      // single symbol left with an empty bucket (:239-240)
        i = count;
      }
      if (pendSym >= 0) { if (pendAge >= 1) SRT_FLUSH(); else pendAge++; }
    }
    }
#undef SRT_FLUSH
  } else {
  for (int sym = 0; sym < 256; sym++) {
    const int32_t bs = bstart[sym];
    const int32_t lim = min(bend[sym], count);
    if (lane < 32) win[sym][lane] = (bs + lane < lim) ? s[bs + lane] : (u8)0;
    if (lane == 0) wbase[sym] = bs;
  }
  SRT_SYNC();
  }
  // Well-formed input (frequencies add up to the payload: every bucket ends inside it) takes the tight loop; anything
  // else the general one, which also watches for buckets that run past the payload.
  if (general) { SRT_INV_LOOP(1) }
  if (lane == 0) { d_len2[b] = bad ? 0 : count; d_flag[b] = bad ? 0 : 1; }
}

size_t kz_srt_scratch(int B, int maxN) {
  const int T = (maxN + SR_TILE - 1) / SR_TILE + 1;
  const int64_t stride = (int64_t)kz_align((size_t)maxN + 4096 + 1024, 256);
  return (size_t)B * ((size_t)T * 1024 + 1024 + 64 + (size_t)stride) + kz_sbrt_scratch(B, maxN) + 16384;
}

int kz_stage_srt_forward(kz_ctx* ctx, kz_batch& bt) {
  const int B = bt.B;
  int maxN = 0;
  for (int b = 0; b < B; b++) if (bt.h_len[b] > maxN) maxN = bt.h_len[b];
  SrtFwd S;
  S.T = (maxN + SR_TILE - 1) / SR_TILE + 1;
  S.tileHist = (u32*)kz_arena_alloc(ctx, (size_t)B * S.T * 1024);
  S.bucket = (u32*)kz_arena_alloc(ctx, (size_t)B * 1024);
  S.hdrLen = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  S.ranks = (u8*)kz_arena_alloc(ctx, (size_t)bt.stride * B);
  if (!S.ranks || !S.hdrLen) { snprintf(ctx->err, sizeof(ctx->err), "srt_forward: arena overflow"); return -KZ_ERR_DEVICE; }
  const u8* src = bt.buf[bt.cur];
  u8* dst = bt.buf[bt.cur ^ 1];
  const int tiles = (maxN + SR_TILE - 1) / SR_TILE;
  if (tiles > 0) {
    KZ_LAUNCH(ctx, KID_SRT_HIST, k_srt_hist, dim3(tiles, B), dim3(KZ_WG), src, bt.stride, bt.d_len, S);
    KZ_LAUNCH(ctx, KID_SRT_PREP, k_srt_prep, dim3(B), dim3(256), bt.d_len, S, dst, bt.stride);
    int rc = kz_sbrt_ranks(ctx, src, S.ranks, bt.stride, bt.d_len, B, maxN, 4);
    if (rc) return rc;
    KZ_LAUNCH(ctx, KID_SRT_SCATTER, k_srt_scatter, dim3(tiles, B), dim3(KZ_WG), src, dst, bt.stride, bt.d_len, S);
  } else {
    KZ_HIP(hipMemsetAsync(S.hdrLen, 0, (size_t)B * 4, ctx->stream));
  }
  KZ_LAUNCH(ctx, KID_COPY_LEN, k_srt_ffin, dim3((B + 255) / 256), dim3(256), bt.d_len, bt.d_len2, bt.d_flag, S, B);
  KZ_HIP(hipGetLastError());
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}

int kz_stage_srt_inverse(kz_ctx* ctx, kz_batch& bt) {
  const int B = bt.B;
  const u8* src = bt.buf[bt.cur];
  u8* dst = bt.buf[bt.cur ^ 1];
  KzPlacement PL;
  { const int prc = kz_place_blocks(ctx, bt, PL); if (prc) return prc; }
  for (int rr = 0; rr < PL.R; rr++)
    KZ_LAUNCH(ctx, KID_SRT_INV, k_srt_inv, dim3(PL.G[rr]), dim3(64 * PL.wpg), src, dst, bt.stride, bt.d_len, bt.d_len2, bt.d_flag, PL.d_order + PL.off[rr], PL.wpg);
  KZ_HIP(hipGetLastError());
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}
