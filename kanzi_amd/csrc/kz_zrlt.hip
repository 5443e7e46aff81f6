// kz_zrlt.hip -- Zero Run Length Transform on gfx950 (scan-based, tile-parallel).
//
// Replaces K/transform/ZRLT.java:54-136 (forward) and :146-233 (inverse).
//  forward token rules (:80-130): zero run of length L -> bits of (L+1) below its MSB, one byte
//  (0/1) each, MSB first; byte v>=0xFE -> 0xFF,(v-0xFE); else v+1.  dstEnd = dstIdx0 + n ("do not
//  expand"): run needs dstIdx < dstEnd-log2 (:94), escape dstIdx < dstEnd-1 (:111), literal
//  dstIdx < dstEnd (:120); any violation => forward returns false (transform skipped).
//
// Parallel form: a token is emitted by the lane that owns the byte where it ENDS (forward: one lane per
// byte, rows of 64).  The length of a zero run reaching back across rows / waves / tiles is (own position) -
// (position after the last non-zero byte before it), an exclusive MAX-scan of "last non-zero position + 1":
// ballot masks inside a row, LDS between the waves of a tile, tile summaries over the block
// (k_zrlt_f2).  Output offsets are an exclusive SUM-scan of token sizes over the same hierarchy.
#include "kz_device.h"
#include "kz_internal.h"

typedef uint32_t u32;
typedef uint8_t u8;

#define ZR_PER 16                     // forward: bytes per thread of the tile's load (one 16-byte access)
#define ZR_TILE (KZ_WG * ZR_PER)      // bytes per workgroup, forward
#define ZI_PER 16                     // inverse: tokens are classified byte by byte with look-back
#define ZI_TILE (KZ_WG * ZI_PER)

struct ZrScratch {
  u32* tLastNz;   // [B][T]  abs position+1 of the last non-zero byte of the tile (0 = none)
  u32* tInner;    // [B][T]  tokens whose size is known inside the tile
  u32* tLead;     // [B][T]  leading zeros of the tile (tile length when all zero)
  u32* tP;        // [B][T]  exclusive max-scan of tLastNz
  u32* tOff;      // [B][T]  exclusive sum-scan of tile output sizes
  int32_t* fail;  // [B]
  int32_t* total; // [B]
  int T;
};

// ---- forward 2/3: per block scan over tiles (one wave) ----------------------------------------
__global__ __launch_bounds__(64) void k_zrlt_f2(const int32_t* __restrict__ d_len, ZrScratch S) {
  const int b = blockIdx.x;
  const int n = d_len[b];
  const int tiles = (n + ZR_TILE - 1) / ZR_TILE;
  const int64_t o = (int64_t)b * S.T;
  const int lane = kz_lane();
  u32 carryMax = 0, carrySum = 0;
  for (int base = 0; base < tiles; base += 64) {
    const int t = base + lane;
    const bool ok = t < tiles;
    const u32 ln = ok ? S.tLastNz[o + t] : 0;
    u32 incm = kz_wave_incl_max(ln);
    u32 P = __shfl_up(incm, 1, 64); if (lane == 0) P = 0;
    P = P > carryMax ? P : carryMax;                       // abs pos+1 of last non-zero before the tile
    u32 sz = 0;
    if (ok) {
      const u32 tstart = (u32)t * ZR_TILE;
      const u32 tlen = (u32)min(ZR_TILE, n - (int)tstart);
      const u32 lead = S.tLead[o + t];
      const bool allzero = (ln == 0);
      sz = S.tInner[o + t];
      if (!allzero) { const u32 run = tstart - P + lead; if (run > 0) sz += (u32)kz_ilog2(run + 1); }
      if (t == tiles - 1) {
        // trailing run of the block whose start lies before the last non-zero of ... any tile
        const u32 lastAll = ln > P ? ln : P;                // last non-zero over the whole block
        const u32 R = (u32)n - lastAll;
        // f1 already counted it when the owning thread knew its start (a non-zero precedes it in-tile)
        const bool countedInF1 = (ln != 0);
        if (R > 0 && !countedInF1) sz += (u32)kz_ilog2(R + 1);
        (void)tlen;
      }
      S.tP[o + t] = P;
    }
    u32 incs = kz_wave_incl_sum(sz);
    if (ok) S.tOff[o + t] = carrySum + incs - sz;
    carrySum += __shfl(incs, 63, 64);
    u32 lastm = __shfl(incm, 63, 64);
    carryMax = carryMax > lastm ? carryMax : lastm;
  }
  if (lane == 0) { S.total[b] = (int32_t)carrySum; S.fail[b] = 0; }
}

// ---- forward 1/3 and 3/3, one LANE per byte ----------------------------------------------------------------------------------
// Until round 4 every thread took 16 consecutive bytes and walked them with per-lane control flow (zero? run ends? how many
// digits?): ~4 300 instructions per wave and kilobyte, most of them exec-mask bookkeeping, and sixteen scattered one-byte stores per
// lane (k_zrlt_f3 37 ms per 8 GiB; now 16).  Here a wave takes its kilobyte as 16 rows of 64 bytes, one byte per lane: the non-zero bytes of a row are a ballot mask, the zero run in front of a non-zero
// byte is the distance to the previous set bit (or to the last non-zero byte of the rows / waves / tiles before), output offsets are
// a wave scan per row.  The tile summaries are the ones k_zrlt_f2 expects (same definitions as above).
#define ZR_WROWS 16                  // rows of 64 bytes per wave: KZ_WG / 64 waves x 16 x 64 = ZR_TILE
static_assert(ZR_TILE == (KZ_WG / 64) * ZR_WROWS * 64, "ZRLT forward tile geometry");

// the tile enters LDS with one 16-byte load per thread (zeros behind the block's end); the rows are read from there
#define ZR_LOAD_ROWS(INB)                                                                                \
  {                                                                                                      \
    const int p16 = tstart + 16 * (int)threadIdx.x;                                                      \
    uint4 q = make_uint4(0u, 0u, 0u, 0u);                                                                \
    if (p16 + 16 <= n) q = *(const uint4*)(s + p16);                                                     \
    else if (p16 < n) { u8 tb[16]; for (int k = 0; k < 16; k++) tb[k] = (p16 + k < n) ? s[p16 + k] : (u8)0; memcpy(&q, tb, 16); } \
    ((uint4*)(INB))[threadIdx.x] = q;                                                                    \
  }                                                                                                      \
  __syncthreads();                                                                                       \
  u32 v[ZR_WROWS]; uint64_t nz[ZR_WROWS];                                                                \
  u32 myLast = 0, myFirst = 0xFFFFFFFFu;                      /* abs position + 1 of the wave's last / abs position of its first non-zero byte */ \
  _Pragma("unroll") for (int r = 0; r < ZR_WROWS; r++) {                                                 \
    v[r] = (u32)(INB)[wave * (64 * ZR_WROWS) + 64 * r + lane];                                           \
    nz[r] = kz_ballot(v[r] != 0u);                                                                       \
    if (nz[r]) {                                                                                         \
      myLast = (u32)(wbase + 64 * r + 64 - (int)__builtin_clzll(nz[r]));                                 \
      if (myFirst == 0xFFFFFFFFu) myFirst = (u32)(wbase + 64 * r + (int)__builtin_ctzll(nz[r]));         \
    }                                                                                                    \
  }

__global__ __launch_bounds__(KZ_WG) void k_zrlt_f1(const u8* __restrict__ src, int64_t stride, const int32_t* __restrict__ d_len, ZrScratch S) {
  const int b = blockIdx.y, t = blockIdx.x;
  const int n = d_len[b];
  const int tstart = t * ZR_TILE;
  if (tstart >= n) return;
  const u8* s = src + (int64_t)b * stride;
  const int wave = threadIdx.x >> 6, lane = kz_lane();
  const int wbase = tstart + wave * (64 * ZR_WROWS);
  __shared__ u32 wLast[KZ_WG / 64], wFirst[KZ_WG / 64], wSum[KZ_WG / 64];
  __shared__ __attribute__((aligned(16))) u8 inb[ZR_TILE];
  ZR_LOAD_ROWS(inb)
  if (lane == 0) { wLast[wave] = myLast; wFirst[wave] = myFirst; }
  __syncthreads();
  u32 carryLast = 0;                                            // abs position + 1 of the last non-zero byte before the row, INSIDE the tile (0: none)
  for (int w = 0; w < wave; w++) carryLast = max(carryLast, wLast[w]);
  const uint64_t lt = kz_lanemask_lt();
  u32 sum = 0;
#pragma unroll
  for (int r = 0; r < ZR_WROWS; r++) {
    const uint64_t m = nz[r];
    if (m == 0) continue;                                       // uniform
    const int pos = wbase + 64 * r + lane;
    const uint64_t below = m & lt;
    const u32 prevp1 = below ? (u32)(wbase + 64 * r + 64 - (int)__builtin_clzll(below)) : carryLast;
    if ((m >> lane) & 1ULL) {
      u32 tok = (v[r] >= 0xFEu) ? 2u : 1u;
      // the run in front of the tile's first non-zero byte starts in an earlier tile: k_zrlt_f2 adds its digits
      if (prevp1 > 0u && (u32)pos > prevp1) tok += (u32)kz_ilog2((u32)pos - prevp1 + 1u);
      sum += tok;
    }
    carryLast = (u32)(wbase + 64 * r + 64 - (int)__builtin_clzll(m));
  }
  const u32 inc = kz_wave_incl_sum(sum);
  if (lane == 63) wSum[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 tileLast = 0, tileFirst = 0xFFFFFFFFu, total = 0;
    for (int w = 0; w < KZ_WG / 64; w++) { tileLast = max(tileLast, wLast[w]); tileFirst = min(tileFirst, wFirst[w]); total += wSum[w]; }
    const int tend = min(tstart + ZR_TILE, n);
    // the block's trailing run is counted here when a non-zero byte precedes it inside the tile (else in k_zrlt_f2)
    if (tend == n && tileLast > 0u && (u32)n > tileLast) total += (u32)kz_ilog2((u32)n - tileLast + 1u);
    const int64_t o = (int64_t)b * S.T + t;
    S.tLastNz[o] = tileLast;
    S.tInner[o] = total;
    S.tLead[o] = (tileFirst == 0xFFFFFFFFu) ? (u32)(tend - tstart) : tileFirst - (u32)tstart;
  }
}

__global__ __launch_bounds__(KZ_WG) void k_zrlt_f3(const u8* __restrict__ src, u8* __restrict__ dst, int64_t stride,
                                                     const int32_t* __restrict__ d_len, ZrScratch S) {
  const int b = blockIdx.y, t = blockIdx.x;
  const int n = d_len[b];
  const int tstart = t * ZR_TILE;
  if (tstart >= n) return;
  const u8* s = src + (int64_t)b * stride;
  u8* d = dst + (int64_t)b * stride;
  const int wave = threadIdx.x >> 6, lane = kz_lane();
  const int wbase = tstart + wave * (64 * ZR_WROWS);
  __shared__ u32 wLast[KZ_WG / 64], wSum[KZ_WG / 64];
  // The tile's tokens are contiguous in the output ([tileBase, tileBase + total)): they are put together in LDS, at the output's
  // alignment modulo 16, and copied out in aligned 16-byte pieces.  At most two bytes per input byte plus one long run: 2 ZR_TILE + 32.
  __shared__ __attribute__((aligned(16))) u8 stage[2 * ZR_TILE + 64];
  ZR_LOAD_ROWS(stage)                                           // (the staging buffer holds the input first: nothing is staged before the next barrier)
  (void)myFirst;
  if (lane == 0) wLast[wave] = myLast;
  __syncthreads();
  const u32 tileP = S.tP[(int64_t)b * S.T + t];               // abs position + 1 of the last non-zero byte before the tile (0: none)
  const u32 tileBase = S.tOff[(int64_t)b * S.T + t];
  const u32 base16 = tileBase & ~15u;
  u32 carry0 = tileP;
  for (int w = 0; w < wave; w++) carry0 = max(carry0, wLast[w]);
  const uint64_t lt = kz_lanemask_lt();
  // pass 1: token sizes (run digits + 1 or 2), the run's length + 1 kept for pass 2
  u32 tokv[ZR_WROWS], rlv[ZR_WROWS];
  u32 sum = 0;
  {
    u32 carryLast = carry0;
#pragma unroll
    for (int r = 0; r < ZR_WROWS; r++) {
      const uint64_t m = nz[r];
      tokv[r] = 0; rlv[r] = 1;
      if (m == 0) continue;                                     // uniform
      const int pos = wbase + 64 * r + lane;
      const uint64_t below = m & lt;
      const u32 prevp1 = below ? (u32)(wbase + 64 * r + 64 - (int)__builtin_clzll(below)) : carryLast;
      if ((m >> lane) & 1ULL) {
        const u32 rl = (u32)pos - prevp1 + 1u;                  // zero run in front of this byte, + 1
        rlv[r] = rl;
        tokv[r] = ((v[r] >= 0xFEu) ? 2u : 1u) + (u32)kz_ilog2(rl);
        sum += tokv[r];
      }
      carryLast = (u32)(wbase + 64 * r + 64 - (int)__builtin_clzll(m));
    }
  }
  {
    const u32 inc = kz_wave_incl_sum(sum);
    if (lane == 63) wSum[wave] = inc;
  }
  __syncthreads();
  u32 off0 = tileBase, inner = 0, tileLast = 0;
  for (int w = 0; w < KZ_WG / 64; w++) { if (w < wave) off0 += wSum[w]; inner += wSum[w]; tileLast = max(tileLast, wLast[w]); }
  // pass 2: emit with the reference's bound checks; dstEnd = n
  const u32 dstEnd = (u32)n;
  bool fail = false;
#pragma unroll
  for (int r = 0; r < ZR_WROWS; r++) {
    if (nz[r] == 0) continue;                                   // uniform
    const u32 tok = tokv[r];
    const u32 inc = kz_wave_incl_sum(tok);
    u32 off = off0 + inc - tok;
    off0 += (u32)__builtin_amdgcn_readlane((int)inc, 63);
    const u32 rl = rlv[r];
    const int lg = kz_ilog2(rl);
    const bool runBad = (lg > 0) && (off + (u32)lg >= dstEnd);            // ZRLT.java:94  dstIdx >= dstEnd - log2
    fail |= runBad;
    for (int dg = 0; kz_ballot(dg < lg) != 0; dg++)
      if (dg < lg && !runBad) stage[off + (u32)dg - base16] = (u8)((rl >> (lg - 1 - dg)) & 1u);
    off += (u32)lg;
    if (tok != 0u) {
      const u32 val = v[r];
      if (val >= 0xFEu) {
        if (off + 1u >= dstEnd) fail = true;                               // :111
        else { stage[off - base16] = 0xFF; stage[off + 1u - base16] = (u8)(val - 0xFEu); }
      } else {
        if (off >= dstEnd) fail = true;                                    // :120
        else stage[off - base16] = (u8)(val + 1u);
      }
    }
  }
  // the block's trailing run (:217-228 of the forward loop's tail): behind every other token of the last tile
  u32 total = inner;
  {
    const int tend = min(tstart + ZR_TILE, n);
    const u32 lastAll = max(tileP, tileLast);
    if (tend == n && (u32)n > lastAll) {
      const u32 rl = (u32)n - lastAll + 1u;
      const int lg = kz_ilog2(rl);
      if (threadIdx.x == 0) {
        const u32 off = tileBase + inner;
        if (off + (u32)lg >= dstEnd) fail = true;
        else for (int dg = 0; dg < lg; dg++) stage[off + (u32)dg - base16] = (u8)((rl >> (lg - 1 - dg)) & 1u);
      }
      total += (u32)lg;
    }
  }
  if (fail) atomicOr(&S.fail[b], 1);
  __syncthreads();
  // (a block that failed a check is declined and its output discarded, k_zrlt_ffin: nothing is written at or behind dstEnd)
  const u32 endLim = min(tileBase + total, dstEnd);
  const bool al = (((uintptr_t)d) & 15) == 0;                      // (the batch buffers are: 256-byte aligned slots)
  for (u32 g0 = base16 + 16u * threadIdx.x; g0 < endLim; g0 += 16u * KZ_WG) {
    if (al && g0 >= tileBase && g0 + 16 <= endLim) *(uint4*)(d + g0) = *(const uint4*)(stage + (g0 - base16));
    else {
#pragma unroll
      for (int k = 0; k < 16; k++) { const u32 g = g0 + k; if (g >= tileBase && g < endLim) d[g] = stage[g - base16]; }
    }
  }
}

// finalize: applied -> length = total ; declined -> copy input through (Sequence.java:95-105)
__global__ void k_zrlt_ffin(const u8* __restrict__ src, u8* __restrict__ dst, int64_t stride,
                            const int32_t* __restrict__ d_len, int32_t* __restrict__ d_len2, int32_t* __restrict__ d_flag, ZrScratch S) {
  const int b = blockIdx.y;
  const int n = d_len[b];
  const bool fail = (S.fail[b] != 0);
  if (blockIdx.x == 0 && threadIdx.x == 0) { d_len2[b] = fail ? n : ((n == 0) ? 0 : S.total[b]); d_flag[b] = fail ? 0 : 1; }
  if (!fail) return;
  const u8* s = src + (int64_t)b * stride;
  u8* d = dst + (int64_t)b * stride;
  for (int i = (blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += gridDim.x * blockDim.x * 16) {
    if (i + 16 <= n) *(uint4*)(d + i) = *(const uint4*)(s + i);
    else for (int k = i; k < n; k++) d[k] = s[k];
  }
}

// =================================================================================================
// inverse
// classification of input byte i (ZRLT.java:167-214): payload = byte following an escape 0xFF
__device__ __forceinline__ bool zr_is_payload(const u8* s, int i) {
  int k = 0;
  while (i - 1 - k >= 0 && s[i - 1 - k] == 0xFF) k++;
  return (k & 1) != 0;
}

// output bytes produced by the token STARTING at input byte i (0 if i is inside a token)
// Output bytes produced by the token starting at i (0 for bytes that continue a token).  A zero run of k digits has
// the value (1 << k | digits) - 1 in Java int arithmetic (ZRLT.java:172-188): runs of more than 30 digits -- which no
// encoder writes -- wrap, and the reference then emits the wrapped count (or nothing when it is <= 0).  A run that
// reaches the end of the input goes through the trailing branch (:217-228), whose test differs at INT_MIN only.
// A count that cannot fit fails later through the total (every such case dies in the reference as well).
// *end (optional): the input position behind the token, set when it starts here.
__device__ __forceinline__ u32 zr_inv_token(const u8* s, int i, int n, int* end) {
  const u32 v = s[i];
  const bool payload = zr_is_payload(s, i);
  if (payload) return 0;
  if (v <= 1) {
    if (i > 0 && s[i - 1] <= 1 && !zr_is_payload(s, i - 1)) return 0;     // not the head digit
    u32 rl = 1;
    int k = i;
    while (k < n && s[k] <= 1) {
      rl = (rl << 1) | s[k]; k++;
      // long runs (corrupted input only): 8 digits per step; blocks start 256-byte aligned
      while ((k & 7) == 0 && k + 8 <= n) {
        const unsigned long long w = *(const unsigned long long*)(s + k);
        if (w & 0xFEFEFEFEFEFEFEFEull) break;
        rl = (rl << 8) | (u32)((w * 0x8040201008040201ull) >> 56);         // byte j of w -> bit 7-j
        k += 8;
      }
    }
    if (end) *end = k;
    if (k >= n) return ((int32_t)rl > 0) ? rl - 1u : 0u;
    const int32_t r = (int32_t)(rl - 1u);
    return (r > 0) ? (u32)r : 0u;
  }
  if (end) *end = (v == 0xFF) ? i + 2 : i + 1;
  if (v == 0xFF) return (i + 1 < n) ? 1u : 0u;
  return 1u;
}

struct ZiScratch { u32* tSum; u32* tOff; u32* tEnd; int32_t* total; int32_t* fail; int T; };   // tEnd: input position behind the tile's last token that produces output (0: none)

__global__ __launch_bounds__(KZ_WG) void k_zrlt_i1(const u8* __restrict__ src, int64_t stride, const int32_t* __restrict__ d_len, ZiScratch S) {
  const int b = blockIdx.y, t = blockIdx.x;
  const int n = d_len[b];
  const int tstart = t * ZI_TILE;
  if (tstart >= n) return;
  __shared__ __attribute__((aligned(8))) u32 lds[32];
  const u8* s = src + (int64_t)b * stride;
  const int pos = tstart + threadIdx.x * ZI_PER;
  unsigned long long sz = 0;
  int lastEnd = 0;                                                // behind the last token of this thread that produces output
  for (int k = 0; k < ZI_PER; k++) if (pos + k < n) { int e = 0; const u32 one = zr_inv_token(s, pos + k, n, &e); sz += one; if (one) lastEnd = e; }
  // tile total, saturated: wrapped run counts can be anything below 2^31
  for (int d = 1; d < 64; d <<= 1) { sz += __shfl_xor(sz, d, 64); lastEnd = max(lastEnd, __shfl_xor(lastEnd, d, 64)); }
  unsigned long long* l64 = (unsigned long long*)lds;
  __shared__ int wEnd[KZ_WG / 64];
  if ((threadIdx.x & 63) == 0) { l64[threadIdx.x >> 6] = sz; wEnd[threadIdx.x >> 6] = lastEnd; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long total = 0;
    int e = 0;
    for (int w = 0; w < KZ_WG / 64; w++) { total += l64[w]; e = max(e, wEnd[w]); }
    S.tSum[(int64_t)b * S.T + t] = (total > 0xFFFFFFFFull) ? 0xFFFFFFFFu : (u32)total;
    S.tEnd[(int64_t)b * S.T + t] = (u32)e;
  }
}

__global__ __launch_bounds__(64) void k_zrlt_i2(const int32_t* __restrict__ d_len, ZiScratch S, int dstCap) {
  const int b = blockIdx.x;
  const int n = d_len[b];
  const int tiles = (n + ZI_TILE - 1) / ZI_TILE;
  const int64_t o = (int64_t)b * S.T;
  const int lane = kz_lane();
  unsigned long long carry = 0;
  u32 lastOut = 0;                                                  // input position behind the block's last token that produces output
  for (int base = 0; base < tiles; base += 64) {
    const int t = base + lane;
    const u32 v = (t < tiles) ? S.tSum[o + t] : 0;
    lastOut = max(lastOut, (t < tiles) ? S.tEnd[o + t] : 0u);
    unsigned long long inc = v;                                     // 64-bit: tile totals may be saturated
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long up = __shfl_up(inc, d, 64); if (lane >= d) inc += up; }
    if (t < tiles) S.tOff[o + t] = (u32)(carry + inc - v);          // only used when the total fits
    carry += __shfl(inc, 63, 64);
  }
  for (int d = 1; d < 64; d <<= 1) lastOut = max(lastOut, (u32)__shfl_xor((int)lastOut, d, 64));
  if (lane == 0) {
    // ZRLT.java:166-233 leaves its loop as soon as the output is full (dstIdx >= dstEnd, :214; a run that would end AT dstEnd, :184) and
    // then reports whether the input was used up: an output that fits exactly is an error when input is left behind the token that
    // filled it -- a lone escape at the end, an over-long run whose wrapped count is <= 0 (both produce nothing).
    if (carry > (unsigned long long)dstCap || (carry == (unsigned long long)dstCap && lastOut < (u32)n)) { S.fail[b] = 1; S.total[b] = 0; }
    else S.total[b] = (int32_t)carry;
  }
}

// zero the output of the blocks this call decodes, and only theirs (zero runs are "written" here).  Other slots of the
// batch may be in use by another stream at this moment (the decoder's expensive-blocks-first schedule, kz_api.hip).
__global__ __launch_bounds__(256) void k_zrlt_izero(u8* __restrict__ dst, int64_t stride, const int32_t* __restrict__ d_len, ZiScratch S) {
  const int b = blockIdx.y;
  if (d_len[b] <= 0 || S.fail[b]) return;
  const int64_t n = ((int64_t)S.total[b] + 15) & ~15LL;             // slots are 256-byte aligned with >= 4 KiB of slack
  uint4* d = (uint4*)(dst + (int64_t)b * stride);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i * 16 < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = make_uint4(0, 0, 0, 0);
}

__global__ __launch_bounds__(KZ_WG) void k_zrlt_i3(const u8* __restrict__ src, u8* __restrict__ dst, int64_t stride,
                                                    const int32_t* __restrict__ d_len, ZiScratch S) {
  const int b = blockIdx.y, t = blockIdx.x;
  const int n = d_len[b];
  const int tstart = t * ZI_TILE;
  if (tstart >= n || S.fail[b]) return;
  __shared__ u32 lds[32];
  const u8* s = src + (int64_t)b * stride;
  u8* d = dst + (int64_t)b * stride;
  const int pos = tstart + threadIdx.x * ZI_PER;
  u32 sz[ZI_PER]; u32 sum = 0;
  for (int k = 0; k < ZI_PER; k++) { sz[k] = (pos + k < n) ? zr_inv_token(s, pos + k, n, nullptr) : 0; sum += sz[k]; }
  u32 total;
  u32 off = kz_wg_excl_sum(sum, lds, &total) + S.tOff[(int64_t)b * S.T + t];
  for (int k = 0; k < ZI_PER; k++) {
    const int i = pos + k;
    if (i >= n) break;
    const u32 v = s[i];
    if (sz[k] && v > 1) d[off] = (v == 0xFF) ? (u8)(0xFE + s[i + 1]) : (u8)(v - 1);   // zero runs: dst pre-zeroed
    off += sz[k];
  }
}

__global__ void k_zrlt_ifin(const int32_t* __restrict__ d_len, int32_t* __restrict__ d_len2, int32_t* __restrict__ d_flag, ZiScratch S, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const bool fail = S.fail[b] != 0;
  d_len2[b] = fail ? 0 : ((d_len[b] == 0) ? 0 : S.total[b]);
  d_flag[b] = fail ? 0 : 1;
}

size_t kz_zrlt_scratch(int B, int maxN) {
  const int T = (maxN + 64 + ZI_TILE - 1) / ZI_TILE + 1;      // the inverse's (smaller) tiles
  return (size_t)B * T * 4 * 6 + (size_t)B * 16 + 8192;
}

int kz_stage_zrlt_forward(kz_ctx* ctx, kz_batch& bt) {
  const int B = bt.B;
  int maxN = 0;
  for (int b = 0; b < B; b++) if (bt.h_len[b] > maxN) maxN = bt.h_len[b];
  ZrScratch S;
  S.T = (maxN + ZR_TILE - 1) / ZR_TILE + 1;
  S.tLastNz = (u32*)kz_arena_alloc(ctx, (size_t)B * S.T * 4);
  S.tInner = (u32*)kz_arena_alloc(ctx, (size_t)B * S.T * 4);
  S.tLead = (u32*)kz_arena_alloc(ctx, (size_t)B * S.T * 4);
  S.tP = (u32*)kz_arena_alloc(ctx, (size_t)B * S.T * 4);
  S.tOff = (u32*)kz_arena_alloc(ctx, (size_t)B * S.T * 4);
  S.fail = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  S.total = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  if (!S.total) { snprintf(ctx->err, sizeof(ctx->err), "zrlt_forward: arena overflow"); return -KZ_ERR_DEVICE; }
  hipStream_t st = ctx->stream;
  const u8* src = bt.buf[bt.cur];
  u8* dst = bt.buf[bt.cur ^ 1];
  KZ_HIP(hipMemsetAsync(S.fail, 0, (size_t)B * 4, st));
  KZ_HIP(hipMemsetAsync(S.total, 0, (size_t)B * 4, st));
  if (maxN > 0) {
    const int tiles = (maxN + ZR_TILE - 1) / ZR_TILE;
    KZ_LAUNCH(ctx, KID_ZRLT_F1, k_zrlt_f1, dim3(tiles, B), dim3(KZ_WG), src, bt.stride, bt.d_len, S);
    KZ_LAUNCH(ctx, KID_ZRLT_F2, k_zrlt_f2, dim3(B), dim3(64), bt.d_len, S);
    KZ_LAUNCH(ctx, KID_ZRLT_F3, k_zrlt_f3, dim3(tiles, B), dim3(KZ_WG), src, dst, bt.stride, bt.d_len, S);
  }
  KZ_LAUNCH(ctx, KID_ZRLT_FFIN, k_zrlt_ffin, dim3(64, B), dim3(256), src, dst, bt.stride, bt.d_len, bt.d_len2, bt.d_flag, S);
  KZ_HIP(hipGetLastError());
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}

int kz_stage_zrlt_inverse(kz_ctx* ctx, kz_batch& bt, int dstCap) {
  const int B = bt.B;
  int maxN = 0;
  for (int b = 0; b < B; b++) if (bt.h_len[b] > maxN) maxN = bt.h_len[b];
  ZiScratch S;
  S.T = (maxN + ZI_TILE - 1) / ZI_TILE + 1;
  S.tSum = (u32*)kz_arena_alloc(ctx, (size_t)B * S.T * 4);
  S.tOff = (u32*)kz_arena_alloc(ctx, (size_t)B * S.T * 4);
  S.tEnd = (u32*)kz_arena_alloc(ctx, (size_t)B * S.T * 4);
  S.fail = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  S.total = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  if (!S.total) { snprintf(ctx->err, sizeof(ctx->err), "zrlt_inverse: arena overflow"); return -KZ_ERR_DEVICE; }
  hipStream_t st = ctx->stream;
  const u8* src = bt.buf[bt.cur];
  u8* dst = bt.buf[bt.cur ^ 1];
  if ((int64_t)dstCap > bt.stride) dstCap = (int)bt.stride;
  KZ_HIP(hipMemsetAsync(S.fail, 0, (size_t)B * 4, st));
  KZ_HIP(hipMemsetAsync(S.total, 0, (size_t)B * 4, st));
  if (maxN > 0) {
    const int tiles = (maxN + ZI_TILE - 1) / ZI_TILE;
    KZ_LAUNCH(ctx, KID_ZRLT_I1, k_zrlt_i1, dim3(tiles, B), dim3(KZ_WG), src, bt.stride, bt.d_len, S);
    KZ_LAUNCH(ctx, KID_ZRLT_I2, k_zrlt_i2, dim3(B), dim3(64), bt.d_len, S, dstCap);
    KZ_LAUNCH(ctx, KID_ZRLT_I2, k_zrlt_izero, dim3(64, B), dim3(256), dst, bt.stride, bt.d_len, S);
    KZ_LAUNCH(ctx, KID_ZRLT_I3, k_zrlt_i3, dim3(tiles, B), dim3(KZ_WG), src, dst, bt.stride, bt.d_len, S);
  }
  KZ_LAUNCH(ctx, KID_ZRLT_IFIN, k_zrlt_ifin, dim3((B + 255) / 256), dim3(256), bt.d_len, bt.d_len2, bt.d_flag, S, B);
  KZ_HIP(hipGetLastError());
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}
