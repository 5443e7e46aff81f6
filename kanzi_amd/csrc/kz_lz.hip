// kz_lz.hip -- LZ / LZX (LZXCodec) on gfx950, first correct version.
//
// Replaces K/transform/LZCodec.java:299-597 (forward), :904-911 (hash), :271-287 (findMatch),
// :211-231 (emitLength), :626-756 (inverseV6).
//
// The parse is a sequential state machine (hash-table last-writer, 2 repeat distances, lazy +1/+2
// evaluation, skip acceleration srcInc>>6: SURVEY F6), and bit-exactness needs the exact visiting
// order.  This version runs it as one wave per block with wave-uniform control flow: every lane
// executes the same scalar logic (loads of one address broadcast), lane 0 performs the stores, and the
// bulk copies (literals, final section assembly, match copies in the inverse) use all 64 lanes.  The hash
// table (2^16 / 2^19 ints) lives in HBM/L2.  Throughput comes from blocks in flight; a wave-speculative
// parse is the planned replacement.
#include "kz_device.h"
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "k_lz_inv relies on the in-order vmcnt store -> load visibility of one wave on gfx9 / CDNA (no separate store counter): see the note at its match copy"
#endif
#include "kz_internal.h"

typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint8_t u8;

// The workgroup is ONE wave: vector memory operations of a wave are issued and performed in program order (and the
// per-CU L1 is write-through), so a load that follows a store to the same address sees it without s_barrier; a
// real barrier would also drain every outstanding store (s_waitcnt vmcnt(0)) at each parse step.  Only the compiler
// has to be kept from reordering.
// Round 3: the ordering is now SAID, not just relied on: an acquire-release fence at wavefront scope (lane 0's table store must be
// visible to the loads the other lanes of the same wave issue next).  At that scope the back end emits no instruction -- the ISA of
// this file with and without the fence differs in one commuted compare -- so it costs nothing; it is what the memory model asks for.
#define LZ_ORDER() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
// per-section cycle counters of block 0 (build with -DKZ_LZ_PROF; tools/README): where a step's time goes
#ifdef KZ_LZ_PROF
#include <stdio.h>
#define LZP(k) { const long long t_ = (long long)__builtin_readcyclecounter(); lzp[k] += t_ - lzt; lzn[k]++; lzt = t_; }
#else
#define LZP(k)
#endif
#define LZ_SEED 0x1E35A7BDULL
#define LZ_MAXD1 ((1 << 16) - 2)
#define LZ_MAXD2 ((1 << 24) - 2)
#define LZ_MAX_MATCH (65535 + 254 + 4)
#define LZ_MIN_BLOCK 24

// little-endian loads at arbitrary byte addresses: global memory supports unaligned accesses, one load each
typedef u64 __attribute__((aligned(1))) lz_u64_unaligned;
typedef u32 __attribute__((aligned(1))) lz_u32_unaligned;
__device__ __forceinline__ u64 lz_le64(const u8* p) { return *(const lz_u64_unaligned*)p; }
__device__ __forceinline__ u32 lz_le32(const u8* p) { return *(const lz_u32_unaligned*)p; }
__device__ __forceinline__ int lz_hash(const u8* p, int extra) { return (int)(((lz_le64(p) << 24) * LZ_SEED) >> (extra ? 45 : 48)); }
__device__ __forceinline__ bool lz_diff4(const u8* a, int i, int j) { return lz_le32(a + i) != lz_le32(a + j); }
// match length (LZCodec.java:271-287: 8 bytes per step, same result): all arguments are wave-uniform; the 64 lanes
// compare 512 bytes per round trip instead of 8
__device__ __forceinline__ int lz_find_match(const u8* src, int srcIdx, int ref, int maxMatch) {
  const int lane = kz_lane();
  int bestLen = 0;
  while (bestLen + 8 <= maxMatch) {
    const int off = bestLen + 8 * lane;
    const bool in = off + 8 <= maxMatch;
    const u64 diff = in ? (lz_le64(src + srcIdx + off) ^ lz_le64(src + ref + off)) : 0ULL;
    const uint64_t inMask = kz_ballot(in);
    const uint64_t dm = kz_ballot(in && diff != 0);
    if (dm) {
      const int l = (int)__builtin_ctzll(dm);
      const u32 dlo = (u32)__builtin_amdgcn_readlane((int)(u32)diff, l), dhi = (u32)__builtin_amdgcn_readlane((int)(u32)(diff >> 32), l);
      const u64 dd = ((u64)dhi << 32) | dlo;
      return bestLen + 8 * l + (int)(__builtin_ctzll(dd) >> 3);
    }
    bestLen += 8 * (int)__builtin_popcountll(inMask);
  }
  return bestLen;
}
// lane-0 stores with uniform index arithmetic
__device__ __forceinline__ int lz_emit_length(u8* block, int idx, int length, bool w) {
  if (length < 254) { if (w) block[idx] = (u8)length; return idx + 1; }
  if (length < 65536 + 254) { length -= 254; if (w) { block[idx] = 254; block[idx + 1] = (u8)(length >> 8); block[idx + 2] = (u8)length; } return idx + 3; }
  length -= 255;
  if (w) { block[idx] = 255; block[idx + 1] = (u8)(length >> 16); block[idx + 2] = (u8)(length >> 8); block[idx + 3] = (u8)length; }
  return idx + 4;
}
__device__ __forceinline__ void lz_copy(u8* d, const u8* s, int n) { for (int i = kz_lane(); i < n; i += 64) d[i] = s[i]; }

__global__ __launch_bounds__(128) void k_lz_fwd(const u8* __restrict__ srcAll, u8* __restrict__ dstAll, int64_t stride,
                                                const int32_t* __restrict__ d_len, int32_t* __restrict__ d_len2, int32_t* __restrict__ d_flag,
                                                int32_t* __restrict__ hashAll, u8* __restrict__ tmpAll, int64_t tmpStride, int extra,
                                                const int32_t* __restrict__ d_dtype) {
  // Two waves per block: wave 0 parses, wave 1 only warms the caches ahead of it (the assist below).
  __shared__ volatile int progress[2];                            // [0] the parser's position, [1] parser done
  const int b = blockIdx.x;
  const int count = d_len[b];
  const int lane = kz_lane();
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool w = lane == 0 && wv == 0;
  const u8* src = srcAll + (int64_t)b * stride;
  u8* dst = dstAll + (int64_t)b * stride;
  if (count < LZ_MIN_BLOCK) { if (w) { d_len2[b] = count; d_flag[b] = 0; } return; }      // :313-315 (count==0 handled by caller)
  const int hsize = extra ? (1 << 19) : (1 << 16);
  int32_t* hashes = hashAll + (int64_t)b * hsize;
  for (int i = (int)threadIdx.x; i < hsize; i += 128) hashes[i] = 0;
  if (threadIdx.x == 0) { progress[0] = 0; progress[1] = 0; }
  __syncthreads();
  u8* tkBuf = tmpAll + (int64_t)b * tmpStride;
  u8* mBuf = tkBuf + tmpStride / 3;
  u8* mLenBuf = mBuf + tmpStride / 3;
  LZ_ORDER();
  // the block's "dataType" context entry (:342-353): DNA -> minMatch 6, SMALL_ALPHABET -> not for LZ
  const int dtype = d_dtype ? __builtin_amdgcn_readfirstlane(d_dtype[b]) : 0;
  if (dtype == 2 /* SMALL_ALPHABET */) { if (w) { d_len2[b] = count; d_flag[b] = 0; } return; }
  const int minMatch = (dtype == 1 /* DNA */) ? 6 : 4;
  const int srcEnd = count - 16 - 2;
  const int maxDist = (srcEnd < 4 * LZ_MAXD1) ? LZ_MAXD1 : LZ_MAXD2;
  if (w) dst[12] = (u8)(((maxDist == LZ_MAXD1) ? 0 : 1) | (((minMatch - 2) & 7) << 1));
  int srcIdx = 0, anchor = 0, dstIdx = 13, mIdx = 0, mLenIdx = 0, tkIdx = 0;
  // the reference's token buffer holds max(count / 5, 256) tokens and never grows (LZCodec.java:324-333): one more is an exception
  // that ends the block with ERR_PROCESS_BLOCK (CompressedOutputStream.java:1041-1044) -> d_flag = -1
  const int tkCap = max(count / 5, 256);
  bool tkOver = false;
  int repd0 = count, repd1 = count;
  int repIdx = 0, srcInc = 0;
  bool ok = true;
  // Assist prefetch: the parse itself is one dependent chain (hash entry -> candidate bytes -> decision) with two
  // cache misses per literal position.  Wave 1, which has nothing else to do, touches the hash-table lines and the candidate lines
  // of the positions 64..192 ahead of the parser, whose position it reads from LDS.  Touching lines never changes what the parse reads
  // (values are always loaded at their proper time), it only turns HBM misses into cache hits; what the helper sees of the table may
  // be stale, which costs a hit, never a result.  (Until round 3 the parsing wave did this itself every 64 positions and waited for
  // the three dependent misses each time: 15 % of a text block's parse, measured with KZ_LZ_PROF.)
  if (wv != 0) {
    int pfPos = 0;
    u32 pfSink = 0;
    for (int polls = 0; polls < (1 << 28); polls++) {              // (bounded: a helper never outlives its kernel by design)
      if (__builtin_amdgcn_readfirstlane(progress[1])) break;
      const int pos = __builtin_amdgcn_readfirstlane(progress[0]);
      if (pos + 128 > pfPos) {
        const int q = max(pfPos, pos + 32) + lane;
        if (q < srcEnd) {
          const int rq = hashes[lz_hash(src + q, extra)];
          pfSink ^= (u32)rq;
          if (rq > 0 && rq < q) pfSink ^= (u32)src[rq];
        }
        pfPos = max(pfPos, pos + 32) + 64;
      } else __builtin_amdgcn_s_sleep(16);
    }
    asm volatile("" :: "v"(pfSink));                               // keep the prefetch loads alive
    return;
  }
  int pubPos = 0;                                                  // position last published to the helper
#ifdef KZ_LZ_PROF
  long long lzp[12] = {0}, lzn[12] = {0}; long long lzt = (long long)__builtin_readcyclecounter();
#endif
  while (srcIdx < srcEnd) {
    if (__builtin_expect(srcIdx >= pubPos + 32, 0)) { pubPos = srcIdx; if (w) progress[0] = srcIdx; }
    LZP(0)
    // Literal runs, 64 positions per round trip.  After two literal steps in a row (srcInc >= 2, repIdx == 0) the next
    // positions are known in advance as long as they are literal steps too: p(k+1) = p(k) + 1 + ((srcInc + k) >> 6).
    // Lane k tests position p(k) the way the step below does -- repeat candidates, hash-table candidate -- and the
    // wave consumes the longest prefix of positions that have no 4-byte candidate at all: those are literal steps
    // whatever minMatch is, and all they do is store their position in the table (:368-371).  The first position with
    // a candidate (or past srcEnd) is left to the full step.  A lane's table entry is stale if an earlier lane of the
    // batch has the same hash: its candidate is then that lane's position (match-any on the hash).
    if (srcInc >= 2) {
      const u32 stp = 1u + ((u32)(srcInc + lane) >> 6);
      const u32 incl = kz_wave_incl_sum(stp);
      const int pk = srcIdx + (int)(incl - stp);
      const bool in = pk < srcEnd;
      u64 ownk = 0; int hk = 0, r0 = 0, mr = 0, rA = 0, rB = 0;
      u32 wAk = 0, wBk = 0, cw = 0;
      if (in) {
        ownk = lz_le64(src + pk);
        hk = (int)(((ownk << 24) * LZ_SEED) >> (extra ? 45 : 48));
        r0 = hashes[hk];
        mr = max(pk - maxDist, 0);
        rA = pk + 1 - repd0; rB = pk + 1 - repd1;
        wAk = lz_le32(src + max(rA, 0)); wBk = lz_le32(src + max(rB, 0));
        cw = lz_le32(src + r0);
      }
      const u32 own0k = (u32)ownk, own1k = (u32)(ownk >> 8);
      uint64_t peers = kz_ballot(in);
      const int hbits = extra ? 19 : 16;
      for (int bb = 0; bb < hbits; bb++) {
        const uint64_t m = kz_ballot(((hk >> bb) & 1) != 0);
        peers &= ((hk >> bb) & 1) ? m : ~m;
      }
      const uint64_t earlier = peers & kz_lanemask_lt();
      {                                                              // the latest earlier position with this hash
        // (the shuffles run with every lane active: a lane that is masked off supplies nothing to ds_bpermute)
        const int j = earlier ? 63 - (int)__builtin_clzll(earlier) : lane;
        const int pj = __shfl(pk, j, 64);
        const u32 oj = (u32)__shfl((int)own0k, j, 64);
        if (earlier) { r0 = pj; cw = oj; }
      }
      const bool stop = !in || ((rA > mr) && (wAk == own1k)) || ((rB > mr) && (wBk == own1k)) || ((r0 > mr) && (cw == own0k));
      const uint64_t sm = kz_ballot(stop);
      const int f = sm ? (int)__builtin_ctzll(sm) : 64;
      if (f > 0) {
        const uint64_t consumed = (f == 64) ? ~0ULL : ((1ULL << f) - 1ULL);
        const uint64_t later = peers & consumed & ~(kz_lanemask_lt() | (1ULL << lane));
        if (lane < f && later == 0) hashes[hk] = pk;                 // the highest position of equal hashes wins, as in order
        LZ_ORDER();
        srcIdx = (f < 64) ? __builtin_amdgcn_readlane(pk, f & 63) : (__builtin_amdgcn_readlane(pk, 63) + (int)__builtin_amdgcn_readlane((int)stp, 63));
        srcInc += f;
        LZP(1)
        continue;
      }
      LZP(2)
    }
    int bestLen = 0;
    // every load whose address is known up front is issued before the dependent hash-table access: the two repeat
    // candidates and the current bytes travel together, the chain is then bytes -> table entry -> candidate
    const int srcIdx1 = srcIdx + 1;
    const int minRef = max(srcIdx - maxDist, 0);
    const int refA = srcIdx1 - (repIdx ? repd1 : repd0), refB = srcIdx1 - (repIdx ? repd0 : repd1);
    const u64 own = lz_le64(src + srcIdx);
    const u32 own1 = lz_le32(src + srcIdx1);
    const u32 wA = lz_le32(src + max(refA, 0)), wB = lz_le32(src + max(refB, 0));
    const int h0 = (int)(((own << 24) * LZ_SEED) >> (extra ? 45 : 48));
    // the lazy probes' table entries (positions srcIdx+1, +2: their 5 hashed bytes are inside `own`) are requested
    // together with ref0 and patched below for the stores that come in between
    const int h1e = (int)((((own >> 8) << 24) * LZ_SEED) >> (extra ? 45 : 48));
    const int h2e = (int)((((own >> 16) << 24) * LZ_SEED) >> (extra ? 45 : 48));
    const int ref0 = hashes[h0];
    const int r1e = hashes[h1e];
    const int r2e = extra ? hashes[h2e] : 0;
    LZ_ORDER();
    if (w) hashes[h0] = srcIdx;
    LZP(3)
    int ref = refA;
    if ((ref > minRef) && (wA == own1)) {
      bestLen = lz_find_match(src, srcIdx1, ref, min(srcEnd - srcIdx1, LZ_MAX_MATCH));
    } else {
      ref = refB;
      if ((ref > minRef) && (wB == own1)) bestLen = lz_find_match(src, srcIdx1, ref, min(srcEnd - srcIdx1, LZ_MAX_MATCH));
    }
    LZP(4)
    if (bestLen < minMatch) {
      ref = ref0;
      // the reference first compares 4 bytes at the candidate, then measures the match (:377-381); here the measuring
      // loads go out at once (one round trip instead of two): fewer than 4 common bytes = its 4-byte test failing, and
      // either way a length below minMatch ends in the literal step
      // (LZ only: with LZX's 8 times larger table most candidates are unrelated and the wide loads cost more than they save)
      if (extra) { if ((ref > minRef) && (lz_le32(src + ref) == (u32)own)) bestLen = lz_find_match(src, srcIdx, ref, min(srcEnd - srcIdx, LZ_MAX_MATCH)); }
      else if (ref > minRef) { const int l0 = lz_find_match(src, srcIdx, ref, min(srcEnd - srcIdx, LZ_MAX_MATCH)); if (l0 >= 4) bestLen = l0; }
      LZP(5)
      if (bestLen < minMatch) { srcIdx = srcIdx1 + (srcInc >> 6); srcInc++; repIdx = 0; LZ_ORDER(); continue; }
      if ((ref != srcIdx - repd0) && (ref != srcIdx - repd1)) {
        const int h1 = h1e;
        const int ref1 = (h1 == h0) ? srcIdx : r1e;                 // hashes[h0] = srcIdx was stored since the early read
        LZ_ORDER();
        if (w) hashes[h1] = srcIdx1;
        if ((ref1 > minRef + 1) && !lz_diff4(src, ref1 + bestLen - 3, srcIdx1 + bestLen - 3)) {
          const int bestLen1 = lz_find_match(src, srcIdx1, ref1, min(srcEnd - srcIdx1, LZ_MAX_MATCH));
          if (bestLen1 >= bestLen) { ref = ref1; bestLen = bestLen1; srcIdx = srcIdx1; }
        }
        if (extra) {
          const int srcIdx2 = srcIdx1 + 1;
          const int h2 = h2e;
          LZ_ORDER();
          const int ref2 = (h2 == h1) ? srcIdx1 : ((h2 == h0) ? (srcIdx1 - 1) : r2e);   // stores at h0 and h1 since the early read
          LZ_ORDER();
          if (w) hashes[h2] = srcIdx2;
          if ((ref2 > minRef + 2) && !lz_diff4(src, ref2 + bestLen - 3, srcIdx2 + bestLen - 3)) {
            const int bestLen2 = lz_find_match(src, srcIdx2, ref2, min(srcEnd - srcIdx2, LZ_MAX_MATCH));
            if (bestLen2 >= bestLen) { ref = ref2; bestLen = bestLen2; srcIdx = srcIdx2; }
          }
        }
      }
      LZP(6)
      // backward extension (:527-531): up to 8 bytes per round trip instead of one
      while ((srcIdx > anchor) && (ref > minRef)) {
        int ext;
        if (ref >= 8) {
          const u64 x = lz_le64(src + srcIdx - 8) ^ lz_le64(src + ref - 8);      // byte k-1 back = bits 63-8(k-1)..
          ext = x ? (int)(__builtin_clzll(x) >> 3) : 8;
        } else {                                                     // first 8 bytes of the block: byte by byte
          while ((srcIdx > anchor) && (ref > minRef) && (src[srcIdx - 1] == src[ref - 1])) { bestLen++; ref--; srcIdx--; }
          break;
        }
        ext = min(ext, min(srcIdx - anchor, ref - minRef));
        if (ext == 0) break;
        bestLen += ext; ref -= ext; srcIdx -= ext;
        if (ext < 8) break;
      }
      if (bestLen > LZ_MAX_MATCH) { ref += (bestLen - LZ_MAX_MATCH); srcIdx += (bestLen - LZ_MAX_MATCH); bestLen = LZ_MAX_MATCH; }
      LZP(7)
    } else {
      if ((bestLen >= LZ_MAX_MATCH) || (src[srcIdx] != src[ref - 1])) {
        srcIdx++;
        const int h1 = lz_hash(src + srcIdx, extra);
        LZ_ORDER();
        if (w) hashes[h1] = srcIdx;
      } else { bestLen++; ref--; }
    }
    srcInc = 0;
    // the bytes of the first 64 covered positions (hash fill below) are requested now and used after the match has been written out
    const int nextPos = srcIdx + bestLen;
    const int fillP = srcIdx + 1 + lane;
    const u64 fillRaw = (fillP < nextPos) ? lz_le64(src + fillP) : 0ULL;
    const int dist = srcIdx - ref;
    int token, mLenTh;
    if (dist == repd0) { token = 0x00; mLenTh = 3; }
    else if (dist == repd1) { token = 0x04; mLenTh = 3; }
    else {
      if (w) mBuf[mIdx] = (u8)(dist >> 16);
      const int inc1 = dist >= 65536 ? 1 : 0; mIdx += inc1;
      if (w) mBuf[mIdx] = (u8)(dist >> 8);
      const int inc2 = dist >= 256 ? 1 : 0; mIdx += inc2;
      if (w) mBuf[mIdx] = (u8)dist;
      mIdx++;
      token = (inc1 + inc2 + 1) << 3;
      mLenTh = 7;
    }
    const int mLen = bestLen - minMatch;
    if (mLen >= mLenTh) { token += mLenTh; mLenIdx = lz_emit_length(mLenBuf, mLenIdx, mLen - mLenTh, w); }
    else token += mLen;
    repd1 = repd0; repd0 = dist; repIdx = 1;
    const int litLen = srcIdx - anchor;
    if (tkIdx >= tkCap) { tkOver = true; break; }
    if (litLen == 0) { if (w) tkBuf[tkIdx] = (u8)token; tkIdx++; }
    else {
      if (litLen >= 7) {
        if (litLen >= (1 << 24)) { ok = false; break; }
        if (w) tkBuf[tkIdx] = (u8)((7 << 5) | token);
        tkIdx++;
        dstIdx = lz_emit_length(dst, dstIdx, litLen - 7, w);
      } else { if (w) tkBuf[tkIdx] = (u8)((litLen << 5) | token); tkIdx++; }
      lz_copy(dst + dstIdx, src + anchor, litLen);
      dstIdx += litLen;
    }
    anchor = nextPos;
    LZP(8)
    LZ_ORDER();
    // hash fill of the covered positions (:554-565): position-monotone, last writer = highest position.  Short matches (the usual
    // case) store position by position from lane 0, in order: no question of which lane wins a shared slot.
    const int nfill = anchor - (srcIdx + 1);
    if (nfill <= 16) {
      const int hh0 = (int)(((fillRaw << 24) * LZ_SEED) >> (extra ? 45 : 48));
      for (int k = 0; k < nfill; k++) { const int hk = __builtin_amdgcn_readlane(hh0, k); if (w) hashes[hk] = srcIdx + 1 + k; }
      LZ_ORDER();
    } else
    for (int p0 = srcIdx + 1; p0 < anchor; p0 += 64) {
      const int pp = p0 + lane;
      int hh = 0; bool act = pp < anchor;
      if (p0 == srcIdx + 1) { if (act) hh = (int)(((fillRaw << 24) * LZ_SEED) >> (extra ? 45 : 48)); }
      else if (act) hh = lz_hash(src + pp, extra);
      // within the wave several positions may share a slot: only the highest position may win
      for (uint64_t am = kz_ballot(act) & ~1ULL; am; am &= am - 1) {
        const int l = (int)__builtin_ctzll(am);                    // an active lane above lane 0
        const int hl = __shfl(hh, l, 64);
        if (act && l > lane && hl == hh) act = false;
      }
      if (act) hashes[hh] = pp;
      LZ_ORDER();
    }
    srcIdx = anchor;
    LZP(9)
  }
  if (w) progress[1] = 1;                                          // the helper leaves
#ifdef KZ_LZ_PROF
  if (b == 0 && w) {
    long long tot = 0; for (int k = 0; k < 10; k++) tot += lzp[k];
    for (int k = 0; k < 10; k++) printf("LZPROF sec %d cycles %lld permille %lld n %lld avg %lld\n", k, lzp[k], lzp[k] * 1000 / tot, lzn[k], lzn[k] ? lzp[k] / lzn[k] : 0LL);
    printf("LZPROF total %lld count %d tokens %d   (0 assist, 1 / 2 literal batch taken / missed, 3 own + table, 4 repeat candidates, 5 table candidate, 6 lazy, 7 backward, 8 write-out, 9 hash fill)\n", tot, count, tkIdx);
  }
#endif
  int res = 0, produced = 0;
  if (ok) {
    const int litLen = count - anchor;
    if (dstIdx + litLen + tkIdx + mIdx + mLenIdx >= count) ok = false;                    // :571-572
    else if (tkIdx >= tkCap) tkOver = true;
    else {
      if (litLen >= 7) { if (w) tkBuf[tkIdx] = (u8)(7 << 5); tkIdx++; dstIdx = lz_emit_length(dst, dstIdx, litLen - 7, w); }
      else { if (w) tkBuf[tkIdx] = (u8)(litLen << 5); tkIdx++; }
      lz_copy(dst + dstIdx, src + anchor, litLen);
      dstIdx += litLen;
      if (w) {
        dst[0] = (u8)dstIdx; dst[1] = (u8)(dstIdx >> 8); dst[2] = (u8)(dstIdx >> 16); dst[3] = (u8)(dstIdx >> 24);
        dst[4] = (u8)tkIdx; dst[5] = (u8)(tkIdx >> 8); dst[6] = (u8)(tkIdx >> 16); dst[7] = (u8)(tkIdx >> 24);
        dst[8] = (u8)mIdx; dst[9] = (u8)(mIdx >> 8); dst[10] = (u8)(mIdx >> 16); dst[11] = (u8)(mIdx >> 24);
      }
      LZ_ORDER();
      lz_copy(dst + dstIdx, tkBuf, tkIdx); dstIdx += tkIdx;
      lz_copy(dst + dstIdx, mBuf, mIdx); dstIdx += mIdx;
      lz_copy(dst + dstIdx, mLenBuf, mLenIdx); dstIdx += mLenIdx;
      produced = dstIdx;
      res = (dstIdx <= count - (count / 100)) ? 1 : 0;                                     // :596
    }
  }
  if (w) { d_flag[b] = tkOver ? -1 : ((ok && res) ? 1 : 0); d_len2[b] = (!tkOver && ok && res) ? produced : count; }
}

// a 64-byte register window over a byte stream that is read front to back (tokens, distance bytes): one load per 64 bytes instead
// of a memory round trip per byte on the decoder's dependent chain; positions at or past `count` read as 0 (callers bound first)
__device__ __forceinline__ int lz_win_get(const u8* a, int count, u32& v, int& base, int idx) {
  if ((u32)(idx - base) >= 64u) { base = idx; const int p = idx + kz_lane(); v = (p < count) ? (u32)a[p] : 0u; }
  return __builtin_amdgcn_readlane((int)v, idx - base);
}
// readLength; reads are bounded by the block length (past it the Java code throws or sees stale bytes: failure)
__device__ __forceinline__ int lz_read_length(const u8* a, int& idx, int count, bool& bad) {
  if (idx + 4 > count) {
    int need = 1;
    if (idx < count) need = (a[idx] < 254) ? 1 : (a[idx] == 254 ? 3 : 4);
    if (idx + need > count) { bad = true; return 0; }
  }
  int res = a[idx++];
  if (res < 254) return res;
  if (res == 254) { res += (a[idx] << 8); res += a[idx + 1]; idx += 2; return res; }
  res += (a[idx] << 16); res += (a[idx + 1] << 8); res += a[idx + 2];
  idx += 3;
  return res;
}

__global__ __launch_bounds__(64) void k_lz_inv(const u8* __restrict__ srcAll, u8* __restrict__ dstAll, int64_t stride,
                                                const int32_t* __restrict__ d_len, int32_t* __restrict__ d_len2, int32_t* __restrict__ d_flag, int dstCap) {
  const int b = blockIdx.x;
  const int count = d_len[b];
  const int lane = kz_lane();
  const u8* src = srcAll + (int64_t)b * stride;
  u8* dst = dstAll + (int64_t)b * stride;
  bool ok = true;
  int dstIdx = 0;
  if (count == 0) { if (lane == 0) { d_len2[b] = 0; d_flag[b] = 1; } return; }
  if (count < 13) ok = false;
  if (ok) {
    const int dstEnd = dstCap;
    const int tkLen = (int)lz_le32(src), mIdxLen = (int)lz_le32(src + 4), mLenLen = (int)lz_le32(src + 8);
    if ((tkLen < 0) || (mIdxLen < 0) || (mLenLen < 0)) ok = false;
    else if ((tkLen < 13) || (tkLen > count) || (mIdxLen > count - tkLen) || (mLenLen > count - tkLen - mIdxLen)) ok = false;
    if (ok) {
      int tkIdx = tkLen, mIdx = tkIdx + mIdxLen, mLenIdx = mIdx + mLenLen;
      const int srcEnd = tkIdx - 13, litEnd = tkIdx;
      const int maxDist = ((src[12] & 1) == 0) ? LZ_MAXD1 : LZ_MAXD2;
      const int minMatch = ((src[12] >> 1) & 7) + 2;
      int srcIdx = 13, repd0 = count, repd1 = count;
      bool bad = false;
      u32 tkWin = 0, mWin = 0; int tkBase = -64, mBase = -64;          // register windows over the token and distance streams
      for (;;) {
        if (tkIdx >= count) { ok = false; break; }
        const int token = lz_win_get(src, count, tkWin, tkBase, tkIdx); tkIdx++;
        if (token >= 32) {
          const int litLen = (token >= 0xE0) ? 7 + lz_read_length(src, srcIdx, count, bad) : token >> 5;
          if (bad) { ok = false; break; }
          if ((litLen > dstEnd - dstIdx) || (litLen > litEnd - srcIdx)) { ok = false; break; }
          lz_copy(dst + dstIdx, src + srcIdx, litLen);
          srcIdx += litLen; dstIdx += litLen;
          if (srcIdx >= srcEnd) { ok = (srcIdx == srcEnd + 13); break; }
        }
        int mLen, dist;
        const int f = token & 0x18;
        if (f == 0) {
          mLen = token & 3;
          mLen += (mLen == 3) ? minMatch + lz_read_length(src, mLenIdx, count, bad) : minMatch;
          dist = ((token & 4) == 0) ? repd0 : repd1;
        } else {
          mLen = token & 7;
          mLen += (mLen == 7) ? minMatch + lz_read_length(src, mLenIdx, count, bad) : minMatch;
          if (mIdx + ((f == 0x18) ? 3 : (f == 0x10) ? 2 : 1) > count) { ok = false; break; }
          dist = lz_win_get(src, count, mWin, mBase, mIdx); mIdx++;
          if (f == 0x18) { dist = (dist << 8) | lz_win_get(src, count, mWin, mBase, mIdx); dist = (dist << 8) | lz_win_get(src, count, mWin, mBase, mIdx + 1); mIdx += 2; }
          else if (f == 0x10) { dist = (dist << 8) | lz_win_get(src, count, mWin, mBase, mIdx); mIdx++; }
        }
        if (bad) { ok = false; break; }
        repd1 = repd0; repd0 = dist;
        const int mEnd = dstIdx + mLen;
        const int ref = dstIdx - dist;
        if ((ref < 0) || (dist > maxDist) || (mEnd > dstEnd) || dist <= 0) { ok = false; break; }
        // the preceding stores of this wave are visible to the copy below in program order (one wave per block: the wavefront-scope
        // fence of LZ_ORDER says it; a workgroup-scope fence here drained every outstanding store once per token)
        LZ_ORDER();
        if (dist >= 64) {
          for (int k = 0; k < mLen; k += 64) { const int i = k + lane; u8 v = 0; if (i < mLen) v = dst[ref + i]; LZ_ORDER(); if (i < mLen) dst[dstIdx + i] = v; LZ_ORDER(); }
        } else {
          // overlapping copy: the output is periodic with period dist over already written bytes
          for (int i = lane; i < mLen; i += 64) dst[dstIdx + i] = dst[ref + (i % dist)];
        }
        LZ_ORDER();
        dstIdx = mEnd;
      }
    }
  }
  if (lane == 0) { d_flag[b] = ok ? 1 : 0; d_len2[b] = ok ? dstIdx : 0; }
}

size_t kz_lz_scratch(int B, int maxN) {
  const size_t tmpStride = kz_align((size_t)maxN * 3 + 3072, 256);
  return (size_t)B * (((size_t)1 << 19) * 4 + tmpStride) + 16384;
}

int kz_stage_lz_forward(kz_ctx* ctx, kz_batch& bt, int extra) {
  const int B = bt.B;
  int maxN = 0;
  for (int b = 0; b < B; b++) if (bt.h_len[b] > maxN) maxN = bt.h_len[b];
  const int hsize = extra ? (1 << 19) : (1 << 16);
  int32_t* hashes = (int32_t*)kz_arena_alloc(ctx, (size_t)B * hsize * 4);
  const int64_t tmpStride = (int64_t)kz_align((size_t)maxN * 3 + 3072, 256) / 3 * 3;
  u8* tmp = (u8*)kz_arena_alloc(ctx, (size_t)tmpStride * B + 256);
  if (!hashes || !tmp) { snprintf(ctx->err, sizeof(ctx->err), "lz_forward: arena overflow"); return -KZ_ERR_DEVICE; }
  KZ_LAUNCH(ctx, KID_LZ_FWD, k_lz_fwd, dim3(B), dim3(128), bt.buf[bt.cur], bt.buf[bt.cur ^ 1], bt.stride, bt.d_len, bt.d_len2, bt.d_flag,
            hashes, tmp, tmpStride, extra, bt.d_dtype);
  KZ_HIP(hipGetLastError());
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}

int kz_stage_lz_inverse(kz_ctx* ctx, kz_batch& bt, int extra, int dstCap) {
  (void)extra;
  const int B = bt.B;
  if ((int64_t)dstCap > bt.stride) dstCap = (int)bt.stride;
  KZ_LAUNCH(ctx, KID_LZ_INV, k_lz_inv, dim3(B), dim3(64), bt.buf[bt.cur], bt.buf[bt.cur ^ 1], bt.stride, bt.d_len, bt.d_len2, bt.d_flag, dstCap);
  KZ_HIP(hipGetLastError());
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}
