// kz_ans.hip -- order-0 range ANS (ANS0) chunk encoder / decoder for gfx950.
//
// Replaces K/entropy/ANSRangeEncoder.java:263-305 (encode), :419-449 (rebuildStatistics),
// :171-200 (updateFrequencies), :211-252 (encodeHeader), :337-407 (encodeChunk),
// :315-328 (encodeSymbol), :473-496 (Symbol.reset); K/entropy/EntropyUtils.java:141-250
// (normalizeFrequencies), :38-75 (encodeAlphabet), :259-276 (varint);
// K/entropy/ANSRangeDecoder.java:189-236, :357-440 (decodeChunkV2), :452-544 (decodeHeader).
//
// Statistics reset every 16 KiB chunk (ANSRangeEncoder.java:39), so a block is ~256 independent
// chunks: one wave64 per chunk.  Histogram: LDS 256 bins, ballot-aggregated.  Normalisation: the
// reference loop restated data-parallel (4 symbols per lane, wave reductions / ballot prefixes give
// the same sequential semantics).  rANS: 4 interleaved states = 4 lanes; the shared output cursor
// of encodeSymbol (:315-321) is reproduced with a 4-lane ballot prefix.  Chunk bit strings are
// concatenated at bit granularity by a scan + funnel-shift kernel (the .knz payload is bit packed).
#include "kz_device.h"
#include "kz_internal.h"
#include "kz_chunk.h"

typedef uint16_t u16;

#define ANS_TOP (1u << 15)
#define ANS_LR 12


// ---- single-lane MSB-first bit writer into a zeroed buffer ------------------------------------
struct BitW { u8* p; u32 pos; };
__device__ __forceinline__ void bw_put(BitW& w, u32 v, int count) {
  // low `count` bits of v, MSB first (DefaultOutputBitStream.java:103-123)
  while (count > 0) {
    const int bitoff = w.pos & 7, room = 8 - bitoff;
    const int take = count < room ? count : room;
    const u32 bits = (v >> (count - take)) & ((1u << take) - 1u);
    w.p[w.pos >> 3] |= (u8)(bits << (room - take));
    w.pos += take; count -= take;
  }
}

typedef u16 __attribute__((aligned(1))) ans_u16_unaligned;
typedef u64 __attribute__((aligned(1))) ans_u64_unaligned;

// =================================================================================================
// encode: one wave per chunk
__global__ __launch_bounds__(64) void k_ans_enc_chunk(const u8* __restrict__ src, int64_t stride,
                                                       const int32_t* __restrict__ d_len, AnsEnc E) {
  const int b = blockIdx.y, ck = blockIdx.x;
  const int count = d_len[b];
  const int lane = kz_lane();
  const int64_t ci = (int64_t)b * E.C + ck;
  const u8* blk = src + (int64_t)b * stride;
  u8* hdr = E.hdr + ci * ANS_HDR_BYTES;
  u8* scr = E.scr + ci * ANS_SCRATCH;
  if (count <= 32) {                                            // ANSRangeEncoder.java:267-270 raw
    if (ck != 0) return;
    if (lane < count) scr[lane] = blk[lane];
    if (lane == 0) { E.hdrBits[ci] = 0; E.tailOff[ci] = 0; E.tailBits[ci] = 8u * (u32)count; }
    return;
  }
  const int start = ck * ANS_CHUNK;
  if (start >= count) return;
  const int end = min(count, start + ANS_CHUNK);
  const int len = end - start;

  __shared__ u32 hist[256];
  __shared__ u16 nfreq[256];
  __shared__ u8 alpha[256];
  __shared__ uint4 symTab[256];                                 // {xmax, reciprocal, bias, cmpl | shift << 16}: one 16-byte LDS read per symbol
  // the chunk's symbols are read from global memory twice (histogram, then the coding loop: 4 consecutive bytes per
  // step, L1 hits): a 16 KiB LDS copy limited the kernel to 6 waves per CU, and the 4-lane coding loop needs many
  // waves per SIMD to fill the issue slots
  const u8* data = blk + start;
  __shared__ u32 hbuf[ANS_HDR_BYTES / 4];                       // header bits are assembled in LDS

  for (int i = lane; i < 256; i += 64) hist[i] = 0;
  for (int i = lane; i < ANS_HDR_BYTES / 4; i += 64) hbuf[i] = 0;
  __syncthreads();
  // ---- stage chunk into LDS + histogram (Global.computeHistogramOrder0, K/Global.java:274-322) ----
  for (int i = lane * 4; i < len; i += 256) {
    u32 w = 0; int nb = min(4, len - i);
    if (nb == 4 && ((start + i) & 3) == 0) w = *(const u32*)(blk + start + i);
    else for (int k = 0; k < nb; k++) w |= (u32)blk[start + i + k] << (8 * k);
    for (int k = 0; k < 4; k++) {
      const bool valid = k < nb;
      const u32 c = (w >> (8 * k)) & 0xFF;
      const uint64_t peers = kz_match8(c, valid);
      if (valid && (peers & kz_lanemask_lt()) == 0) atomicAdd(&hist[c], (u32)__popcll(peers));
    }
  }
  __syncthreads();

  // ---- normalizeFrequencies(freqs, alphabet, total, 4096) (EntropyUtils.java:141-250) ----
  // symbol s = q*64 + lane, q = 0..3 (symbol order = (q, lane) lexicographic)
  const u32 total = (u32)len, scale = 1u << ANS_LR;
  u32 f[4]; bool present[4];
  u32 alphabetSize = 0;
  u32 apos[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    f[q] = hist[q * 64 + lane];
    present[q] = f[q] != 0;
    const uint64_t bal = kz_ballot(present[q]);
    apos[q] = alphabetSize + (u32)__popcll(bal & kz_lanemask_lt());
    alphabetSize += (u32)__popcll(bal);
    if (present[q]) alpha[apos[q]] = (u8)(q * 64 + lane);
  }
  if (total != scale) {                                          // :155-162 shortcut otherwise
    u32 sumScaled = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (present[q]) {
        const u32 sf = f[q] * scale;                             // <= 2^26, fits
        f[q] = (sf <= total) ? 1u : (sf + (total >> 1)) / total;
      }
      sumScaled += kz_wave_sum(present[q] ? f[q] : 0);
    }
    if (alphabetSize == 1) {
#pragma unroll
      for (int q = 0; q < 4; q++) if (present[q]) f[q] = scale;
    } else if (sumScaled != scale) {
      // idxMax = first symbol (in symbol order) holding the maximum scaled frequency (:184-185)
      u32 best = 0;
#pragma unroll
      for (int q = 0; q < 4; q++) { u32 v = present[q] ? f[q] : 0; best = max(best, v); }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) best = max(best, (u32)__shfl_xor(best, d, 64));
      int idxMax = 256;
#pragma unroll
      for (int q = 3; q >= 0; q--) {
        const uint64_t bal = kz_ballot(present[q] && f[q] == best);
        if (bal) idxMax = q * 64 + (int)__builtin_ctzll(bal);
      }
      int delta = (int)sumScaled - (int)scale;
      const int errThr = (int)(best >> 4);
      const int mq = idxMax >> 6, ml = idxMax & 63;
      int adjMax = 0;                                            // signed adjustment applied to freqs[idxMax]
      const int ad = delta < 0 ? -delta : delta;
      if (ad <= errThr) {
        adjMax = -delta;                                          // :204-208 fast path
      } else {
        if (delta < 0) { delta += errThr; adjMax = errThr; } else { delta -= errThr; adjMax = -errThr; }
        // apply the first part now: the slow path tests freqs[idx] <= 2 on the updated value
#pragma unroll
        for (int q = 0; q < 4; q++) if (q == mq && lane == ml) f[q] = (u32)((int)f[q] + adjMax);
        adjMax = 0;
        const int inc = (delta > 0) ? -1 : 1;                     // :219-246
        delta = delta < 0 ? -delta : delta;
        int round = 0;
        while ((++round < 6) && (delta > 0)) {
          int adjustments = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const bool elig = present[q] && f[q] > 2;
            const uint64_t bal = kz_ballot(elig);
            const int pre = (int)__popcll(bal & kz_lanemask_lt());
            const int tot = (int)__popcll(bal);
            if (elig && pre < delta) f[q] = (u32)((int)f[q] + inc);
            const int used = tot < delta ? tot : delta;
            adjustments += used; delta -= used;
          }
          if (adjustments == 0) break;
        }
        // freqs[idxMax] = max(freqs[idxMax] - delta, 1)  (:248)
#pragma unroll
        for (int q = 0; q < 4; q++) if (q == mq && lane == ml) { int v = (int)f[q] - delta; f[q] = (u32)(v > 1 ? v : 1); }
      }
      if (adjMax != 0) {
#pragma unroll
        for (int q = 0; q < 4; q++) if (q == mq && lane == ml) f[q] = (u32)((int)f[q] + adjMax);
      }
    }
  }
  // ---- symbol tables (Symbol.reset :473-496) ----
  u32 cum = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const u32 fv = present[q] ? f[q] : 0;
    const u32 inc = kz_wave_incl_sum(fv);
    const u32 cumFreq = cum + inc - fv;
    cum += __shfl(inc, 63, 64);
    const int s = q * 64 + lane;
    nfreq[s] = (u16)fv;
    if (present[q]) {
      u32 freq = fv;
      if (freq >= scale) freq = scale - 1;
      const u32 xmaxv = ((ANS_TOP >> ANS_LR) << 16) * freq;
      const u32 cmplv = scale - freq;
      if (freq < 2) symTab[s] = make_uint4(xmaxv, 0xFFFFFFFFu, cumFreq + scale - 1, cmplv | (32u << 16));
      else {
        int shift = 0;
        while (freq > (1u << shift)) shift++;
        symTab[s] = make_uint4(xmaxv, (u32)((((1ULL << (shift + 31)) + freq - 1) / freq) & 0xFFFFFFFFULL), cumFreq, cmplv | ((u32)(32 + shift - 1) << 16));
      }
    }
  }
  __syncthreads();
  // ---- header (updateFrequencies :174 + encodeHeader :211-252), lane 0 ----
  u32 hb = 0;
  if (lane == 0) {
    BitW w{(u8*)hbuf, 0};
    bw_put(w, ANS_LR - 8, 3);
    if (alphabetSize == 256) { bw_put(w, 0, 1); bw_put(w, 0, 1); }       // FULL_ALPHABET, ALPHABET_256
    else if (alphabetSize == 0) { bw_put(w, 0, 1); bw_put(w, 1, 1); }
    else {
      bw_put(w, 1, 1);
      const int lastMask = alpha[alphabetSize - 1] >> 3;
      bw_put(w, (u32)lastMask, 5);
      for (int i = 0; i <= lastMask; i++) {
        u32 m = 0;
        for (int j = 0; j < 8; j++) if (nfreq[i * 8 + j] != 0) m |= 1u << j;
        bw_put(w, m, 8);
      }
    }
    if (alphabetSize > 1) {
      const int chkSize = (alphabetSize >= 64) ? 8 : 6;
      int llr = 3;
      while ((1 << llr) <= ANS_LR) llr++;
      for (int i = 1; i < (int)alphabetSize; i += chkSize) {
        const int endj = (i + chkSize < (int)alphabetSize) ? i + chkSize : (int)alphabetSize;
        int mx = (int)nfreq[alpha[i]] - 1;
        for (int j = i + 1; j < endj; j++) { int v = (int)nfreq[alpha[j]] - 1; if (v > mx) mx = v; }
        int logMax = 0;
        while ((1 << logMax) <= mx) logMax++;
        bw_put(w, (u32)logMax, llr);
        if (logMax == 0) continue;
        for (int j = i; j < endj; j++) bw_put(w, (u32)nfreq[alpha[j]] - 1u, logMax);
      }
    }
    hb = w.pos;
    E.hdrBits[ci] = hb;
  }
  __syncthreads();
  for (int i = lane; i < ANS_HDR_BYTES / 4; i += 64) ((u32*)hdr)[i] = hbuf[i];
  if (alphabetSize <= 1) {                                        // :295-298 no payload
    if (lane == 0) { E.tailOff[ci] = 0; E.tailBits[ci] = 0; }
    return;
  }
  // ---- encodeChunk (:337-407): 4 lanes = st0..st3, walking backwards ----
  const int bufLen = ANS_SCRATCH;
  const int end4 = len & -4;                                       // chunk-relative
  int n = bufLen - 1;
  if (lane == 0) for (int i = len - 1; i >= end4; i--) scr[n - (len - 1 - i)] = data[i];
  n -= (len - end4);
  u32 st = ANS_TOP;
  int idx = n;
  if (lane < 4) {
    // the step for source dword w (bytes 4w .. 4w+3; lane l codes byte 4w + 3 - l = data[i - l] for i = 4w + 3)
#define ANS_ENC_STEP(WD)                                                                                      \
    { const u32 c = ((WD) >> (8 * (3 - lane))) & 0xFFu;                                                       \
      const uint4 sy = symTab[c];                                                                             \
      const bool x = st >= sy.x;                                    /* (int) compare: both < 2^31 */           \
      const uint64_t bal = kz_ballot(x) & 0xFULL;                                                             \
      const int pre = (int)__popcll(bal & kz_lanemask_lt());                                                  \
      if (x) {                                                      /* scr[e] = low byte, scr[e - 1] = next byte: one 2-byte store */ \
        const int e = idx - 2 * pre;                                                                          \
        *(ans_u16_unaligned*)(scr + e - 1) = (u16)(((st & 0xFFu) << 8) | ((st >> 8) & 0xFFu));                \
        st >>= 16;                                                                                            \
      }                                                                                                       \
      idx -= 2 * (int)__popcll(bal);                                                                          \
      const u32 q = (u32)(((u64)st * (u64)sy.y) >> (sy.w >> 16));                                             \
      st = st + sy.z + q * (sy.w & 0xFFFFu); }
    // the source is read 16 bytes at a time (chunks start 16-byte aligned), the next group requested while this one is coded:
    // the byte loads used to sit on the loop's dependent chain (one L1 round trip per step)
    const int T = end4 >> 2;                                         // dwords to code, from T - 1 down to 0
    const uint4* d16 = (const uint4*)data;
    int g = (T - 1) >> 2;
    uint4 cur = (T > 0) ? d16[g] : make_uint4(0, 0, 0, 0);
    for (; g >= 0; g--) {
      const uint4 nxt = (g > 0) ? d16[g - 1] : make_uint4(0, 0, 0, 0);
      const int top = min(3, T - 1 - 4 * g);                         // the highest group may be partial
      if (top >= 3) ANS_ENC_STEP(cur.w)
      if (top >= 2) ANS_ENC_STEP(cur.z)
      if (top >= 1) ANS_ENC_STEP(cur.y)
      ANS_ENC_STEP(cur.x)
      cur = nxt;
    }
#undef ANS_ENC_STEP
  }
  idx = __shfl(idx, 0, 64);
  n = idx + 1;
  const u32 payload = (u32)(bufLen - n);
  // varint(payload) | st0..st3 (32 bits each, big endian) | payload   -- contiguous before scr[n]
  u32 vlen = 1; { u32 v = payload; while (v >= 128) { v >>= 7; vlen++; } }
  const int tail = n - 16 - (int)vlen;
  if (lane == 0) {
    u32 v = payload; int p = tail;
    while (v >= 128) { scr[p++] = (u8)(0x80 | (v & 0x7F)); v >>= 7; }     // EntropyUtils.java:259-276
    scr[p++] = (u8)v;
    E.tailOff[ci] = (u32)tail; E.tailBits[ci] = 8u * (payload + 16 + vlen);
  }
  if (lane < 4) {
    u8* p = scr + tail + vlen + 4 * lane;
    p[0] = (u8)(st >> 24); p[1] = (u8)(st >> 16); p[2] = (u8)(st >> 8); p[3] = (u8)st;
  }
}

// per block: exclusive scan of chunk bit lengths (<= 257 chunks) -> bitOff, total payload bits
__global__ __launch_bounds__(64) void k_ans_enc_scan(const int32_t* __restrict__ d_len, AnsEnc E, int64_t* __restrict__ d_bits, int rawLimit) {
  const int b = blockIdx.x;
  const int count = d_len[b];
  const int chunks = (count <= rawLimit) ? (count > 0 ? 1 : 0) : (count + ANS_CHUNK - 1) / ANS_CHUNK;
  const int lane = kz_lane();
  u64 carry = 0;
  for (int base = 0; base < chunks; base += 64) {
    const int c = base + lane;
    const int64_t ci = (int64_t)b * E.C + c;
    const u32 v = (c < chunks) ? E.hdrBits[ci] + E.tailBits[ci] : 0;
    const u32 inc = kz_wave_incl_sum(v);
    if (c < chunks) E.bitOff[ci] = carry + inc - v;
    carry += __shfl(inc, 63, 64);
  }
  if (lane == 0) d_bits[b] = (int64_t)carry;
}

// 32 bits (MSB first) of the bit string p[0..nbits) starting at bit r; zero outside
__device__ __forceinline__ u32 kz_fetch32(const u8* __restrict__ p, int64_t nbits, int64_t r) {
  if (r >= nbits || r + 32 <= 0) return 0;
  u64 acc = 0;
  const int64_t byte0 = (r >= 0) ? (r >> 3) : -((-r + 7) >> 3);
  const int64_t nbytes = (nbits + 7) >> 3;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const int64_t bi = byte0 + k;
    const u64 v = (bi >= 0 && bi < nbytes) ? p[bi] : 0;
    acc = (acc << 8) | v;
  }
  const int sh = (int)(r - byte0 * 8);                    // 0..7
  u32 w = (u32)((acc << sh) >> 8);
  // mask bits beyond nbits
  const int64_t over = r + 32 - nbits;
  if (over > 0) w &= (over >= 32) ? 0u : (0xFFFFFFFFu << over);
  if (r < 0) { const int64_t under = -r; w &= (under >= 32) ? 0u : (0xFFFFFFFFu >> under); }
  return w;
}

// concatenate chunk bit strings into out[b] at bit offset 8*hdrBytes[b] + bitOff (out pre-zeroed)
__global__ __launch_bounds__(KZ_WG) void k_ans_enc_concat(const int32_t* __restrict__ d_len, AnsEnc E, u8* __restrict__ out,
                                                           int64_t outStride, const int32_t* __restrict__ d_hdrBytes, int rawLimit) {
  const int b = blockIdx.y, ck = blockIdx.x;
  const int count = d_len[b];
  const int chunks = (count <= rawLimit) ? (count > 0 ? 1 : 0) : (count + ANS_CHUNK - 1) / ANS_CHUNK;
  if (ck >= chunks) return;
  const int64_t ci = (int64_t)b * E.C + ck;
  const u8* hdr = E.hdr + ci * ANS_HDR_BYTES;
  const int64_t hb = E.hdrBits[ci];
  const int64_t tb = (int64_t)E.tailBits[ci];
  const u8* tail = E.scr + ci * ANS_SCRATCH + E.tailOff[ci];
  const int64_t len = hb + tb;
  if (len == 0) return;
  const int64_t base = 8LL * d_hdrBytes[b] + (int64_t)E.bitOff[ci];
  u32* o = (u32*)(out + (int64_t)b * outStride);
  const int64_t w0 = base >> 5, w1 = (base + len - 1) >> 5;
  for (int64_t w = w0 + threadIdx.x; w <= w1; w += KZ_WG) {
    const int64_t r = w * 32 - base;                       // chunk-relative bit of this word's first bit
    u32 v = kz_fetch32(hdr, hb, r) | kz_fetch32(tail, tb, r - hb);
    v = __builtin_bswap32(v);
    if (r >= 0 && r + 32 <= len) o[w] = v; else atomicOr(&o[w], v);
  }
}

size_t kz_ans_scratch(int B, int maxN) {
  const int C = (maxN + 64) / ANS_CHUNK + 2;
  return (size_t)B * C * (ANS_HDR_BYTES + ANS_SCRATCH + 32) + (size_t)B * 64 + 8192;
}

int kz_chunk_enc_alloc(kz_ctx* ctx, kz_batch& bt, AnsEnc& E, int* chunksOut) {
  const int B = bt.B;
  int maxN = 0;
  for (int b = 0; b < B; b++) if (bt.h_len[b] > maxN) maxN = bt.h_len[b];
  E.C = (maxN + ANS_CHUNK - 1) / ANS_CHUNK + 1;
  E.hdr = (u8*)kz_arena_alloc(ctx, (size_t)B * E.C * ANS_HDR_BYTES);
  E.scr = (u8*)kz_arena_alloc(ctx, (size_t)B * E.C * ANS_SCRATCH);
  E.hdrBits = (u32*)kz_arena_alloc(ctx, (size_t)B * E.C * 4);
  E.tailOff = (u32*)kz_arena_alloc(ctx, (size_t)B * E.C * 4);
  E.tailBits = (u32*)kz_arena_alloc(ctx, (size_t)B * E.C * 4);
  E.bitOff = (u64*)kz_arena_alloc(ctx, (size_t)B * E.C * 8);
  if (!E.bitOff || !E.scr) { snprintf(ctx->err, sizeof(ctx->err), "entropy encode: arena overflow"); return -KZ_ERR_DEVICE; }
  KZ_HIP(hipMemsetAsync(E.hdrBits, 0, (size_t)B * E.C * 4, ctx->stream));
  KZ_HIP(hipMemsetAsync(E.tailBits, 0, (size_t)B * E.C * 4, ctx->stream));
  *chunksOut = (maxN + ANS_CHUNK - 1) / ANS_CHUNK;
  return 0;
}
int kz_chunk_enc_finish(kz_ctx* ctx, kz_batch& bt, AnsEnc& E, int chunks, uint8_t* out, int64_t outStride,
                        const int32_t* d_hdrBytes, int64_t* d_bits, int rawLimit) {
  const int B = bt.B;
  KZ_LAUNCH(ctx, KID_ANS_ENC_SCAN, k_ans_enc_scan, dim3(B), dim3(64), bt.d_len, E, d_bits, rawLimit);
  if (chunks > 0) KZ_LAUNCH(ctx, KID_ANS_ENC_CONCAT, k_ans_enc_concat, dim3(chunks, B), dim3(KZ_WG), bt.d_len, E, out, outStride, d_hdrBytes, rawLimit);
  KZ_HIP(hipGetLastError());
  return 0;
}

int kz_stage_ans0_encode(kz_ctx* ctx, kz_batch& bt, uint8_t* out, int64_t outStride, const int32_t* d_hdrBytes, int64_t* d_bits) {
  AnsEnc E; int chunks = 0;
  int rc = kz_chunk_enc_alloc(ctx, bt, E, &chunks);
  if (rc) return rc;
  if (chunks > 0) KZ_LAUNCH(ctx, KID_ANS_ENC_CHUNK, k_ans_enc_chunk, dim3(chunks, bt.B), dim3(64), bt.buf[bt.cur], bt.stride, bt.d_len, E);
  return kz_chunk_enc_finish(ctx, bt, E, chunks, out, outStride, d_hdrBytes, d_bits, 32);
}

// =================================================================================================
// decode
// What ends a block's decode early, keyed so that the event of the lowest chunk wins (the reference decodes chunks in
// order and stops at the first one, ANSRangeDecoder.java:210-233): key = chunk*4 + kind.
//   ANS_EV_FAIL  header rejected / bits exhausted / empty alphabet -> decode() != count -> ERR_PROCESS_BLOCK
//   ANS_EV_SKIP  chunk size >= MAX_CHUNK_SIZE: decodeChunkV2 returns false before writing (:360-361) -> decode() breaks, returns count
//   ANS_EV_STOP  consumed bytes != chunk size after the chunk was written (:439)              -> decode() breaks, returns count
// After SKIP/STOP the reference hands the caller `count` bytes whose tail it never wrote (stale buffer contents there);
// here that tail is zero-filled, as the oracle does.
#define ANS_EV_NONE 0x7FFFFFFF
#define ANS_EV_FAIL 0
#define ANS_EV_SKIP 1
#define ANS_EV_STOP 2
#define ANS_MAX_CHUNK_SIZE (1u << 27)
struct AnsDec {
  u64* chunkBit;     // [B][C] absolute bit offset (in the block stream) of each chunk header
  int32_t* event;    // [B] ANS_EV_NONE or chunk*4+kind
  int32_t* nIdx;     // [B] chunks whose chunkBit is valid
  int32_t* hdrOnly;  // [B] chunk whose header must still be validated although its payload is skipped (ANS_EV_SKIP), or -1
  int C;
};

__device__ __forceinline__ u32 kz_peek(const u8* __restrict__ p, u64 pos, int count) {   // count <= 25
  const u64 by = pos >> 3;
  u32 acc = ((u32)p[by] << 24) | ((u32)p[by + 1] << 16) | ((u32)p[by + 2] << 8) | (u32)p[by + 3];
  acc <<= (pos & 7);
  return count ? (acc >> (32 - count)) : 0;
}

// index pass: one lane per block walks the chunk headers (sizes are only known by parsing)
__global__ void k_ans_dec_index(const u8* __restrict__ in, int64_t inStride, const int64_t* __restrict__ d_bitOff,
                                const int64_t* __restrict__ d_bitEnd, const int32_t* __restrict__ d_len, AnsDec D, int B, long long* __restrict__ endOut) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int count = d_len[b];
  const u8* p = in + (int64_t)b * inStride;
  u64 pos = (u64)d_bitOff[b];
  const u64 endBits = (u64)d_bitEnd[b];
  int event = ANS_EV_NONE, nIdx = 0, bufLen = 0, hdrOnly = -1;
  if (count > 32) {
    const int chunks = (count + ANS_CHUNK - 1) / ANS_CHUNK;
    for (int c = 0; c < chunks; c++) {
      D.chunkBit[(int64_t)b * D.C + c] = pos;
      if (pos + 5 > endBits) { event = c * 4 + ANS_EV_FAIL; break; }
      const int lr = 8 + (int)kz_peek(p, pos, 3); pos += 3;
      int asz;
      if (kz_peek(p, pos, 1) == 0) { asz = (kz_peek(p, pos + 1, 1) == 1) ? 0 : 256; pos += 2; }
      else {
        const int lastMask = (int)kz_peek(p, pos + 1, 5); pos += 6;
        asz = 0;
        for (int i = 0; i <= lastMask; i++) { asz += __popc(kz_peek(p, pos, 8)); pos += 8; }
      }
      if (asz == 0) { event = c * 4 + ANS_EV_FAIL; break; }          // ANSRangeDecoder.java:214-215 returns startChunk != count
      {
        const int chkSize = (asz >= 64) ? 8 : 6;
        int llr = 3;
        while ((1 << llr) <= lr) llr++;
        for (int i = 1; i < asz; i += chkSize) {
          const int logMax = (int)kz_peek(p, pos, llr); pos += llr;
          const int endj = (i + chkSize < asz) ? i + chkSize : asz;
          pos += (u64)logMax * (u64)(endj - i);
        }
      }
      if (pos > endBits) { event = c * 4 + ANS_EV_FAIL; break; }
      nIdx = c + 1;
      if (asz == 1) continue;
      // varint (EntropyUtils.java:284-300)
      u32 v = kz_peek(p, pos, 8); pos += 8;
      u32 sz = v & 0x7F; int shift = 7;
      while (v >= 128) { v = kz_peek(p, pos, 8); pos += 8; sz |= (v & 0x7F) << shift; if (shift == 28) break; shift += 7; }
      // :360-361 -- decodeHeader ran before this and throws on a bad table (:213): the chunk kernel still checks it
      if (sz >= ANS_MAX_CHUNK_SIZE) { hdrOnly = c; event = c * 4 + ANS_EV_SKIP; break; }
      const int clen = min(count - c * ANS_CHUNK, ANS_CHUNK);
      bufLen = max(bufLen, max(2 * clen, 256));                       // this.buffer only grows (:371-374)
      if (sz > (u32)bufLen) { nIdx = c; event = c * 4 + ANS_EV_FAIL; break; }          // readBits past the array end throws (:379)
      pos += 128 + 8ULL * sz;
      if (pos > endBits) { nIdx = c; event = c * 4 + ANS_EV_FAIL; break; }
    }
  } else {                                                          // :193-196 bulk read of the raw bytes: past the block's bits it throws
    pos += 8ULL * (u64)(count > 0 ? count : 0);
    if (pos > endBits) event = ANS_EV_FAIL;                          // (chunk 0)
  }
  if (endOut) endOut[b] = (long long)pos;                          // bits consumed (EntropyDecoder contract)
  D.event[b] = event;
  D.nIdx[b] = nIdx;
  D.hdrOnly[b] = hdrOnly;
}

// freq2sym (the reverse mapping of ANSRangeDecoder.java:529-538) by SLOT: lane L fills slots [64 L, 64 L + 64) of the 2^lr <= 4096,
// walking the cumulative table from the symbol that owns its first slot (binary search: the last symbol whose cumulative frequency is
// <= the slot; symbols of frequency 0 share their successor's value and are stepped over).  By symbol -- every lane writing the run
// of its symbol -- the wave waits for the most frequent symbol, a third of the table behind ZRLT: ~1 500 iterations per chunk.
__device__ __forceinline__ void ans_fill_f2s(const u16* __restrict__ cumf, u8* __restrict__ f2s, int scale, int lane) {
  const int x0 = lane * 64;
  if (x0 >= scale) return;
  int lo = 0, hi = 256;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)cumf[mid] <= x0) lo = mid; else hi = mid; }
  int sy = lo;
  int nextAt = (sy + 1 < 256) ? (int)cumf[sy + 1] : 0x7FFFFFFF;
  for (int x = x0; x < x0 + 64; x += 4) {                          // (scale is a power of two >= 256: whole groups of four)
    u32 w = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      while (nextAt <= x + k) { sy++; nextAt = (sy + 1 < 256) ? (int)cumf[sy + 1] : 0x7FFFFFFF; }
      w |= (u32)sy << (8 * k);
    }
    *(u32*)(f2s + x) = w;
  }
}

// chunk decode: one wave per chunk
__global__ __launch_bounds__(64) void k_ans_dec_chunk(const u8* __restrict__ in, int64_t inStride, const int64_t* __restrict__ d_bitOff,
                                                       const int32_t* __restrict__ d_len, AnsDec D, u8* __restrict__ dst, int64_t stride) {
  const int b = blockIdx.y, ck = blockIdx.x;
  const int count = d_len[b];
  const int lane = kz_lane();
  const u8* p = in + (int64_t)b * inStride;
  u8* o = dst + (int64_t)b * stride;
  if (count <= 32) {                                              // raw (ANSRangeDecoder.java:194-197)
    if (ck != 0) return;
    if (lane < count) o[lane] = (u8)kz_peek(p, (u64)d_bitOff[b] + 8ULL * lane, 8);
    return;
  }
  const int start = ck * ANS_CHUNK;
  if (start >= count || ck >= D.nIdx[b]) return;
  const int end = min(count, start + ANS_CHUNK);
  __shared__ u16 freq[256];
  __shared__ u16 cumf[256];
  __shared__ u8 alpha[256];
  __shared__ __attribute__((aligned(4))) u8 f2s[4096];
  __shared__ int sh_asz, sh_lr, sh_bad;
  __shared__ u64 sh_pos;
  for (int i = lane; i < 256; i += 64) freq[i] = 0;
  __syncthreads();
  if (lane == 0) {                                                // decodeHeader :452-544
    u64 pos = D.chunkBit[(int64_t)b * D.C + ck];
    int bad = 0;
    const int lr = 8 + (int)kz_peek(p, pos, 3); pos += 3;
    const int scale = 1 << lr;
    int asz = 0;
    if (kz_peek(p, pos, 1) == 0) {
      if (kz_peek(p, pos + 1, 1) == 1) asz = 0; else { asz = 256; for (int i = 0; i < 256; i++) alpha[i] = (u8)i; }
      pos += 2;
    } else {
      const int lastMask = (int)kz_peek(p, pos + 1, 5); pos += 6;
      for (int i = 0; i <= lastMask; i++) {
        const u32 m = kz_peek(p, pos, 8); pos += 8;
        for (int j = 0; j < 8; j++) if (m & (1u << j)) alpha[asz++] = (u8)((i << 3) + j);
      }
    }
    if (asz > 0) {
      int llr = 3;
      while ((1 << llr) <= lr) llr++;
      const int chkSize = (asz >= 64) ? 8 : 6;
      int sum = 0;
      for (int i = 1; i < asz && !bad; i += chkSize) {
        const int logMax = (int)kz_peek(p, pos, llr); pos += llr;
        if ((1 << logMax) > scale) { bad = 1; break; }
        const int endj = (i + chkSize < asz) ? i + chkSize : asz;
        for (int j = i; j < endj; j++) {
          const int fq = (logMax == 0) ? 1 : 1 + (int)kz_peek(p, pos, logMax);
          pos += logMax;
          if (fq <= 0 || fq >= scale) { bad = 1; break; }
          freq[alpha[j]] = (u16)fq; sum += fq;
        }
      }
      if (scale <= sum) bad = 1;
      if (!bad) freq[alpha[0]] = (u16)(scale - sum);
    }
    sh_asz = asz; sh_lr = lr; sh_bad = bad; sh_pos = pos;
  }
  __syncthreads();
  const int asz = sh_asz, lr = sh_lr;
  if (sh_bad || asz == 0) { if (lane == 0) atomicMin(&D.event[b], ck * 4 + ANS_EV_FAIL); return; }
  if (ck == D.hdrOnly[b]) return;                                 // header was all that had to be checked
  if (asz == 1) {                                                 // :217-220
    const u8 c = alpha[0];
    for (int i = start + lane; i < end; i += 64) o[i] = c;
    return;
  }
  // cumulative frequencies + freq2sym (reverse mapping :529-538)
  u32 cum = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const u32 fv = freq[q * 64 + lane];
    const u32 inc = kz_wave_incl_sum(fv);
    cumf[q * 64 + lane] = (u16)(cum + inc - fv);
    cum += __shfl(inc, 63, 64);
  }
  __syncthreads();
  // the encoder never goes above logRange 12 (ANSRangeEncoder.java:38); a header may still say up to 15, which the
  // reference accepts (:453-458): those chunks look symbols up by searching the cumulative table instead of f2s
  const bool wide = lr > 12;
  if (!wide) ans_fill_f2s(cumf, f2s, 1 << lr, lane);
  __syncthreads();
  // decodeChunkV2 :357-440
  u64 pos = sh_pos;
  u32 v = kz_peek(p, pos, 8); pos += 8;
  u32 sz = v & 0x7F; int shift = 7;
  while (v >= 128) { v = kz_peek(p, pos, 8); pos += 8; sz |= (v & 0x7F) << shift; if (shift == 28) break; shift += 7; }
  // lane j decodes state st(3-j): order st3, st2, st1, st0 (:392-405)
  u32 st = 0;
  if (lane < 4) { const u64 sp = pos + 32ULL * (3 - lane); st = (kz_peek(p, sp, 16) << 16) | kz_peek(p, sp + 16, 16); }
  pos += 128;
  const u64 payloadBit = pos;
  const u32 mask = (1u << lr) - 1u;
  const int len = end - start;
  const int end4 = len & -4;
  u32 n = 0;
  if (lane < 4) {
    // The payload (bit-unaligned in the stream) is read through a 16-byte register window [rHi | rLo] of raw stream bytes that
    // slides 8 bytes at a time; the next 8 bytes are requested when the window slides, a step or more before anything in them is
    // used.  A step consumes at most 8 payload bytes, so with n <= wbase + 7 at its start every byte pair it may read lies inside the
    // window.  (Before: two 4-byte bit peeks per lane at the END of the step's dependent chain.)
    const u64 payByte = payloadBit >> 3;
    const u32 payShift = (u32)(payloadBit & 7);
    const bool winOk = payByte + (u64)sz + 48 <= (u64)inStride;          // the window never reads past the block's slot
    u32 wbase = 0;
    u64 rHi = 0, rLo = 0, rNx = 0, rNx2 = 0;                             // rNx, rNx2: raw words (byte-swapped when they enter the window)
    if (winOk) {
      rHi = __builtin_bswap64(*(const ans_u64_unaligned*)(p + payByte));
      rLo = __builtin_bswap64(*(const ans_u64_unaligned*)(p + payByte + 8));
      rNx = *(const ans_u64_unaligned*)(p + payByte + 16);
      rNx2 = *(const ans_u64_unaligned*)(p + payByte + 24);
    }
    for (int i = 0; i < end4; i += 4) {
      if (winOk && n > wbase + 7) {                                     // uniform over the four lanes (n is)
        rHi = rLo; rLo = __builtin_bswap64(rNx); rNx = rNx2; wbase += 8;
        // A malformed chunk may consume more than its size says: the address stops at the chunk's end (those bytes are never used).
        // (The compiler copies the loaded word into the loop-carried register at once, i.e. waits for the load right here; the ~29
        // waves per CU hide that.  Four or eight chunks per wave with the load issued from an asm statement and the stores batched at
        // the slides were built and measured slower, 62 .. 100 ms against 67: the chunks of a wave slide at different steps, so the
        // wave meets a wait for a just-issued operation on nearly every step.  DESIGN 5.0.)
        rNx2 = *(const ans_u64_unaligned*)(p + payByte + min(wbase, sz) + 24);
      }
      u32 cur;
      if (!wide) cur = f2s[st & mask];
      else {                                                      // last symbol whose cumulative frequency is <= the slot
        const u32 x = st & mask;
        int lo = 0, hi = 256;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((u32)cumf[mid] <= x) lo = mid; else hi = mid; }
        cur = (u32)lo;
      }
      o[start + i + lane] = (u8)cur;
      u32 fq = freq[cur]; if (fq >= (1u << lr)) fq = (1u << lr) - 1u;         // Symbol.reset mirror :576-579
      st = fq * (st >> lr) + (st & mask) - (u32)cumf[cur];
      const bool need = (int)st < (int)ANS_TOP;
      const uint64_t bal = kz_ballot(need) & 0xFULL;
      if (need) {
        const u32 off = 2u * (u32)__popcll(bal & kz_lanemask_lt());
        const u32 at = n + off;
        u32 hi, lo;
        if (winOk) {
          const u32 bp = 8u * (at - wbase) + payShift;                    // bit position of the pair inside the window: < 112
          const u64 v = (bp < 64) ? ((rHi << bp) | (bp ? (rLo >> (64 - bp)) : 0ULL)) : (rLo << (bp - 64));
          const u32 two = (u32)(v >> 48);
          hi = (at < sz) ? (two >> 8) : 0u; lo = (at + 1 < sz) ? (two & 0xFFu) : 0u;
        }
        else { hi = (at < sz) ? kz_peek(p, payloadBit + 8ULL * at, 8) : 0; lo = (at + 1 < sz) ? kz_peek(p, payloadBit + 8ULL * (at + 1), 8) : 0; }
        st = (st << 16) | (hi << 8) | lo;
      }
      n += 2u * (u32)__popcll(bal);
    }
  }
  n = __shfl(n, 0, 64);
  if (lane < len - end4) { const u32 at = n + (u32)lane; o[start + end4 + lane] = (at < sz) ? (u8)kz_peek(p, payloadBit + 8ULL * at, 8) : 0; }
  n += (u32)(len - end4);
  if (lane == 0 && n != sz) atomicMin(&D.event[b], ck * 4 + ANS_EV_STOP);          // :439
}

// one wave per block: verdict of the lowest-chunk event, zero fill of what the reference leaves unwritten
__global__ __launch_bounds__(64) void k_ans_dec_fin(const int32_t* __restrict__ d_len, int32_t* __restrict__ d_len2, int32_t* __restrict__ d_flag,
                                                     AnsDec D, u8* __restrict__ dst, int64_t stride) {
  const int b = blockIdx.x, lane = kz_lane();
  const int count = d_len[b];
  const int ev = D.event[b];
  if (lane == 0) { d_len2[b] = count; d_flag[b] = (ev == ANS_EV_NONE || (ev & 3) != ANS_EV_FAIL) ? 1 : 0; }
  if (ev == ANS_EV_NONE || (ev & 3) == ANS_EV_FAIL) return;
  const int64_t from = (int64_t)((ev >> 2) + ((ev & 3) == ANS_EV_STOP ? 1 : 0)) * ANS_CHUNK;
  u8* o = dst + (int64_t)b * stride;
  for (int64_t i = from + lane; i < count; i += 64) o[i] = 0;
}

int kz_stage_ans0_decode(kz_ctx* ctx, kz_batch& bt, const uint8_t* in, int64_t inStride, const int64_t* d_bitOff, const int64_t* d_bitEnd) {
  const int B = bt.B;
  int maxN = 0;
  for (int b = 0; b < B; b++) if (bt.h_len[b] > maxN) maxN = bt.h_len[b];
  AnsDec D;
  D.C = (maxN + ANS_CHUNK - 1) / ANS_CHUNK + 1;
  D.chunkBit = (u64*)kz_arena_alloc(ctx, (size_t)B * D.C * 8);
  D.event = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  D.nIdx = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  D.hdrOnly = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  if (!D.event || !D.nIdx || !D.hdrOnly || !D.chunkBit) { snprintf(ctx->err, sizeof(ctx->err), "ans0_decode: arena overflow"); return -KZ_ERR_DEVICE; }
  hipStream_t st = ctx->stream;
  u8* dst = bt.buf[bt.cur ^ 1];
  KZ_LAUNCH(ctx, KID_ANS_DEC_INDEX, k_ans_dec_index, dim3((B + 63) / 64), dim3(64), in, inStride, d_bitOff, d_bitEnd, bt.d_len, D, B, ctx->d_endBits);
  const int chunks = (maxN + ANS_CHUNK - 1) / ANS_CHUNK;
  if (chunks > 0) KZ_LAUNCH(ctx, KID_ANS_DEC_CHUNK, k_ans_dec_chunk, dim3(chunks, B), dim3(64), in, inStride, d_bitOff, bt.d_len, D, dst, bt.stride);
  KZ_LAUNCH(ctx, KID_ANS_DEC_FIN, k_ans_dec_fin, dim3(B), dim3(64), bt.d_len, bt.d_len2, bt.d_flag, D, dst, bt.stride);
  KZ_HIP(hipGetLastError());
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}
