// kz_fpaq.hip -- FPAQ adaptive order-0 binary arithmetic coder on gfx950.
//
// Replaces K/entropy/FPAQEncoder.java:128-173 (encode), :182-199 (encodeBit), :208-213 (flush),
// :232-238 (dispose) and K/entropy/FPAQDecoder.java:161-242, :290-314 (decodeBitV2), :322-335 (read).
//
// The coder state (56-bit low/high, 4 x 256 adaptive probabilities) is carried across the whole block
// and every bit depends on the previous one: there is no parallelism inside a block, only
// across the blocks of the batch.  A wave64 instruction costs the same whether one lane or 64 lanes use
// it, so ONE LANE codes ONE BLOCK: a wave runs 64 independent coders in lock step on the VALU (the
// per-bit work is ~60 instructions for 64 blocks instead of ~45 for one), each lane with its own
// probability table in LDS (1024 x u16 per lane, interleaved [entry][lane]: 128 KiB per wave, at most
// 2-way bank conflicts), its own input cursor (next 32-bit word prefetched) and its own output cursor.
// The output of a block is a byte string [varint(n) | n bytes | 56-bit tail]* per 4 MiB chunk, written
// byte aligned behind the block header by a second, data-parallel kernel.
#include "kz_device.h"
#include "kz_internal.h"
#include <stdlib.h>
#include <vector>

typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t u8;

#define FP_TOP 0x00FFFFFFFFFFFFFFULL
#define FP_M2456 0x00FFFFFFFF000000ULL
#define FP_M024 0x0000000000FFFFFFULL
#define FP_M032 0x00000000FFFFFFFFULL
#define FP_M056 0x00FFFFFFFFFFFFFFULL
#define FP_CHUNK (4 * 1024 * 1024)
#define FP_PSCALE 65536            // probabilities stay in [63, 65472]: they fit u16

typedef u32 __attribute__((aligned(1))) fp_u32_unaligned;
__device__ __forceinline__ u32 fp_be32(const u8* p) { return __builtin_bswap32(*(const fp_u32_unaligned*)p); }

// per-block, per-chunk results of the encoder lanes
struct FpaqChunks {
  u32* bytes;     // [B][maxChunks] payload bytes of the chunk
  u64* tail;      // [B][maxChunks] low | MASK_0_24 after the chunk
  int maxChunks;
};

// One range-coder step (FPAQEncoder.java:182-199 encodeBit + :208-213 flush) of this lane's block.
#define FP_ENC_BIT(PP, BIT)                                                                    \
  { const u64 split = (((high - low) >> 8) * (u64)(u32)(PP)) >> 8;                             \
    const bool one = (BIT) != 0;                                                               \
    const u64 nh = low + split, nl = nh + 1;                                                   \
    high = one ? nh : high; low = one ? low : nl;                                              \
    if (((low ^ high) & FP_M2456) == 0) {            /* never twice in a row: bits 24..31 then differ (00 vs FF) */ \
      *(u32*)(sba + idx) = __builtin_bswap32((u32)(high >> 24));                               \
      idx += 4;                                                                                \
      low <<= 32;                                                                              \
      high = (high << 32) | FP_M032;                                                           \
    } }
#define FP_UPD(P, BIT) (u16)((BIT) ? (P) - (((P) - FP_PSCALE + 64) >> 6) : (P) - ((P) >> 6))

// Encoder: the 8 contexts of a byte are known up front (the byte is known) and are 8 distinct table entries,
// and the probability update does not depend on the coder state: 8 LDS reads, 8 LDS writes, then the 8
// sequential range-coder steps.
__global__ __launch_bounds__(64) void k_fpaq_enc(const u8* __restrict__ src, int64_t stride, const int32_t* __restrict__ d_len,
                                                  u8* __restrict__ scr, int64_t scrStride, FpaqChunks C, int B) {
  __shared__ u16 probs[1024][64];
  const int lane = threadIdx.x;
  const int b = blockIdx.x * 64 + lane;
  for (int i = 0; i < 1024; i++) probs[i][lane] = (u16)(FP_PSCALE >> 1);   // own column only: no barrier needed
  const int count = (b < B) ? d_len[b] : 0;
  if (count <= 0) return;
  const u8* blk = src + (int64_t)b * stride;
  u8* sba = scr + (int64_t)b * scrStride;
  u32* cbytes = C.bytes + (int64_t)b * C.maxChunks;
  u64* ctail = C.tail + (int64_t)b * C.maxChunks;
  u64 low = 0, high = FP_TOP;
  u32 idx = 0, chunkStart = 0;
  int chunk = 0, chunkEnd = min(FP_CHUNK, count);
  int tb = 0;                                                       // this.p = this.probs[0] (:148)
  u32 word = 0, wordNext = *(const u32*)blk;                        // input words are fetched one group (4 bytes) ahead
  for (int i = 0; i < count; i++) {
    if (i == chunkEnd) {                                            // :164-169: close the chunk
      cbytes[chunk] = idx - chunkStart; ctail[chunk] = low | FP_M024;
      chunk++; chunkStart = idx; chunkEnd = min(i + FP_CHUNK, count); tb = 0;
    }
    if ((i & 3) == 0) { word = wordNext; wordNext = *(const u32*)(blk + i + 4); }   // blocks start 256-byte aligned, strides have slack
    const int val = (int)((word >> (8 * (i & 3))) & 0xFFu);
    const int bits = val + 256;
    const int i7 = tb + 1, i6 = tb + (bits >> 7), i5 = tb + (bits >> 6), i4 = tb + (bits >> 5),
              i3 = tb + (bits >> 4), i2 = tb + (bits >> 3), i1 = tb + (bits >> 2), i0 = tb + (bits >> 1);
    const int p7 = probs[i7][lane], p6 = probs[i6][lane], p5 = probs[i5][lane], p4 = probs[i4][lane],
              p3 = probs[i3][lane], p2 = probs[i2][lane], p1 = probs[i1][lane], p0 = probs[i0][lane];
    probs[i7][lane] = FP_UPD(p7, val & 0x80); probs[i6][lane] = FP_UPD(p6, val & 0x40);
    probs[i5][lane] = FP_UPD(p5, val & 0x20); probs[i4][lane] = FP_UPD(p4, val & 0x10);
    probs[i3][lane] = FP_UPD(p3, val & 0x08); probs[i2][lane] = FP_UPD(p2, val & 0x04);
    probs[i1][lane] = FP_UPD(p1, val & 0x02); probs[i0][lane] = FP_UPD(p0, val & 0x01);
    FP_ENC_BIT(p7, val & 0x80) FP_ENC_BIT(p6, val & 0x40) FP_ENC_BIT(p5, val & 0x20) FP_ENC_BIT(p4, val & 0x10)
    FP_ENC_BIT(p3, val & 0x08) FP_ENC_BIT(p2, val & 0x04) FP_ENC_BIT(p1, val & 0x02) FP_ENC_BIT(p0, val & 0x01)
    tb = (val >> 6) << 8;                                           // :161
  }
  cbytes[chunk] = idx - chunkStart; ctail[chunk] = low | FP_M024;   // last chunk; its tail is dispose() :232-238
}

// [varint(n) | n bytes | 7-byte tail] per chunk (EntropyUtils.writeVarInt; FPAQEncoder.java:164-169, :232-238)
__global__ __launch_bounds__(256) void k_fpaq_pack(const int32_t* __restrict__ d_len, const u8* __restrict__ scr, int64_t scrStride,
                                                    FpaqChunks C, u8* __restrict__ out, int64_t outStride,
                                                    const int32_t* __restrict__ d_hdrBytes, int64_t* __restrict__ d_bits) {
  const int b = blockIdx.x;
  const int count = d_len[b];
  if (count <= 0) { if (threadIdx.x == 0) d_bits[b] = 0; return; }
  const int nchunks = (count + FP_CHUNK - 1) / FP_CHUNK;
  const u8* sba = scr + (int64_t)b * scrStride;
  u8* o = out + (int64_t)b * outStride + d_hdrBytes[b];
  int64_t opos = 0, spos = 0;
  for (int c = 0; c < nchunks; c++) {
    const u32 nbytes = C.bytes[(int64_t)b * C.maxChunks + c];
    const u64 t = C.tail[(int64_t)b * C.maxChunks + c];
    u32 v = nbytes;
    while (v >= 128) { if (threadIdx.x == 0) o[opos] = (u8)(0x80 | (v & 0x7F)); opos++; v >>= 7; }
    if (threadIdx.x == 0) o[opos] = (u8)v;
    opos++;
    for (u32 k = threadIdx.x; k < nbytes; k += 256) o[opos + k] = sba[spos + k];
    opos += nbytes; spos += nbytes;
    if (threadIdx.x < 7) o[opos + threadIdx.x] = (u8)(t >> (8 * (6 - threadIdx.x)));
    opos += 7;
  }
  if (threadIdx.x == 0) d_bits[b] = 8LL * opos;
}

// One decoder step (FPAQDecoder.java:290-314 decodeBitV2 + :322-335 read) of this lane's block.  PR = probability
// of the current context; its two children were fetched from LDS one level earlier, the next input word when
// the previous one was consumed.
#define FP_DEC_BIT(LEVEL)                                                                      \
  { int c0 = 0, c1 = 0;                                                                        \
    if (LEVEL < 7) { c0 = probs[tb + 2 * ctx][lane]; c1 = probs[tb + 2 * ctx + 1][lane]; }     \
    const u64 split = ((((high - low) >> 8) * (u64)(u32)pr) >> 8) + low;                       \
    const bool one = split >= current;                                                         \
    const int np = one ? pr - ((pr - FP_PSCALE + 64) >> 6) : pr - (pr >> 6);                   \
    high = one ? split : high; low = one ? low : split + 1;                                    \
    probs[tb + ctx][lane] = (u16)np;                                                           \
    ctx = (ctx << 1) + (one ? 1 : 0);                                                          \
    if (LEVEL == 1) rootNext = probs[((ctx & 3) << 8) + 1][lane];  /* next byte's first context (:233-239) */ \
    if (((low ^ high) & FP_M2456) == 0) {            /* never twice in a row: bits 24..31 then differ (00 vs FF) */ \
      low = (low << 32) & FP_M056;                                                             \
      high = ((high << 32) | FP_M032) & FP_M056;                                               \
      if (idx + 4 > bufLimit) { current = (current << 32) & FP_M056; idx = bufLimit + 1; }     \
      else { current = ((current << 32) | (u64)__builtin_bswap32(nextw)) & FP_M056; idx += 4; nextw = ring[(idx >> 2) & 63][lane]; } \
    }                                                                                          \
    if (LEVEL < 7) pr = one ? c1 : c0; }

__global__ __launch_bounds__(64) void k_fpaq_dec(const u8* __restrict__ in, int64_t inStride, const int64_t* __restrict__ d_bitOff,
                                                  const int64_t* __restrict__ d_bitEnd, const int32_t* __restrict__ d_len,
                                                  u8* __restrict__ dst, int64_t stride, int32_t* __restrict__ d_len2, int32_t* __restrict__ d_flag, int B, long long* __restrict__ endOut) {
  __shared__ u16 probs[1024][64];
  // input ring: the next 64 dwords of this lane's chunk, [slot][lane].  A global load issued when a lane consumes a
  // word would be waited for by whichever lane flushes next (one VGPR, in-order vmcnt): every bit step would pay a
  // memory round trip.  The ring is topped up at byte granularity instead, one step behind the load.
  __shared__ u32 ring[64][64];
  const int lane = threadIdx.x;
  const int b = blockIdx.x * 64 + lane;
  for (int i = 0; i < 1024; i++) probs[i][lane] = (u16)(FP_PSCALE >> 1);   // own column only: no barrier needed
  if (b >= B) return;
  const int count = d_len[b];
  d_len2[b] = count; d_flag[b] = 1;
  if (count <= 0) return;
  const u8* p = in + (int64_t)b * inStride + (d_bitOff[b] >> 3);   // payload is byte aligned behind the block header
  const int64_t avail = (d_bitEnd[b] - d_bitOff[b]) >> 3;
  u8* o = dst + (int64_t)b * stride;
  int64_t ipos = 0;
  u64 low = 0, high = FP_TOP, current = 0;
  bool bad = ((d_bitOff[b] & 7) != 0);
  const u8* buf = p;
  int bufLimit = 0, idx = 0, chunkEnd = 0, tb = 0;
  u32 nextw = 0, outw = 0;
  int rootNext = FP_PSCALE >> 1;
  int fillw = 0, pend = 0;                                          // dwords in the ring / dwords loaded but not yet stored to it
  u32 pw0 = 0, pw1 = 0, pw2 = 0, pw3 = 0;
  int i = 0;
  while (i < count && !bad) {
    // top up the ring: store the 4 dwords requested at the previous byte, request the next 4 (a byte consumes <= 3)
    if (pend) { ring[fillw & 63][lane] = pw0; ring[(fillw + 1) & 63][lane] = pw1; ring[(fillw + 2) & 63][lane] = pw2; ring[(fillw + 3) & 63][lane] = pw3; fillw += 4; pend = 0; }
    if (i != chunkEnd && fillw - (idx >> 2) <= 56 && 4 * fillw < bufLimit + 4) {       // never more than 19 bytes past the chunk
      const u8* q = buf + 4 * fillw;
      pw0 = *(const fp_u32_unaligned*)q; pw1 = *(const fp_u32_unaligned*)(q + 4); pw2 = *(const fp_u32_unaligned*)(q + 8); pw3 = *(const fp_u32_unaligned*)(q + 12);
      pend = 1;
    }
    if (i == chunkEnd) {                                            // chunk header (FPAQDecoder.java:170-183)
      u32 v = p[ipos++]; u32 sz = v & 0x7F; int shift = 7;         // EntropyUtils.readVarInt
      while (v >= 128) { v = p[ipos++]; sz |= (v & 0x7F) << shift; if (shift == 28) break; shift += 7; }
      const int szBytes = (int)sz;
      if (szBytes < 0 || szBytes >= 2 * count || ipos + 7 + szBytes > avail) { bad = true; break; }   // :176-177
      current = 0;
      for (int k = 0; k < 7; k++) current = (current << 8) | (u64)p[ipos + k];
      ipos += 7;
      buf = p + ipos; bufLimit = szBytes; idx = 0;
      ipos += szBytes;
      for (int k = 0; k < 16; k++) ring[k][lane] = (4 * k < bufLimit + 4) ? *(const fp_u32_unaligned*)(buf + 4 * k) : 0u;   // at most 7 bytes past the chunk
      fillw = 16; pend = 0;
      nextw = ring[0][lane];                                        // raw (little-endian load): swapped when consumed
      chunkEnd = i + min(FP_CHUNK, count - i);
      tb = 0;
      rootNext = probs[1][lane];
    }
    int ctx = 1;
    int pr = rootNext;
    FP_DEC_BIT(0) FP_DEC_BIT(1) FP_DEC_BIT(2) FP_DEC_BIT(3) FP_DEC_BIT(4) FP_DEC_BIT(5) FP_DEC_BIT(6) FP_DEC_BIT(7)
    outw |= ((u32)ctx & 0xFFu) << (8 * (i & 3));
    if ((i & 3) == 3) { *(u32*)(o + i - 3) = outw; outw = 0; }     // blocks start 256-byte aligned
    if (idx > bufLimit) { bad = true; break; }                      // :231-232
    tb = ((ctx & 0xFF) >> 6) << 8;
    i++;
  }
  if (!bad) for (int k = 0; k < (count & 3); k++) o[(count & ~3) + k] = (u8)(outw >> (8 * k));
  if (endOut) endOut[b] = (long long)(d_bitOff[b] + 8LL * ipos);     // bits consumed (EntropyDecoder contract)
  if (bad) d_flag[b] = 0;
}

// ================================================================================================
// One WAVE per block (up to 8 blocks per workgroup): used for batches of at most 8 blocks per CU, where it is the
// faster arrangement (a lone block: 2.1 s encode / 3.6 s decode instead of 4.3 / 5.2 s); the lane-per-block kernels
// above take over for larger batches, where their time stays constant.  Range arithmetic runs on the scalar unit.
#define FPW_SYNC() __builtin_amdgcn_wave_barrier()   /* one wave per block: program order suffices */
// wave-cooperative byte copy (all lanes must call)
__device__ __forceinline__ void fp_copy(u8* __restrict__ d, const u8* __restrict__ s, int n) {
  for (int i = kz_lane(); i < n; i += 64) d[i] = s[i];
}

// One range-coder step (FPAQEncoder.java:182-199 encodeBit + :208-213 flush), all operands wave-uniform.
// Written branch-free (selects) except for the rare flush: a taken scalar branch costs a lone wave ~40 cycles.
// ((range >> 8) * p) >> 8 for range < 2^56, p < 2^16 in nine scalar instructions: with range >> 8 = a * 2^32 + b the product is
// ((a * p + hi32(b * p)) << 32) | lo32(b * p) (a * p < 2^32: a < 2^16).  The empty asm keeps the compiler from re-deriving a as
// range.hi >> 8 with a shift of its own.
__device__ __forceinline__ u64 fpw_scale(u64 range, u32 p) {
  u64 r8 = range >> 8;
  asm("" : "+s"(r8));
  const u32 a = (u32)(r8 >> 32), b = (u32)r8;
  const u64 bp = (u64)b * p;
  typedef u32 fpw_u32x2 __attribute__((ext_vector_type(2)));
  const fpw_u32x2 w = {(u32)bp, (u32)(bp >> 32) + a * p};
  return __builtin_bit_cast(u64, w) >> 8;
}
// Round 3: the selects and the test of :190 as five scalar instructions (the compiler's form took ten: 32-bit halves, a separate
// compare per use, the mask in two literals); KBIT = position of the coded bit in VAL.  t == 0: flush (:208-213).
#define FPW_ENC_BIT(PP, VAL, KBIT)                                                              \
  { const u64 nh = low + fpw_scale(high - low, (u32)(PP)), nl = nh + 1;                        \
    u64 t;                                                                                     \
    asm volatile("s_bitcmp1_b32 %[val], " #KBIT "\n\t"                                         \
                 "s_cselect_b64 %[high], %[nh], %[high]\n\t"                                  \
                 "s_cselect_b64 %[low], %[low], %[nl]\n\t"                                    \
                 "s_xor_b64 %[t], %[low], %[high]\n\t"                                        \
                 "s_and_b64 %[t], %[t], %[mask]"                                               \
                 : [high] "+s"(high), [low] "+s"(low), [t] "=&s"(t) : [val] "s"(VAL), [nh] "s"(nh), [nl] "s"(nl), [mask] "s"(m2456) : "scc"); \
    if (__builtin_expect(t == 0, 0)) {              /* never twice in a row: bits 24..31 then differ (00 vs FF) */ \
      if (lane == 0) { const u32 w = (u32)(high >> 24); sba[idx] = (u8)(w >> 24); sba[idx + 1] = (u8)(w >> 16); sba[idx + 2] = (u8)(w >> 8); sba[idx + 3] = (u8)w; } \
      idx += 4;                                                                                \
      low <<= 32;                                                                              \
      high = (high << 32) | FP_M032;                                                           \
    } }

// Encoder: the 8 contexts of a byte are known up front (the byte is known) and are 8 distinct table entries,
// and the probability update does not depend on the coder state: lanes 0..7 gather, update and write back the
// 8 probabilities of a byte with ONE LDS read and ONE LDS write; the gather for the next byte is issued before
// the 8 sequential range-coder steps of the current one, which run on the scalar unit.
__global__ __launch_bounds__(512) void k_fpaq_enc_wave(const u8* __restrict__ src, int64_t stride, const int32_t* __restrict__ d_len,
                                                  u8* __restrict__ scr, int64_t scrStride, u8* __restrict__ out, int64_t outStride,
                                                  const int32_t* __restrict__ d_hdrBytes, int64_t* __restrict__ d_bits,
                                                  const int32_t* __restrict__ order, int wavesPerGroup) {
  __shared__ int probsAll[8][1024];
  const int wv = (int)(threadIdx.x >> 6);
  const int b = __builtin_amdgcn_readfirstlane(order[blockIdx.x * wavesPerGroup + __builtin_amdgcn_readfirstlane(wv)]);   // uniform on purpose: the coder state stays on the scalar unit
  if (b < 0) return;
  int* probs = probsAll[wv];
  const int count = d_len[b];
  const int lane = kz_lane();
  if (count <= 0) { if (lane == 0) d_bits[b] = 0; return; }
  for (int i = lane; i < 1024; i += 64) probs[i] = FP_PSCALE >> 1;
  FPW_SYNC();
  const u8* blk = src + (int64_t)b * stride;
  u8* sba = scr + (int64_t)b * scrStride;
  u8* o = out + (int64_t)b * outStride + d_hdrBytes[b];
  int opos = 0;
  u64 low = 0, high = FP_TOP;
  const u64 m2456 = FP_M2456;
  int startChunk = 0;
  const int kbit = 7 - (lane & 7);                                // lane k codes bit 7-k (MSB first)
  while (startChunk < count) {
    const int chunkSize = min(FP_CHUNK, count - startChunk);
    const int chunkEnd = startChunk + chunkSize;
    int idx = 0;
    u32 rowv = (startChunk + lane < chunkEnd) ? (u32)blk[startChunk + lane] : 0u;
    // contexts of the first byte: this.p = this.probs[0] (:148)
    int val = __builtin_amdgcn_readlane((int)rowv, 0);
    int pIdx = 0 + ((kbit == 7) ? 1 : ((val + 256) >> (kbit + 1)));
    int pp = probs[pIdx];
    for (int row = startChunk; row < chunkEnd; row += 64) {         // rows of 64 bytes, the next row requested a row ahead
      const int rowCnt = min(64, chunkEnd - row);
      const u32 rowNext = (row + 64 + lane < chunkEnd) ? (u32)blk[row + 64 + lane] : 0u;
      for (int j = 0; j < rowCnt; j++) {
        // update and write back this byte's 8 probabilities, fetch the next byte's.  Lane l works on bit 7 - (l & 7): the eight
        // copies of a context hold the same value and store it to the same word, which spares the EXEC round trip of "lanes 0..7".
        // Behind the last byte of the chunk the fetch reads the contexts of a byte 0 that is never coded.
        const int bit = (val >> kbit) & 1;
        const int np = bit ? pp - ((pp - FP_PSCALE + 64) >> 6) : pp - (pp >> 6);
        probs[pIdx] = np;
        const int cur = pp;
        const int curVal = val;
        const int nx = __builtin_amdgcn_readlane((int)rowv, (j + 1) & 63), nf = __builtin_amdgcn_readlane((int)rowNext, 0);
        val = (j == 63) ? nf : nx;
        pIdx = ((curVal >> 6) << 8) + ((kbit == 7) ? 1 : ((val + 256) >> (kbit + 1)));    // :161
        pp = probs[pIdx];
        const int p7 = __builtin_amdgcn_readlane(cur, 0), p6 = __builtin_amdgcn_readlane(cur, 1), p5 = __builtin_amdgcn_readlane(cur, 2),
                  p4 = __builtin_amdgcn_readlane(cur, 3), p3 = __builtin_amdgcn_readlane(cur, 4), p2 = __builtin_amdgcn_readlane(cur, 5),
                  p1 = __builtin_amdgcn_readlane(cur, 6), p0 = __builtin_amdgcn_readlane(cur, 7);
        FPW_ENC_BIT(p7, curVal, 7) FPW_ENC_BIT(p6, curVal, 6) FPW_ENC_BIT(p5, curVal, 5) FPW_ENC_BIT(p4, curVal, 4)
        FPW_ENC_BIT(p3, curVal, 3) FPW_ENC_BIT(p2, curVal, 2) FPW_ENC_BIT(p1, curVal, 1) FPW_ENC_BIT(p0, curVal, 0)
      }
      rowv = rowNext;
    }
    // varint(idx) | bytes   (EntropyUtils.writeVarInt; :164-165)
    { u32 v = (u32)idx; while (v >= 128) { if (lane == 0) o[opos] = (u8)(0x80 | (v & 0x7F)); opos++; v >>= 7; } if (lane == 0) o[opos] = (u8)v; opos++; }
    FPW_SYNC();
    fp_copy(o + opos, sba, idx);
    opos += idx;
    startChunk += chunkSize;
    if (startChunk < count) {                                       // :168-169
      const u64 t = low | FP_M024;
      if (lane < 7) o[opos + lane] = (u8)(t >> (8 * (6 - lane)));
      opos += 7;
    }
    FPW_SYNC();
  }
  { const u64 t = low | FP_M024; if (lane < 7) o[opos + lane] = (u8)(t >> (8 * (6 - lane))); opos += 7; }   // dispose :232-238
  if (lane == 0) d_bits[b] = 8LL * opos;
}

// One decoder step (FPAQDecoder.java:290-314 decodeBitV2 + :322-335 read): every bit's context depends on the previous bit, so the
// chain is serial; what can be hidden is the LDS latency of the probability (both children of the current context are requested
// while the current bit is decoded).  Range arithmetic runs on the scalar unit; output bytes are collected per row and stored 64 at a
// time.  (The round-2 form of the step, k_fpaq_dec_wave, was removed in round 6: the round-3 restatement below replaced it.)
// Round 3: the step restated so that a wave issues fewer instructions for it.  A lone wave issues one instruction every ~5.7 cycles
// whatever unit it goes to (two waves per SIMD interleave at that rate each: 2048 blocks take as long as one), so the cost of a
// bit is its instruction count.  The probability and its LDS address never leave the vector unit (update: 4 VALU; the address of
// the node, of its children and of the next node follow from the previous address and the bit: byte address w of probs[tb + ctx]
// -> 2w - tbBase + 4*bit), only the probability itself is copied to the scalar unit for the 56-bit range arithmetic.  `V` marks
// values the compiler must treat as per-lane (they are uniform in fact): that keeps them on the VALU.
typedef __attribute__((address_space(3))) int fpw_lds_int_t;
__device__ __forceinline__ u32 fpw_lds_addr(const int* p) { return (u32)(size_t)(__attribute__((address_space(3))) const int*)p; }   // what ds_* take
__device__ __forceinline__ int fpw_lds_int(int addr) { return *(const fpw_lds_int_t*)(size_t)(u32)addr; }
#define FPW_V(x) asm volatile("" : "+v"(x))
#define FPW_V64(x) asm volatile("" : "+v"(x))
// the start of a step for the node at byte address VW with probability VP: request both children (the two candidates for the
// next node: LDS latency is then covered by the step's arithmetic), copy the probability to the scalar unit
#define FPW_DEC_OPEN()                                                                          \
  asm volatile("v_lshl_add_u32 %[ca], %[vw], 1, %[ntb]\n\t"         /* children of ctx: probs[tb + 2 ctx] */ \
               "ds_read_b32 %[c0], %[ca]\n\t"                                                 \
               "ds_read_b32 %[c1], %[ca] offset:4"                                             \
               : [ca] "=&v"(ca), [c0] "=&v"(c0), [c1] "=&v"(c1) : [vw] "v"(vw), [ntb] "v"(vNegTb) : "memory");   \
  pr = __builtin_amdgcn_readfirstlane(vp)
#define FPW_DEC_BIT2(LEVEL)                                                                     \
  { const u64 split = fpw_scale(high - low, (u32)pr) + low;                                              \
    const u64 split1 = split + 1;                                                              \
    u64 onem, t; int tmp;                                                                      \
    /* the decision (:296-309), the probability update, the next node and the test of :312 in one go; then the next step's start */ \
    asm volatile("v_cmp_ge_u64 vcc, %[split], %[cur]\n\t"           /* bit = split >= current */ \
                 "s_and_b64 %[one], vcc, exec\n\t"                  /* SCC = bit */          \
                 "s_cselect_b64 %[high], %[split], %[high]\n\t"                               \
                 "s_cselect_b64 %[low], %[low], %[split1]\n\t"                                \
                 "s_xor_b64 %[t], %[low], %[high]\n\t"                                        \
                 "s_and_b64 %[t], %[t], %[mask]\n\t"                                          \
                 "v_cndmask_b32 %[tmp], 0, %[vk], vcc\n\t"          /* p -= (p - (bit ? PSCALE - 64 : 0)) >> 6  (:302 / :307) */ \
                 "v_sub_u32 %[tmp], %[vp], %[tmp]\n\t"                                        \
                 "v_ashrrev_i32 %[tmp], 6, %[tmp]\n\t"                                        \
                 "v_sub_u32 %[tmp], %[vp], %[tmp]\n\t"                                        \
                 "ds_write_b32 %[vw], %[tmp]\n\t"                                             \
                 "v_cndmask_b32 %[tmp], 0, %[four], vcc\n\t"                                  \
                 "v_add_u32 %[vw], %[ca], %[tmp]\n\t"               /* the next node: probs[tb + 2 ctx + bit] */ \
                 "s_waitcnt lgkmcnt(1)\n\t"                         /* the two reads (LDS returns in order; the write may still be out) */ \
                 "v_cndmask_b32 %[vp], %[c0], %[c1], vcc"                                      \
                 : [one] "=&s"(onem), [high] "+s"(high), [low] "+s"(low), [t] "=&s"(t), [tmp] "=&v"(tmp), [vw] "+v"(vw), [vp] "+v"(vp) \
                 : [cur] "v"(current), [split] "s"(split), [split1] "s"(split1), [mask] "s"(m2456), [vk] "v"(vK), [four] "v"(vFour), [ca] "v"(ca), [c0] "v"(c0), [c1] "v"(c1) \
                 : "vcc", "scc", "memory");                                                    \
    if (LEVEL == 1) { const int tbn = ((vw + vNegTb) >> 2) & 3; vRootAddr = (tbn << 10) + vBase4; rootNext = fpw_lds_int(vRootAddr); } \
    if (LEVEL < 7) FPW_DEC_OPEN();                                                             \
    if (__builtin_expect(t == 0, 0)) {                              /* never twice in a row: bits 24..31 then differ (00 vs FF) */ \
      low = (low << 32) & FP_M056;                                                             \
      high = ((high << 32) | FP_M032) & FP_M056;                                               \
      if (idx + 4 > bufLimit) { current = (current << 32) & FP_M056; idx = bufLimit + 1; }     \
      else {                                                                                   \
        if (idx >= wbase + 256) { wbase = idx; const u8* q = buf + wbase + 4 * lane; win = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3]; } \
        const u64 val = (u64)(u32)__builtin_amdgcn_readlane((int)win, (idx - wbase) >> 2);     \
        current = ((current << 32) | val) & FP_M056;                                           \
        idx += 4;                                                                              \
      }                                                                                        \
      FPW_V64(current);                                                                        \
    } }

__global__ __launch_bounds__(512) void k_fpaq_dec_wave2(const u8* __restrict__ in, int64_t inStride, const int64_t* __restrict__ d_bitOff,
                                                  const int64_t* __restrict__ d_bitEnd, const int32_t* __restrict__ d_len,
                                                  u8* __restrict__ dst, int64_t stride, int32_t* __restrict__ d_len2, int32_t* __restrict__ d_flag,
                                                  const int32_t* __restrict__ order, int wavesPerGroup, long long* __restrict__ endOut) {
  __shared__ __attribute__((aligned(8))) int probsAll[8][1024];
  const int wv = (int)(threadIdx.x >> 6);
  const int b = __builtin_amdgcn_readfirstlane(order[blockIdx.x * wavesPerGroup + __builtin_amdgcn_readfirstlane(wv)]);   // uniform on purpose: the coder state stays on the scalar unit
  if (b < 0) return;
  int* probs = probsAll[wv];
  const int count = d_len[b];
  const int lane = kz_lane();
  if (lane == 0) { d_len2[b] = count; d_flag[b] = 1; }
  if (count <= 0) { if (endOut && lane == 0) endOut[b] = d_bitOff[b]; return; }
  for (int i = lane; i < 1024; i += 64) probs[i] = FP_PSCALE >> 1;
  FPW_SYNC();
  const u8* p = in + (int64_t)b * inStride + (d_bitOff[b] >> 3);   // payload is byte aligned behind the block header
  const int64_t avail = (d_bitEnd[b] - d_bitOff[b]) >> 3;
  u8* o = dst + (int64_t)b * stride;
  int64_t ipos = 0;
  u64 low = 0, high = FP_TOP, current = 0;
  bool bad = ((d_bitOff[b] & 7) != 0);
  int startChunk = 0;
  while (startChunk < count && !bad) {
    // varint (EntropyUtils.readVarInt)
    u32 v = p[ipos++]; u32 sz = v & 0x7F; int shift = 7;
    while (v >= 128) { v = p[ipos++]; sz |= (v & 0x7F) << shift; if (shift == 28) break; shift += 7; }
    const int szBytes = (int)sz;
    if (szBytes < 0 || szBytes >= 2 * count || ipos + 7 + szBytes > avail) { bad = true; break; }   // :176-177
    current = 0;
    for (int k = 0; k < 7; k++) current = (current << 8) | (u64)p[ipos + k];
    ipos += 7;
    const u8* buf = p + ipos;
    const int bufLimit = szBytes;
    int idx = 0;
    // 256-byte read window (one big-endian word per lane) over the chunk's byte stream
    int wbase = 0;
    u32 win;
    { const u8* q = buf + 4 * lane; win = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3]; }
    const int chunkSize = min(FP_CHUNK, count - startChunk);
    int vBase4 = (int)fpw_lds_addr(&probs[1]);                       // LDS byte address of probs[0 * 256 + 1]
    int vK = FP_PSCALE - 64;
    const u64 m2456 = FP_M2456;
    int vFour = 4;
    FPW_V(vBase4); FPW_V(vK); FPW_V(vFour); FPW_V64(current);
    int rootNext = probs[1];
    int vRootAddr = vBase4;                                         // where rootNext was read: probs[tb + 1]
    u32 outv = 0;
    for (int i = startChunk; i < startChunk + chunkSize && !bad; i += 64) {      // rows of 64 bytes: one store per row
      const int rowCnt = min(64, startChunk + chunkSize - i);
      for (int j = 0; j < rowCnt; j++) {
        int vp = rootNext;
        int vw = vRootAddr;                                         // byte address of probs[tb + ctx], ctx = 1
        const int vNegTb = 4 - vw;                                  // -tbBase (tbBase = address of probs[tb])
        int ca, c0, c1, pr;
        FPW_DEC_OPEN();
        FPW_DEC_BIT2(0) FPW_DEC_BIT2(1) FPW_DEC_BIT2(2) FPW_DEC_BIT2(3) FPW_DEC_BIT2(4) FPW_DEC_BIT2(5) FPW_DEC_BIT2(6) FPW_DEC_BIT2(7)
        const u32 cb = (u32)__builtin_amdgcn_readfirstlane((vw + vNegTb) >> 2);   // 256 + the byte (v_writelane takes bits 7..0 below)
        outv = (lane == j) ? cb : outv;                                // (no inline v_writelane: its lane select needs m0, which inline asm may not clobber)
        if (__builtin_expect(idx > szBytes, 0)) { bad = true; break; }   // :231-232
      }
      if (!bad && lane < rowCnt) o[i + lane] = (u8)outv;
    }
    ipos += szBytes;
    startChunk += chunkSize;
  }
  if (endOut && lane == 0) endOut[b] = (long long)(d_bitOff[b] + 8LL * ipos);
  if (bad && lane == 0) d_flag[b] = 0;
}


// which arrangement: one wave per block up to 8 blocks per CU, one lane per block above (KZ_FPAQ_FORCE=lanes|waves
// overrides, for tests)
static bool fpaq_use_waves(const kz_ctx* ctx, int B) {
  if (ctx->sw.fpaqForce == 2) return false;
  if (ctx->sw.fpaqForce == 1) return true;
  return B <= 8 * ctx->numCUs;
}

size_t kz_fpaq_scratch(int B, int maxN) {
  const int maxChunks = (maxN + FP_CHUNK - 1) / FP_CHUNK + 1;
  return (size_t)B * (kz_align((size_t)maxN + (size_t)(maxN >> 3) + 64, 256) + (size_t)maxChunks * 12 + 512) + 4096;
}

int kz_stage_fpaq_encode(kz_ctx* ctx, kz_batch& bt, uint8_t* out, int64_t outStride, const int32_t* d_hdrBytes, int64_t* d_bits) {
  const int B = bt.B;
  int maxN = 0;
  for (int b = 0; b < B; b++) if (bt.h_len[b] > maxN) maxN = bt.h_len[b];
  const int64_t scrStride = (int64_t)kz_align((size_t)maxN + (size_t)(maxN >> 3) + 64, 256);
  u8* scr = (u8*)kz_arena_alloc(ctx, (size_t)scrStride * B);
  FpaqChunks C;
  C.maxChunks = (maxN + FP_CHUNK - 1) / FP_CHUNK + 1;
  C.bytes = (u32*)kz_arena_alloc(ctx, (size_t)B * C.maxChunks * 4);
  C.tail = (u64*)kz_arena_alloc(ctx, (size_t)B * C.maxChunks * 8);
  if (!scr || !C.bytes || !C.tail) { snprintf(ctx->err, sizeof(ctx->err), "fpaq_encode: arena overflow"); return -KZ_ERR_DEVICE; }
  if (fpaq_use_waves(ctx, B)) {
    // one wave per block; the cost of a block is its length
    kz_batch view = bt;
    view.h_cost = bt.h_len;
    KzPlacement PL;
    { const int prc = kz_place_blocks(ctx, view, PL); if (prc) return prc; }
    for (int rr = 0; rr < PL.R; rr++)
      KZ_LAUNCH(ctx, KID_FPAQ_ENC, k_fpaq_enc_wave, dim3(PL.G[rr]), dim3(64 * PL.wpg), bt.buf[bt.cur], bt.stride, bt.d_len, scr, scrStride, out, outStride,
                d_hdrBytes, d_bits, PL.d_order + PL.off[rr], PL.wpg);
  } else {
    KZ_LAUNCH(ctx, KID_FPAQ_ENC, k_fpaq_enc, dim3((B + 63) / 64), dim3(64), bt.buf[bt.cur], bt.stride, bt.d_len, scr, scrStride, C, B);
    KZ_LAUNCH(ctx, KID_FPAQ_PACK, k_fpaq_pack, dim3(B), dim3(256), bt.d_len, scr, scrStride, C, out, outStride, d_hdrBytes, d_bits);
  }
  KZ_HIP(hipGetLastError());
  return 0;
}

int kz_stage_fpaq_decode(kz_ctx* ctx, kz_batch& bt, const uint8_t* in, int64_t inStride, const int64_t* d_bitOff, const int64_t* d_bitEnd) {
  const int B = bt.B;
  u8* dst = bt.buf[bt.cur ^ 1];
  if (fpaq_use_waves(ctx, B)) {
    kz_batch view = bt;
    view.h_cost = bt.h_len;                                         // decoded length ~ coding steps
    KzPlacement PL;
    { const int prc = kz_place_blocks(ctx, view, PL); if (prc) return prc; }
    for (int rr = 0; rr < PL.R; rr++) {
      KZ_LAUNCH(ctx, KID_FPAQ_DEC, k_fpaq_dec_wave2, dim3(PL.G[rr]), dim3(64 * PL.wpg), in, inStride, d_bitOff, d_bitEnd, bt.d_len, dst, bt.stride,
                  bt.d_len2, bt.d_flag, PL.d_order + PL.off[rr], PL.wpg, ctx->d_endBits);
    }
  } else {
    KZ_LAUNCH(ctx, KID_FPAQ_DEC, k_fpaq_dec, dim3((B + 63) / 64), dim3(64), in, inStride, d_bitOff, d_bitEnd, bt.d_len, dst, bt.stride, bt.d_len2, bt.d_flag, B, ctx->d_endBits);
  }
  KZ_HIP(hipGetLastError());
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}
