// kz_fpaq.hip -- FPAQ adaptive order-0 binary arithmetic coder on gfx950.
//
// Replaces K/entropy/FPAQEncoder.java:128-173 (encode), :182-199 (encodeBit), :208-213 (flush),
// :232-238 (dispose) and K/entropy/FPAQDecoder.java:161-242, :290-314 (decodeBitV2), :322-335 (read).
//
// The coder state (56-bit low/high, 4 x 256 16-bit probabilities) is carried across the whole block
// and every bit depends on the previous one, so the only parallelism is across blocks:
// one wave per block, all lanes execute the same (uniform) control flow, the 4 KiB probability table
// lives in LDS, lane 0 performs the stores.  The output of a block is a byte string
// [varint(n) | n bytes | 56-bit tail]* so it is written byte aligned behind the block header.
#include "kz_device.h"
#include "kz_internal.h"

typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint8_t u8;

#define FP_TOP 0x00FFFFFFFFFFFFFFULL
#define FP_M2456 0x00FFFFFFFF000000ULL
#define FP_M024 0x0000000000FFFFFFULL
#define FP_M032 0x00000000FFFFFFFFULL
#define FP_M056 0x00FFFFFFFFFFFFFFULL
#define FP_CHUNK (4 * 1024 * 1024)
#define FP_PSCALE 65536

// wave-cooperative byte copy (all lanes must call)
__device__ __forceinline__ void fp_copy(u8* __restrict__ d, const u8* __restrict__ s, int n) {
  for (int i = kz_lane(); i < n; i += 64) d[i] = s[i];
}

// One range-coder step (FPAQEncoder.java:182-199 encodeBit + :208-213 flush), all operands wave-uniform.
// Written branch-free (selects) except for the rare flush: a taken scalar branch costs a lone wave ~40 cycles.
#define FP_ENC_BIT(PP, BIT)                                                                    \
  { const u64 split = (((high - low) >> 8) * (u64)(u32)(PP)) >> 8;                             \
    const bool one = (BIT) != 0;                                                               \
    const u64 nh = low + split, nl = nh + 1;                                                   \
    high = one ? nh : high; low = one ? low : nl;                                              \
    while (__builtin_expect(((low ^ high) & FP_M2456) == 0, 0)) {                              \
      if (lane == 0) { const u32 w = (u32)(high >> 24); sba[idx] = (u8)(w >> 24); sba[idx + 1] = (u8)(w >> 16); sba[idx + 2] = (u8)(w >> 8); sba[idx + 3] = (u8)w; } \
      idx += 4;                                                                                \
      low <<= 32;                                                                              \
      high = (high << 32) | FP_M032;                                                           \
    } }

// Encoder: the 8 contexts of a byte are known up front (the byte is known) and are 8 distinct table entries,
// and the probability update does not depend on the coder state: lanes 0..7 gather, update and write back the
// 8 probabilities of a byte with ONE LDS read and ONE LDS write; the gather for the next byte is issued before
// the 8 sequential range-coder steps of the current one, which run on the scalar unit.
__global__ __launch_bounds__(64) void k_fpaq_enc(const u8* __restrict__ src, int64_t stride, const int32_t* __restrict__ d_len,
                                                  u8* __restrict__ scr, int64_t scrStride, u8* __restrict__ out, int64_t outStride,
                                                  const int32_t* __restrict__ d_hdrBytes, int64_t* __restrict__ d_bits) {
  const int b = blockIdx.x;
  const int count = d_len[b];
  const int lane = kz_lane();
  if (count <= 0) { if (lane == 0) d_bits[b] = 0; return; }
  __shared__ int probs[1024];
  for (int i = lane; i < 1024; i += 64) probs[i] = FP_PSCALE >> 1;
  __syncthreads();
  const u8* blk = src + (int64_t)b * stride;
  u8* sba = scr + (int64_t)b * scrStride;
  u8* o = out + (int64_t)b * outStride + d_hdrBytes[b];
  int opos = 0;
  u64 low = 0, high = FP_TOP;
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
    for (int i = startChunk; i < chunkEnd; i++) {
      // update and write back this byte's 8 probabilities (lanes 0..7), fetch the next byte's
      const int bit = (val >> kbit) & 1;
      const int np = bit ? pp - ((pp - FP_PSCALE + 64) >> 6) : pp - (pp >> 6);
      if (lane < 8) probs[pIdx] = np;
      const int cur = pp;
      const int curVal = val;
      const int j1 = (i + 1 - startChunk) & 63;
      if (i + 1 < chunkEnd) {
        if (j1 == 0) rowv = (i + 1 + lane < chunkEnd) ? (u32)blk[i + 1 + lane] : 0u;   // 64 bytes per load
        val = __builtin_amdgcn_readlane((int)rowv, j1);
        pIdx = ((curVal >> 6) << 8) + ((kbit == 7) ? 1 : ((val + 256) >> (kbit + 1)));    // :161
        pp = probs[pIdx];
      }
      const int p7 = __builtin_amdgcn_readlane(cur, 0), p6 = __builtin_amdgcn_readlane(cur, 1), p5 = __builtin_amdgcn_readlane(cur, 2),
                p4 = __builtin_amdgcn_readlane(cur, 3), p3 = __builtin_amdgcn_readlane(cur, 4), p2 = __builtin_amdgcn_readlane(cur, 5),
                p1 = __builtin_amdgcn_readlane(cur, 6), p0 = __builtin_amdgcn_readlane(cur, 7);
      FP_ENC_BIT(p7, curVal & 0x80) FP_ENC_BIT(p6, curVal & 0x40) FP_ENC_BIT(p5, curVal & 0x20) FP_ENC_BIT(p4, curVal & 0x10)
      FP_ENC_BIT(p3, curVal & 0x08) FP_ENC_BIT(p2, curVal & 0x04) FP_ENC_BIT(p1, curVal & 0x02) FP_ENC_BIT(p0, curVal & 0x01)
    }
    // varint(idx) | bytes   (EntropyUtils.writeVarInt; :164-165)
    { u32 v = (u32)idx; while (v >= 128) { if (lane == 0) o[opos] = (u8)(0x80 | (v & 0x7F)); opos++; v >>= 7; } if (lane == 0) o[opos] = (u8)v; opos++; }
    __syncthreads();
    fp_copy(o + opos, sba, idx);
    opos += idx;
    startChunk += chunkSize;
    if (startChunk < count) {                                       // :168-169
      const u64 t = low | FP_M024;
      if (lane < 7) o[opos + lane] = (u8)(t >> (8 * (6 - lane)));
      opos += 7;
    }
    __syncthreads();
  }
  { const u64 t = low | FP_M024; if (lane < 7) o[opos + lane] = (u8)(t >> (8 * (6 - lane))); opos += 7; }   // dispose :232-238
  if (lane == 0) d_bits[b] = 8LL * opos;
}

// One decoder step (FPAQDecoder.java:290-314 decodeBitV2 + :322-335 read).  PR = probability of the current
// context (scalar); the two children of the context were fetched from LDS one level earlier.
#define FP_DEC_BIT(LEVEL)                                                                      \
  { int2 ch = make_int2(0, 0);                                                                 \
    if (LEVEL < 7) ch = *(const int2*)&probs[tb + 2 * ctx];        /* children of ctx: used at the next level */ \
    const u64 split = ((((high - low) >> 8) * (u64)(u32)pr) >> 8) + low;                       \
    const bool one = (int)((split - current) >> 32) >= 0;           /* split >= current (both < 2^56): sign of a scalar subtract */ \
    const int np = one ? pr - ((pr - FP_PSCALE + 64) >> 6) : pr - (pr >> 6);                   \
    high = one ? split : high; low = one ? low : split + 1;                                    \
    probs[tb + ctx] = np;                                                                      \
    ctx = (ctx << 1) + (one ? 1 : 0);                                                          \
    if (LEVEL == 1) rootNext = probs[((ctx & 3) << 8) + 1];        /* next byte's first context (:233-239) */ \
    while (__builtin_expect(((low ^ high) & FP_M2456) == 0, 0)) {                              \
      low = (low << 32) & FP_M056;                                                             \
      high = ((high << 32) | FP_M032) & FP_M056;                                               \
      if (idx + 4 > bufLimit) { current = (current << 32) & FP_M056; idx = bufLimit + 1; continue; } \
      if (idx >= wbase + 256) { wbase = idx; const u8* q = buf + wbase + 4 * lane; win = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3]; } \
      const u64 val = (u64)(u32)__builtin_amdgcn_readlane((int)win, (idx - wbase) >> 2);       \
      current = ((current << 32) | val) & FP_M056;                                             \
      idx += 4;                                                                                \
    }                                                                                          \
    if (LEVEL < 7) pr = __builtin_amdgcn_readfirstlane(one ? ch.y : ch.x); }

// Decoder: every bit's context depends on the previous bit, so the chain is serial; what can be hidden is the
// LDS latency of the probability: both children of the current context (adjacent ints) are fetched with one
// 8-byte LDS read while the current bit is decoded, the next byte's first context after its top two bits are
// known.  Range arithmetic runs on the scalar unit; output bytes are collected with v_writelane and stored
// 64 at a time.
__global__ __launch_bounds__(64) void k_fpaq_dec(const u8* __restrict__ in, int64_t inStride, const int64_t* __restrict__ d_bitOff,
                                                  const int64_t* __restrict__ d_bitEnd, const int32_t* __restrict__ d_len,
                                                  u8* __restrict__ dst, int64_t stride, int32_t* __restrict__ d_len2, int32_t* __restrict__ d_flag) {
  const int b = blockIdx.x;
  const int count = d_len[b];
  const int lane = kz_lane();
  if (lane == 0) { d_len2[b] = count; d_flag[b] = 1; }
  if (count <= 0) return;
  __shared__ __attribute__((aligned(8))) int probs[1024];
  for (int i = lane; i < 1024; i += 64) probs[i] = FP_PSCALE >> 1;
  __syncthreads();
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
    if (szBytes < 0 || szBytes >= 2 * count || ipos + 7 + szBytes > avail + 8) { bad = true; break; }   // :176-177
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
    int tb = 0;
    int rootNext = probs[1];
    u32 outv = 0;
    for (int i = startChunk; i < startChunk + chunkSize; i++) {
      int ctx = 1;
      int pr = __builtin_amdgcn_readfirstlane(rootNext);
      FP_DEC_BIT(0) FP_DEC_BIT(1) FP_DEC_BIT(2) FP_DEC_BIT(3) FP_DEC_BIT(4) FP_DEC_BIT(5) FP_DEC_BIT(6) FP_DEC_BIT(7)
      const int j = (i - startChunk) & 63;
      { const u32 cb = (u32)__builtin_amdgcn_readfirstlane(ctx & 0xFF); asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(outv) : "s"(cb), "s"(j) : "m0"); }
      if (j == 63 || i + 1 == startChunk + chunkSize) { if (lane <= j) o[i - j + lane] = (u8)outv; }
      if (idx > szBytes) { bad = true; break; }                      // :231-232
      tb = ((ctx & 0xFF) >> 6) << 8;
    }
    ipos += szBytes;
    startChunk += chunkSize;
  }
  if (bad && lane == 0) d_flag[b] = 0;
}

int kz_stage_fpaq_encode(kz_ctx* ctx, kz_batch& bt, uint8_t* out, int64_t outStride, const int32_t* d_hdrBytes, int64_t* d_bits) {
  const int B = bt.B;
  int maxN = 0;
  for (int b = 0; b < B; b++) if (bt.h_len[b] > maxN) maxN = bt.h_len[b];
  const int64_t scrStride = (int64_t)kz_align((size_t)maxN + (size_t)(maxN >> 3) + 64, 256);
  u8* scr = (u8*)kz_arena_alloc(ctx, (size_t)scrStride * B);
  if (!scr) { snprintf(ctx->err, sizeof(ctx->err), "fpaq_encode: arena overflow"); return -KZ_ERR_DEVICE; }
  KZ_LAUNCH(ctx, KID_FPAQ_ENC, k_fpaq_enc, dim3(B), dim3(64), bt.buf[bt.cur], bt.stride, bt.d_len, scr, scrStride, out, outStride, d_hdrBytes, d_bits);
  KZ_HIP(hipGetLastError());
  return 0;
}

int kz_stage_fpaq_decode(kz_ctx* ctx, kz_batch& bt, const uint8_t* in, int64_t inStride, const int64_t* d_bitOff, const int64_t* d_bitEnd) {
  const int B = bt.B;
  u8* dst = bt.buf[bt.cur ^ 1];
  KZ_LAUNCH(ctx, KID_FPAQ_DEC, k_fpaq_dec, dim3(B), dim3(64), in, inStride, d_bitOff, d_bitEnd, bt.d_len, dst, bt.stride, bt.d_len2, bt.d_flag);
  KZ_HIP(hipGetLastError());
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}
