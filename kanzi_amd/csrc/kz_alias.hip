// kz_alias.hip -- AliasCodec (transforms PACK = 18 and DNA = 19) for a batch of blocks on gfx950.
//
// Replaces K/transform/AliasCodec.java:76-279 (forward), :289-470 (inverse), :472-475 (getMaxEncodedLength) with
// K/Global.java:341-420 (computeHistogramOrder1) and :556-605 (detectSimpleType).  DNA is the same codec restricted to
// DNA-looking blocks (TransformFactory.java:341-343).
//
// forward:  k_alias_analyze (one workgroup per block: order-0 histogram in LDS, absent symbols, the "dataType" rules,
//   which of the four codings applies) -> for the digram coding k_alias_hist1 (pair histogram, 65536 bins per block,
//   counted in LDS one half at a time) and k_alias_select (the n0 most frequent pairs in the reference's TreeSet order = descending
//   (frequency << 16 | pair), picked one by one with a workgroup max; header; alias table) -> k_alias_emit (one wave
//   per block).  Bit packing is position independent.  The digram parse is greedy -- a pair consumes two bytes, so
//   whether position i starts a token depends on i-1 -- but the dependence is a parity: i starts a token iff an even
//   number of aliased pairs start consecutively right before it, which one ballot and a count-leading-ones give for
//   64 positions at a time.
// inverse:  one wave per block; every source byte expands to a fixed (packing) or table-driven (1 or 2 bytes) number of
//   output bytes, offsets by a wave prefix sum per row of 64.
#include "kz_device.h"
#include "kz_internal.h"
#include "kz_datatype.h"

typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t u8;

#define AL_MIN_BLOCK 1024
#define AL_DECLINE 0
#define AL_ONE 1
#define AL_2BIT 2
#define AL_4BIT 3
#define AL_DIGRAM 4

struct AliasFwd {
  int32_t* branch;   // [B] AL_*
  int32_t* n0;       // [B] absent symbols (aliases available)
  u8* absent;        // [B][256] absent symbols, ascending
  u8* map8;          // [B][256] symbol -> index among the present symbols
  u32* freqs1;       // [B][65536] pair histogram, later the alias table (0x200 | alias for aliased pairs, else 0)
};

__global__ __launch_bounds__(256) void k_alias_analyze(const u8* __restrict__ srcAll, int64_t stride, const int32_t* __restrict__ d_len,
                                                        int32_t* __restrict__ d_dtype, AliasFwd A, int onlyDNA) {
  const int b = blockIdx.x;
  const int count = d_len[b];
  const int tid = threadIdx.x;
  __shared__ int h[256];
  __shared__ long long lds4[4];
  __shared__ u32 scan[32];
  if (tid == 0) { A.branch[b] = AL_DECLINE; A.n0[b] = 0; }
  if (count < AL_MIN_BLOCK) return;                                              // :88-89 (count == 0: caller)
  int dt = d_dtype[b];
  if (dt == DT_MULTIMEDIA || dt == DT_UTF8 || dt == DT_EXE || dt == DT_BIN) return;                 // :103-109
  if (onlyDNA && dt != DT_UNDEFINED && dt != DT_DNA) return;                                        // :111-113
  const u8* src = srcAll + (int64_t)b * stride;
  h[tid] = 0;
  __syncthreads();
  for (int row = 0; row < count; row += 256) {                                    // order-0 histogram: one LDS add per group of equal bytes
    const int i = row + tid;
    const bool valid = i < count;
    const u32 v = valid ? (u32)src[i] : 0u;
    const uint64_t peers = kz_match8(v, valid);
    if (valid && (peers & kz_lanemask_lt()) == 0) atomicAdd(&h[v], (int)__popcll(peers));
  }
  __syncthreads();
  const int f = h[tid];
  // absent symbols in ascending order: exclusive scan of (f == 0)
  u32 total;
  const u32 pos = kz_wg_excl_sum(f == 0 ? 1u : 0u, scan, &total);
  const int n0 = (int)total;
  if (f == 0) A.absent[(int64_t)b * 256 + pos] = (u8)tid;
  else A.map8[(int64_t)b * 256 + tid] = (u8)((u32)tid - pos);                     // index among the present symbols
  if (n0 < 16) return;                                                            // :129-130
  if (dt == DT_UNDEFINED) {                                                       // :133-141
    dt = kz_detect_simple_type_wg(count, f, h[0x3D], lds4);
    if (tid == 0 && dt != DT_UNDEFINED) d_dtype[b] = dt;
    if (dt != DT_DNA && onlyDNA) return;
  }
  if (tid == 0) {
    A.n0[b] = n0;
    A.branch[b] = (n0 == 255) ? AL_ONE : (n0 >= 252) ? AL_2BIT : (n0 >= 240) ? AL_4BIT : AL_DIGRAM;
  }
}

// pair histogram (Global.computeHistogramOrder1: pair = (previous byte, byte); the first byte's previous is 0).
// 65536 bins do not fit LDS as 32-bit counters, half of them do (128 KiB): two workgroups per block, each reads the
// whole block and counts the pairs whose first byte falls in its half, then stores its 32768 counters.  (Global
// atomics on 64K bins per block took 206 ms for 1024 text blocks of 4 MiB, this takes a few.)
__global__ __launch_bounds__(1024) void k_alias_hist1(const u8* __restrict__ srcAll, int64_t stride, const int32_t* __restrict__ d_len, AliasFwd A) {
  const int b = blockIdx.y;
  if (A.branch[b] != AL_DIGRAM) return;
  const int half = blockIdx.x;
  const int count = d_len[b];
  __shared__ u32 hist[32768];
  for (int i = threadIdx.x; i < 32768; i += 1024) hist[i] = 0;
  __syncthreads();
  const u8* src = srcAll + (int64_t)b * stride;
  for (int i = threadIdx.x; i < count; i += 1024) {
    const u32 prv = (i == 0) ? 0u : (u32)src[i - 1];
    if ((int)(prv >> 7) == half) atomicAdd(&hist[((prv & 127u) << 8) | (u32)src[i]], 1u);
  }
  __syncthreads();
  u32* fr = A.freqs1 + (int64_t)b * 65536 + half * 32768;
  for (int i = threadIdx.x; i < 32768; i += 1024) fr[i] = hist[i];
}

__device__ __forceinline__ unsigned long long al_wg_max64(unsigned long long v, unsigned long long* lds4) {
  for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long m = lds4[0];
  for (int k = 1; k < 4; k++) m = lds4[k] > m ? lds4[k] : m;
  return m;
}

// the n0 most frequent pairs, TreeSet order (:203-245), header, savings test, alias table
__global__ __launch_bounds__(256) void k_alias_select(u8* __restrict__ dstAll, int64_t stride, const int32_t* __restrict__ d_len, AliasFwd A) {
  const int b = blockIdx.x;
  if (A.branch[b] != AL_DIGRAM) return;
  const int count = d_len[b];
  const int tid = threadIdx.x;
  __shared__ unsigned long long lds4[4];
  __shared__ long long lsum[4];
  __shared__ u32 chosen[256];
  u32* fr = A.freqs1 + (int64_t)b * 65536;
  u8* dst = dstAll + (int64_t)b * stride;
  const u8* absent = A.absent + (int64_t)b * 256;
  int n0 = A.n0[b];
  long long nz = 0;
  for (int k = 0; k < 256; k++) nz += fr[k * 256 + tid] != 0 ? 1 : 0;
  const int n1 = (int)kz_wg256_sum64(nz, lsum);
  if (n1 < n0) {                                                                  // :215-221
    n0 = n1;
    if (n0 < 16) { if (tid == 0) A.branch[b] = AL_DECLINE; return; }
  }
  unsigned long long prev = ~0ULL;
  long long savings = 0;
  for (int i = 0; i < n0; i++) {
    unsigned long long best = 0;
    for (int k = 0; k < 256; k++) {
      const u32 idx = (u32)(k * 256 + tid);
      const u32 f = fr[idx];
      const unsigned long long key = ((unsigned long long)f << 16) | idx;
      if (f != 0 && key < prev && key > best) best = key;
    }
    best = al_wg_max64(best, lds4);
    prev = best;
    savings += (long long)(best >> 16);
    if (tid == 0) chosen[i] = (u32)(best & 0xFFFFu);
  }
  __syncthreads();
  if (tid == 0) { dst[0] = (u8)n0; dst[1] = 0; A.n0[b] = n0; }
  if (tid < n0) {
    const u32 idx = chosen[tid];
    dst[2 + 3 * tid] = (u8)(idx >> 8); dst[2 + 3 * tid + 1] = (u8)idx; dst[2 + 3 * tid + 2] = absent[tid];
  }
  if (savings < count / 20) { if (tid == 0) A.branch[b] = AL_DECLINE; return; }   // :247-249
  for (int k = 0; k < 256; k++) fr[k * 256 + tid] = 0;
  __syncthreads();
  if (tid < n0) fr[chosen[tid]] = 0x200u | (u32)absent[tid];
}

__global__ __launch_bounds__(64) void k_alias_emit(const u8* __restrict__ srcAll, u8* __restrict__ dstAll, int64_t stride,
                                                    const int32_t* __restrict__ d_len, int32_t* __restrict__ d_len2, int32_t* __restrict__ d_flag, AliasFwd A) {
  const int b = blockIdx.x;
  const int count = __builtin_amdgcn_readfirstlane(d_len[b]);
  const int lane = kz_lane();
  if (count == 0) { if (lane == 0) { d_len2[b] = 0; d_flag[b] = 1; } return; }
  const int branch = __builtin_amdgcn_readfirstlane(A.branch[b]);
  if (branch == AL_DECLINE) { if (lane == 0) { d_len2[b] = count; d_flag[b] = 0; } return; }
  const u8* src = srcAll + (int64_t)b * stride;
  u8* dst = dstAll + (int64_t)b * stride;
  const int n0 = __builtin_amdgcn_readfirstlane(A.n0[b]);
  const u8* map8 = A.map8 + (int64_t)b * 256;
  int dstIdx = 0;
  if (branch == AL_ONE) {                                                         // :146-152
    if (lane == 0) { dst[0] = (u8)n0; dst[1] = src[0]; dst[2] = (u8)count; dst[3] = (u8)(count >> 8); dst[4] = (u8)(count >> 16); dst[5] = (u8)(count >> 24); }
    dstIdx = 6;
  } else if (branch == AL_2BIT || branch == AL_4BIT) {
    const int present = 256 - n0;
    if (lane == 0) dst[0] = (u8)n0;
    // present symbols in ascending order: symbol s sits at 1 + map8[s]
    for (int s = lane; s < 256; s += 64) {
      bool isAbsent = false;                                                      // s is present iff it is not in the absent list: map8 was written for present symbols only
      const u8* absent = A.absent + (int64_t)b * 256;
      // binary search in the ascending absent list
      int lo = 0, hi = n0;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (absent[mid] < (u8)s) lo = mid + 1; else hi = mid; }
      isAbsent = lo < n0 && absent[lo] == (u8)s;
      if (!isAbsent) dst[1 + map8[s]] = (u8)s;
    }
    const int per = (branch == AL_2BIT) ? 4 : 2;
    const int adjust = count & (per - 1);
    const int hdr = 1 + present;
    if (lane == 0) dst[hdr] = (u8)adjust;
    if (lane < adjust) dst[hdr + 1 + lane] = src[lane];
    const int base = hdr + 1 + adjust;
    const int nOut = (count - adjust) / per;
    for (int j = lane; j < nOut; j += 64) {
      const u8* p = src + adjust + j * per;
      u32 v;
      if (per == 4) v = ((u32)map8[p[0]] << 6) | ((u32)map8[p[1]] << 4) | ((u32)map8[p[2]] << 2) | (u32)map8[p[3]];
      else v = ((u32)map8[p[0]] << 4) | (u32)map8[p[1]];
      dst[base + j] = (u8)v;
    }
    dstIdx = base + nOut;
  } else {                                                                        // digram aliases :251-268
    const u32* tab = A.freqs1 + (int64_t)b * 65536;
    dstIdx = 2 + 3 * n0;
    const int srcEnd = count - 1;
    bool startRow = true;                                                          // position 0 starts a token
    bool lastIsStart = false;
    for (int row = 0; row < count; row += 64) {
      const int i = row + lane;
      const bool inPair = i < srcEnd;                                              // a pair starts here only below the last byte
      u32 a = 0;
      if (inPair) a = tab[((u32)src[i] << 8) | (u32)src[i + 1]];
      const bool al = a != 0;
      const uint64_t Am = kz_ballot(al);
      const uint64_t below = Am & kz_lanemask_lt();
      const int run = (lane == 0) ? 0 : min((int)__builtin_clzll(~(below << (64 - lane))), lane);
      bool start = (run & 1) == 0;                                                 // even number of aliased pairs right before
      if (run == lane) start = ((lane & 1) == 0) == startRow;                      // the run reaches the row start
      const bool tok = start && i < count;
      const bool emit = tok && i < srcEnd;                                         // the last byte is handled below
      const uint64_t Em = kz_ballot(emit);
      const int at = dstIdx + (int)__popcll(Em & kz_lanemask_lt());
      if (emit) dst[at] = al ? (u8)a : src[i];
      dstIdx += (int)__popcll(Em);
      // does the next row's first position start a token?  start(64) = !(start(63) && aliased(63))
      const bool s63 = __builtin_amdgcn_readlane((int)start, 63) != 0;
      const bool a63 = (Am >> 63) & 1ULL;
      startRow = !(s63 && a63);
      if (kz_ballot(tok && i == srcEnd)) lastIsStart = true;
    }
    if (lastIsStart) {                                                             // :263-266
      if (lane == 0) { dst[1] = 1; dst[dstIdx] = src[count - 1]; }
      dstIdx++;
    }
  }
  if (lane == 0) { const bool applied = dstIdx < count; d_len2[b] = applied ? dstIdx : count; d_flag[b] = applied ? 1 : 0; }   // :276
}

// ---- inverse: one wave per block ----
__global__ __launch_bounds__(64) void k_alias_inv(const u8* __restrict__ srcAll, u8* __restrict__ dstAll, int64_t stride,
                                                   const int32_t* __restrict__ d_len, int32_t* __restrict__ d_len2, int32_t* __restrict__ d_flag, int dstCap) {
  const int b = blockIdx.x;
  const int count = __builtin_amdgcn_readfirstlane(d_len[b]);
  const int lane = kz_lane();
  if (count == 0) { if (lane == 0) { d_len2[b] = 0; d_flag[b] = 1; } return; }
  const u8* src = srcAll + (int64_t)b * stride;
  u8* dst = dstAll + (int64_t)b * stride;
  __shared__ u32 map16[256];
  bool ok = true;
  int produced = 0;
  int n = __builtin_amdgcn_readfirstlane((int)src[0]);
  if (n < 16) ok = false;                                                          // :297-298
  else if (n >= 240) {
    n = 256 - n;
    if (n == 1) {                                                                   // :303-316
      if (count < 6) ok = false;
      else {
        const u32 val = (u32)src[1];
        const int oSize = __builtin_amdgcn_readfirstlane((int)((u32)src[2] | ((u32)src[3] << 8) | ((u32)src[4] << 16) | ((u32)src[5] << 24)));
        if (oSize < 0 || oSize > dstCap) ok = false;
        else { for (int i = lane; i < oSize; i += 64) dst[i] = (u8)val; produced = oSize; }
      }
    } else if (1 + n + 1 > count) ok = false;
    else {
      const u8* idx2symb = src + 1;                                                 // n bytes; indexes >= n map to 0 (a fresh byte[16])
      const int adjust = __builtin_amdgcn_readfirstlane((int)src[1 + n]);
      int srcIdx = 2 + n;
      if (adjust >= 4) ok = false;                                                  // :328-329
      else {
        const int per = (n <= 4) ? 4 : 2;
        const int raw = (n <= 4) ? adjust : (adjust != 0 ? 1 : 0);                 // :381-382: one raw byte whatever the value
        if (srcIdx + raw > count) ok = false;
        else if ((long long)raw + (long long)per * (count - srcIdx - raw) > (long long)dstCap) ok = false;
        else {
          if (lane < raw) dst[lane] = src[srcIdx + lane];
          srcIdx += raw;
          const int nIn = count - srcIdx;
          for (int j = lane; j < nIn; j += 64) {
            const u32 v = src[srcIdx + j];
            u8* o = dst + raw + (int64_t)j * per;
            if (per == 4) {
              const u32 i0 = (v >> 6) & 3, i1 = (v >> 4) & 3, i2 = (v >> 2) & 3, i3 = v & 3;
              o[0] = (i0 < (u32)n) ? idx2symb[i0] : (u8)0; o[1] = (i1 < (u32)n) ? idx2symb[i1] : (u8)0;
              o[2] = (i2 < (u32)n) ? idx2symb[i2] : (u8)0; o[3] = (i3 < (u32)n) ? idx2symb[i3] : (u8)0;
            } else {
              const u32 i0 = v >> 4, i1 = v & 15;
              o[0] = (i0 < (u32)n) ? idx2symb[i0] : (u8)0; o[1] = (i1 < (u32)n) ? idx2symb[i1] : (u8)0;
            }
          }
          produced = raw + per * nIn;
        }
      }
    }
  } else if (2 + 3 * n > count) ok = false;
  else {                                                                            // digram aliases :391-440
    const int adjust = __builtin_amdgcn_readfirstlane((int)src[1]);
    const int srcEnd = count - adjust;
    for (int i = lane; i < 256; i += 64) map16[i] = 0x10000u | (u32)i;
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) for (int i = 0; i < n; i++) { const u8* e = src + 2 + 3 * i; map16[e[2]] = 0x20000u | (u32)e[0] | ((u32)e[1] << 8); }   // later entries win
    __builtin_amdgcn_wave_barrier();
    int srcIdx = 2 + 3 * n;
    int dstIdx = 0;
    for (int row = srcIdx; row < srcEnd && ok; row += 64) {
      const int i = row + lane;
      const bool valid = i < srcEnd;
      const u32 val = valid ? map16[src[i]] : 0u;
      const u32 inc = val >> 16;
      const u32 incl = kz_wave_incl_sum(inc);
      const int total = (int)__shfl((int)incl, 63, 64);
      if (dstIdx + total > dstCap) { ok = false; break; }                           // some token would end past the array
      const int at = dstIdx + (int)(incl - inc);
      if (valid) { dst[at] = (u8)val; if (inc == 2) dst[at + 1] = (u8)(val >> 8); }
      dstIdx += total;
    }
    if (ok) srcIdx = max(srcIdx, srcEnd);
    if (ok && adjust != 0) {                                                        // :433-439
      if (dstIdx >= dstCap || srcIdx >= count || srcIdx < 0) ok = false;
      else { if (lane == 0) dst[dstIdx] = src[srcIdx]; dstIdx++; }
    }
    produced = dstIdx;
  }
  if (lane == 0) { d_len2[b] = ok ? produced : 0; d_flag[b] = ok ? 1 : 0; }
}

size_t kz_alias_scratch(int B, int, bool decode) { return decode ? 4096 : (size_t)B * (65536 * 4 + 512 + 16) + 8192; }

int kz_stage_alias_forward(kz_ctx* ctx, kz_batch& bt, int onlyDNA) {
  const int B = bt.B;
  int maxN = 0;
  for (int b = 0; b < B; b++) if (bt.h_len[b] > maxN) maxN = bt.h_len[b];
  AliasFwd A;
  A.branch = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  A.n0 = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  A.absent = (u8*)kz_arena_alloc(ctx, (size_t)B * 256);
  A.map8 = (u8*)kz_arena_alloc(ctx, (size_t)B * 256);
  A.freqs1 = (u32*)kz_arena_alloc(ctx, (size_t)B * 65536 * 4);
  if (!A.freqs1 || !A.map8 || !A.branch) { snprintf(ctx->err, sizeof(ctx->err), "alias_forward: arena overflow"); return -KZ_ERR_DEVICE; }
  hipStream_t st = ctx->stream;
  const u8* src = bt.buf[bt.cur];
  u8* dst = bt.buf[bt.cur ^ 1];
  KZ_HIP(hipMemsetAsync(A.map8, 0, (size_t)B * 256, st));
  KZ_LAUNCH(ctx, KID_ALIAS_ANALYZE, k_alias_analyze, dim3(B), dim3(256), src, bt.stride, bt.d_len, bt.d_dtype, A, onlyDNA);
  if (maxN > 0) KZ_LAUNCH(ctx, KID_ALIAS_HIST1, k_alias_hist1, dim3(2, B), dim3(1024), src, bt.stride, bt.d_len, A);
  KZ_LAUNCH(ctx, KID_ALIAS_SELECT, k_alias_select, dim3(B), dim3(256), dst, bt.stride, bt.d_len, A);
  KZ_LAUNCH(ctx, KID_ALIAS_EMIT, k_alias_emit, dim3(B), dim3(64), src, dst, bt.stride, bt.d_len, bt.d_len2, bt.d_flag, A);
  KZ_HIP(hipGetLastError());
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}

int kz_stage_alias_inverse(kz_ctx* ctx, kz_batch& bt, int dstCap) {
  const int B = bt.B;
  if ((int64_t)dstCap > bt.stride) dstCap = (int)bt.stride;
  KZ_LAUNCH(ctx, KID_ALIAS_INV, k_alias_inv, dim3(B), dim3(64), bt.buf[bt.cur], bt.buf[bt.cur ^ 1], bt.stride, bt.d_len, bt.d_len2, bt.d_flag, dstCap);
  KZ_HIP(hipGetLastError());
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}
