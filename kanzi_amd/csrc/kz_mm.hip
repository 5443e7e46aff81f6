// kz_mm.hip -- fixed-step delta codec (transform id MM = 15) and the per-block data-type tag, for a batch on gfx950.
//
// Replaces K/transform/FSDCodec.java:60-244 (forward), :246-318 (inverse), :320-323 (getMaxEncodedLength) and the
// helpers it leans on: K/Global.java:222-235 (log2_1024), :440-456 (computeFirstOrderEntropy1024), :556-605
// (detectSimpleType), K/Magic.java (getType and its three predicates), CompressedOutputStream.java:795-804 (the
// writer's BIN / MULTIMEDIA / EXE tag per block).
//
// forward, per block:  k_mm_analyze (one workgroup: 7 sampled 256-bin histograms in LDS -> 7 first-order entropies
//   -> step, coding mode, or "declined" plus the data type the reference stores in its context)
//   -> k_mm_emit (one wave: XOR coding is 1:1; delta coding writes 1 or 2 bytes per input byte, the offsets of a row
//   of 64 come from one ballot) -> k_mm_check (sampled histogram of the coded form against the plain entropy).
// inverse, per block (one wave): XOR coding is a prefix XOR along each of the `dist` interleaved chains = a strided
//   wave scan per row of 64; delta coding mixes additions with the XOR of an escape, which does not compose into one
//   scan operator: token starts by ballot, tokens compacted through LDS, a segmented strided scan of the deltas that
//   restarts at escapes, the (rare) escapes resolved in order, bases added.
// All arithmetic is the reference's integer arithmetic (64-bit sums, truncating divide); nothing here is tunable.
#include "kz_device.h"
#include "kz_internal.h"
#include "kz_datatype.h"
#include "kz_magic.h"
#include <cmath>
#include <mutex>

typedef uint32_t u32;
typedef uint8_t u8;

#define MM_MIN_LENGTH 1024
#define MM_ESCAPE 0xFFu
#define MM_DELTA 0
#define MM_XOR 1

// the writer's tag (CompressedOutputStream.java:795-804); `init` is what the context held before (UNDEFINED for the
// batched calls, kz_ctx_set_data_type for the single-block ones)
__global__ void k_block_magic(const u8* __restrict__ src, int64_t stride, const int32_t* __restrict__ d_len, int32_t* __restrict__ d_dtype,
                              int init, int sniff, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int dt = init;
  if (sniff && d_len[b] >= 4) {
    const int32_t m = mm_magic_type(src + (int64_t)b * stride);
    if (mm_is_compressed(m)) dt = DT_BIN;
    else if (mm_is_multimedia(m)) dt = DT_MULTIMEDIA;
    else if (mm_is_executable(m)) dt = DT_EXE;
  }
  d_dtype[b] = dt;
}

// ---- Global.log2_1024 / computeFirstOrderEntropy1024 ----
__device__ __forceinline__ int mm_log2_1024(const int32_t* __restrict__ tab, int x) {     // x > 0
  if (x < 256) return (tab[x] + 2) >> 2;
  const int lg = 31 - __clz(x);
  if ((x & (x - 1)) == 0) return lg << 10;
  return ((lg - 7) * 1024) + ((tab[x >> (lg - 7)] + 2) >> 2);
}
#define mm_wg_sum64 kz_wg256_sum64
__device__ __forceinline__ int mm_entropy1024(const int32_t* __restrict__ tab, int length, int h, long long* lds4) {
  long long term = 0;
  if (h != 0) term = ((long long)h * (long long)(mm_log2_1024(tab, length) - mm_log2_1024(tab, h))) >> 3;
  const long long sum = mm_wg_sum64(term, lds4);
  return (length == 0) ? 0 : (int)(sum / length);
}

// The writer's "skipBlocks" option (CompressedOutputStream.java:769-788): a block whose first bytes carry the magic
// number of a compressed format, or whose order-0 entropy is at least 0.95 * 8 bits (EntropyUtils.INCOMPRESSIBLE_THRESHOLD
// = 973 on the x1024 scale), is stored as a copy block.  One workgroup per block.
#define MM_INCOMPRESSIBLE_THRESHOLD 973
__global__ __launch_bounds__(256) void k_skip_decide(const u8* __restrict__ srcAll, int64_t stride, const int32_t* __restrict__ d_len,
                                                      const int32_t* __restrict__ log2tab, int32_t* __restrict__ d_skip) {
  const int b = blockIdx.x;
  const int count = d_len[b];
  const int tid = threadIdx.x;
  __shared__ int h[256];
  __shared__ long long lds4[4];
  if (count <= 15) { if (tid == 0) d_skip[b] = 0; return; }                       // small blocks are copy blocks anyway (:764-767)
  const u8* src = srcAll + (int64_t)b * stride;
  if (mm_is_compressed(mm_magic_type(src))) { if (tid == 0) d_skip[b] = 1; return; }
  h[tid] = 0;
  __syncthreads();
  for (int row = 0; row < count; row += 256) {
    const int i = row + tid;
    const bool valid = i < count;
    const u32 v = valid ? (u32)src[i] : 0u;
    const uint64_t peers = kz_match8(v, valid);
    if (valid && (peers & kz_lanemask_lt()) == 0) atomicAdd(&h[v], (int)__popcll(peers));
  }
  __syncthreads();
  const int e = mm_entropy1024(log2tab, count, h[tid], lds4);
  if (tid == 0) d_skip[b] = (e >= MM_INCOMPRESSIBLE_THRESHOLD) ? 1 : 0;
}

struct MmFwd {
  int32_t* go;       // [B] 1 = analysis chose a coding, emit it
  int32_t* mode;     // [B]
  int32_t* dist;     // [B]
  int32_t* ent0;     // [B] entropy of the plain samples (x1024)
  int32_t* produced; // [B] bytes written by k_mm_emit, -1 = did not fit
  const int32_t* log2tab;
};

__global__ __launch_bounds__(256) void k_mm_analyze(const u8* __restrict__ srcAll, int64_t stride, const int32_t* __restrict__ d_len,
                                                     int32_t* __restrict__ d_dtype, MmFwd M) {
  const int b = blockIdx.x;
  const int count = d_len[b];
  const int tid = threadIdx.x;
  __shared__ int histo[7][256];
  __shared__ long long lds4[4];
  __shared__ int sh_ent[7];
  __shared__ int sh_pick;
  if (tid == 0) M.go[b] = 0;
  if (count < MM_MIN_LENGTH) return;                                             // :75-76 (count == 0: caller)
  const int dt = d_dtype[b];
  if (dt != DT_UNDEFINED && dt != DT_MULTIMEDIA && dt != DT_BIN) return;        // :78-85
  const u8* src = srcAll + (int64_t)b * stride;
  {
    const int32_t magic = mm_magic_type(src);                                    // :87-100
    const u32 k = (u32)magic;
    if (!(k == 0x424Du || k == 0x52494646u || k == 0x5034u || k == 0x5035u || k == 0x5036u || magic == 0)) return;
  }
  for (int i = tid; i < 7 * 256; i += 256) (&histo[0][0])[i] = 0;
  __syncthreads();
  const int count10 = count / 10, count5 = 2 * count10;
  for (int i = count10 + tid; i < count5; i += 256) {                            // :118-148
#pragma unroll
    for (int w = 0; w < 3; w++) {
      const u8* p = src + 2 * w * count5 + i;
      const u32 v = p[0];
      atomicAdd(&histo[0][v], 1);
      atomicAdd(&histo[1][v ^ p[-1]], 1);
      atomicAdd(&histo[2][v ^ p[-2]], 1);
      atomicAdd(&histo[3][v ^ p[-3]], 1);
      atomicAdd(&histo[4][v ^ p[-4]], 1);
      atomicAdd(&histo[5][v ^ p[-8]], 1);
      atomicAdd(&histo[6][v ^ p[-16]], 1);
    }
  }
  __syncthreads();
  for (int k = 0; k < 7; k++) {
    const int e = mm_entropy1024(M.log2tab, 3 * count10, histo[k][tid], lds4);
    if (tid == 0) sh_ent[k] = e;
  }
  __syncthreads();
  if (tid == 0) {
    int minIdx = 0;
    for (int k = 0; k < 7; k++) if (sh_ent[k] < sh_ent[minIdx]) minIdx = k;
    sh_pick = (sh_ent[minIdx] >= sh_ent[0]) ? -1 : minIdx;
  }
  __syncthreads();
  if (sh_pick < 0) {                                                             // :160-165 -> Global.detectSimpleType :556-605
    const int t = kz_detect_simple_type_wg(3 * count10, histo[0][tid], histo[0][0x3D], lds4);
    if (tid == 0) d_dtype[b] = t;
    return;
  }
  const int DIST[7] = {0, 1, 2, 3, 4, 8, 16};
  const int dist = DIST[sh_pick];
  long long large = 0;
  for (int i = 2 * count5 + tid; i < 3 * count5; i += 256) {                     // :173-179
    const int delta = (int)src[i] - (int)src[i - dist];
    if (delta < -127 || delta > 127) large++;
  }
  const long long largeDeltas = mm_wg_sum64(large, lds4);
  if (tid == 0) {
    d_dtype[b] = DT_MULTIMEDIA;
    M.mode[b] = (largeDeltas > (count5 >> 5)) ? MM_XOR : MM_DELTA;
    M.dist[b] = dist;
    M.ent0[b] = sh_ent[0];
    M.go[b] = 1;
  }
}

__device__ __forceinline__ int mm_max_encoded_len(int n) { return n + max(64, n >> 4); }

// one wave per block: header, the first `dist` bytes as they are, then one coded token per input byte (:186-217)
__global__ __launch_bounds__(64) void k_mm_emit(const u8* __restrict__ srcAll, u8* __restrict__ dstAll, int64_t stride,
                                                 const int32_t* __restrict__ d_len, MmFwd M) {
  const int b = blockIdx.x;
  if (!M.go[b]) return;
  const int count = __builtin_amdgcn_readfirstlane(d_len[b]);
  const int lane = kz_lane();
  const u8* src = srcAll + (int64_t)b * stride;
  u8* dst = dstAll + (int64_t)b * stride;
  const int mode = __builtin_amdgcn_readfirstlane(M.mode[b]), dist = __builtin_amdgcn_readfirstlane(M.dist[b]);
  const int dstEnd = mm_max_encoded_len(count);
  if (lane == 0) { dst[0] = (u8)mode; dst[1] = (u8)dist; }
  if (lane < dist) dst[2 + lane] = src[lane];
  int dstIdx = 2 + dist;
  bool fits = true;
  if (mode == MM_XOR) {
    for (int row = dist; row < count; row += 64) {
      const int i = row + lane;
      if (i < count) dst[2 + i] = (u8)(src[i] ^ src[i - dist]);
    }
    dstIdx = 2 + count;
  } else {
    for (int row = dist; row < count; row += 64) {
      const int i = row + lane;
      const bool valid = i < count;
      int delta = 0;
      u32 x = 0;
      if (valid) { const u32 a = src[i], p = src[i - dist]; delta = (int)a - (int)p; x = a ^ p; }
      const bool esc = valid && (delta < -127 || delta > 127);
      const uint64_t em = kz_ballot(esc);
      const int at = dstIdx + lane + (int)__popcll(em & kz_lanemask_lt());
      // the reference's loop stops before a token that would start at or past dstEnd - 1 and then reports failure
      if (valid && at >= dstEnd - 1) fits = false;
      if (valid && at < dstEnd - 1) {
        if (esc) { dst[at] = (u8)MM_ESCAPE; dst[at + 1] = (u8)x; }
        else dst[at] = (u8)((delta >> 31) ^ (delta << 1));                       // zigzag
      }
      dstIdx += min(64, count - row) + (int)__popcll(em);
    }
  }
  const bool ok = kz_ballot(!fits) == 0;
  if (lane == 0) M.produced[b] = ok ? dstIdx : -1;
}

// does the coded form look better?  (:222-233)
__global__ __launch_bounds__(256) void k_mm_check(const u8* __restrict__ dstAll, int64_t stride, const int32_t* __restrict__ d_len,
                                                   int32_t* __restrict__ d_len2, int32_t* __restrict__ d_flag, MmFwd M) {
  const int b = blockIdx.x;
  const int count = d_len[b];
  const int tid = threadIdx.x;
  __shared__ int h[256];
  __shared__ long long lds4[4];
  if (count == 0) { if (tid == 0) { d_len2[b] = 0; d_flag[b] = 1; } return; }    // :62-63
  if (!M.go[b] || M.produced[b] < 0) { if (tid == 0) { d_len2[b] = count; d_flag[b] = 0; } return; }
  const u8* dst = dstAll + (int64_t)b * stride;
  const int count10 = count / 10, count5 = 2 * count10;
  h[tid] = 0;
  __syncthreads();
  for (int i = tid; i < count10; i += 256) { atomicAdd(&h[dst[count5 + i]], 1); atomicAdd(&h[dst[3 * count5 + i]], 1); }
  __syncthreads();
  const int e = mm_entropy1024(M.log2tab, count5, h[tid], lds4);
  if (tid == 0) {
    const bool applied = e < M.ent0[b];
    d_len2[b] = applied ? M.produced[b] : count;
    d_flag[b] = applied ? 1 : 0;
  }
}

// ---- inverse: one wave per block ----
__global__ __launch_bounds__(64) void k_mm_inv(const u8* __restrict__ srcAll, u8* __restrict__ dstAll, int64_t stride,
                                                const int32_t* __restrict__ d_len, int32_t* __restrict__ d_len2, int32_t* __restrict__ d_flag, int dstCap) {
  const int b = blockIdx.x;
  const int count = __builtin_amdgcn_readfirstlane(d_len[b]);
  const int lane = kz_lane();
  const u8* src = srcAll + (int64_t)b * stride;
  u8* dst = dstAll + (int64_t)b * stride;
  if (count == 0) { if (lane == 0) { d_len2[b] = 0; d_flag[b] = 1; } return; }
  bool ok = count >= 2;
  int produced = 0;
  if (ok) {
    const int mode = __builtin_amdgcn_readfirstlane((int)src[0]), dist = __builtin_amdgcn_readfirstlane((int)src[1]);
    if (dist < 1 || (dist > 4 && dist != 8 && dist != 16)) ok = false;           // :268-270
    else if (2 + dist > count || dist > dstCap) ok = false;                      // first bytes past the block / the array
    else if (mode == MM_XOR) {
      const int n = count - 2;                                                   // output bytes, 1:1 with the payload
      if (n > dstCap) ok = false;
      else {
        // out[i] = in[i] ^ out[i - dist]: inclusive XOR scan with stride dist inside the row, then the chain's last
        // value of the previous row (lane 64 - dist + (lane mod dist))
        u32 prev = 0;
        for (int row = 0; row < n; row += 64) {
          const int i = row + lane;
          u32 x = (i < n) ? (u32)src[2 + i] : 0u;
          for (int s = dist; s < 64; s <<= 1) { const u32 up = (u32)__shfl_up((int)x, s, 64); if (lane >= s) x ^= up; }
          if (row > 0) x ^= (u32)__shfl((int)prev, 64 - dist + (lane % dist), 64);
          if (i < n) dst[i] = (u8)x;
          prev = x;
        }
        produced = n;
      }
    } else if (mode == MM_DELTA) {
      // Tokens are 1 byte (zigzag delta: out = out[-dist] + d) or 2 bytes (escape: out = out[-dist] ^ v).  Per row of
      // 64 source bytes: find the token starts (a 0xFF is a payload when an odd number of 0xFF precede it), compact
      // the tokens through LDS, sum the deltas along each of the `dist` chains with a segmented strided scan that
      // restarts at escapes, resolve the (rare) escapes in order, then add the bases.  Same values as the reference's
      // token-by-token loop (:273-296) on any input.
      __shared__ u32 tok[64];
      const int jm = lane % dist;
      const int dstEnd = dstCap;
      u32 prevVals = (lane < dist) ? (u32)src[2 + lane] : 0u;                    // the first `dist` bytes, as they are
      int prevM = dist;
      int outBase = dist;
      if (lane < dist) dst[lane] = (u8)prevVals;
      bool carryPayload = false;
      bool loneTail = false;
      for (int rs = 2 + dist; rs < count && ok; rs += 64) {
        const int i = rs + lane;
        const bool valid = i < count;
        const u32 x = valid ? (u32)src[i] : 0u;
        const u32 xn = (i + 1 < count) ? (u32)src[i + 1] : 0u;                   // payload of an escape starting here
        const uint64_t F = kz_ballot(valid && x == MM_ESCAPE);
        // run of 0xFF bytes right below this lane
        const uint64_t below = F & kz_lanemask_lt();
        const int run = (lane == 0) ? 0 : (int)__builtin_clzll(~(below << (64 - lane)) | 0ULL) ;
        const int runc = (lane == 0) ? 0 : min(run, lane);
        bool payload = (runc & 1) != 0;
        if (runc == lane) payload = ((lane & 1) != 0) != carryPayload;           // the run reaches the row start
        const bool isTok = valid && !payload;
        const bool esc = isTok && x == MM_ESCAPE;
        const bool lone = esc && (i + 1 >= count);                                // escape with nothing behind it (:278-279)
        if (kz_ballot(lone)) loneTail = true;
        const uint64_t T = kz_ballot(isTok && !lone);
        const int m = (int)__popcll(T);
        // next row's lane 0 is a payload iff this row's last byte is an escape token start
        carryPayload = ((T | kz_ballot(lone)) >> 63) & ((F >> 63) & 1ULL);
        if (outBase + m > dstEnd) { ok = false; break; }                          // output full before the input ends
        const int rank = (int)__popcll(T & kz_lanemask_lt());
        if (isTok && !lone) tok[rank] = esc ? (0x100u | xn) : x;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);                                       // lgkmcnt(0): LDS write visible to the wave
        const u32 t = (lane < m) ? tok[lane] : 0u;
        __builtin_amdgcn_wave_barrier();
        const bool isEsc = (t & 0x100u) != 0 && lane < m;
        int sum = (lane < m && !isEsc) ? ((int)((t & 0xFFu) >> 1) ^ -(int)(t & 1u)) : 0;      // zigzag decode
        int flag = isEsc ? 1 : 0;
        int seg = isEsc ? lane : -1;
        for (int st = dist; st < 64; st <<= 1) {
          const int us = __shfl_up(sum, st, 64), uf = __shfl_up(flag, st, 64), ug = __shfl_up(seg, st, 64);
          if (lane >= st && !flag) { sum += us; flag = uf; seg = ug; }
        }
        // base of a lane whose segment started before this row: its chain's last value in the previous row
        const u32 carryBase = (u32)__shfl((int)prevVals, (prevM - dist + jm) & 63, 64);
        u32 escVal = 0;
        uint64_t E = kz_ballot(isEsc);
        while (E) {                                                               // escapes, in order
          const int e = (int)__builtin_ctzll(E);
          E &= E - 1;
          const int p = e - dist;
          u32 pv;
          if (p < 0) pv = (u32)__builtin_amdgcn_readlane((int)prevVals, (prevM + p) & 63);
          else {
            const int sp = __builtin_amdgcn_readlane(seg, p);
            const u32 bp = (sp >= 0) ? (u32)__builtin_amdgcn_readlane((int)escVal, sp & 63) : (u32)__builtin_amdgcn_readlane((int)carryBase, p);
            pv = (bp + (u32)__builtin_amdgcn_readlane(sum, p)) & 0xFFu;
          }
          const u32 v = ((u32)__builtin_amdgcn_readlane((int)t, e) ^ pv) & 0xFFu;
          escVal = (lane == e) ? v : escVal;
        }
        const u32 segVal = (u32)__shfl((int)escVal, seg & 63, 64);               // every lane active: masked-off lanes supply nothing
        const u32 base = (seg >= 0) ? segVal : carryBase;
        const u32 val = (base + (u32)sum) & 0xFFu;
        if (lane < m) dst[outBase + lane] = (u8)val;
        if (m > 0) { prevVals = val; prevM = m; }
        outBase += m;
      }
      if (ok && loneTail && outBase >= dstEnd) ok = false;                        // the loop would have stopped before the lone escape
      produced = outBase;
    } else ok = false;                                                            // :302-305
  }
  if (lane == 0) { d_len2[b] = ok ? produced : 0; d_flag[b] = ok ? 1 : 0; }
}

// Global.LOG2_4096 (K/Global.java:104-127): round(4096 * log2(x)), x = 1..256; generated once per process
static const int32_t* mm_log2_table_host() {
  static int32_t tab[257];
  static std::once_flag once;
  std::call_once(once, [] {
    tab[0] = 0;
    for (int x = 1; x <= 256; x++) tab[x] = (int32_t)std::floor(4096.0 * std::log2((double)x) + 0.5);
  });
  return tab;
}

size_t kz_mm_scratch(int B, int) { return (size_t)B * 4 * 6 + 257 * 4 + 4096; }

int kz_block_data_types(kz_ctx* ctx, kz_batch& bt, int init, bool sniff) {
  KZ_LAUNCH(ctx, KID_BLOCK_MAGIC, k_block_magic, dim3((bt.B + 255) / 256), dim3(256), bt.buf[bt.cur], bt.stride, bt.d_len, bt.d_dtype, init, sniff ? 1 : 0, bt.B);
  KZ_HIP(hipGetLastError());
  return 0;
}

static int mm_upload_table(kz_ctx* ctx, int32_t** d_tab) {
  *d_tab = (int32_t*)kz_arena_alloc(ctx, 257 * 4);
  if (!*d_tab) { snprintf(ctx->err, sizeof(ctx->err), "mm: arena overflow"); return -KZ_ERR_DEVICE; }
  KZ_HIP(hipMemcpyAsync(*d_tab, mm_log2_table_host(), 257 * 4, hipMemcpyHostToDevice, ctx->stream));
  return 0;
}

// d_skip[b] = 1 for the blocks the writer would store as copy blocks under its "skipBlocks" option
int kz_skip_block_flags(kz_ctx* ctx, kz_batch& bt, int32_t* d_skip) {
  int32_t* tab = nullptr;
  { const int rc = mm_upload_table(ctx, &tab); if (rc) return rc; }
  KZ_LAUNCH(ctx, KID_SKIP_DECIDE, k_skip_decide, dim3(bt.B), dim3(256), bt.buf[bt.cur], bt.stride, bt.d_len, tab, d_skip);
  KZ_HIP(hipGetLastError());
  return 0;
}

int kz_stage_mm_forward(kz_ctx* ctx, kz_batch& bt) {
  const int B = bt.B;
  MmFwd M;
  int32_t* tab = nullptr;
  { const int rc = mm_upload_table(ctx, &tab); if (rc) return rc; }
  M.log2tab = tab;
  M.go = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  M.mode = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  M.dist = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  M.ent0 = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  M.produced = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  if (!M.produced || !M.go) { snprintf(ctx->err, sizeof(ctx->err), "mm_forward: arena overflow"); return -KZ_ERR_DEVICE; }
  const u8* src = bt.buf[bt.cur];
  u8* dst = bt.buf[bt.cur ^ 1];
  KZ_LAUNCH(ctx, KID_MM_ANALYZE, k_mm_analyze, dim3(B), dim3(256), src, bt.stride, bt.d_len, bt.d_dtype, M);
  KZ_LAUNCH(ctx, KID_MM_EMIT, k_mm_emit, dim3(B), dim3(64), src, dst, bt.stride, bt.d_len, M);
  KZ_LAUNCH(ctx, KID_MM_CHECK, k_mm_check, dim3(B), dim3(256), dst, bt.stride, bt.d_len, bt.d_len2, bt.d_flag, M);
  KZ_HIP(hipGetLastError());
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}

int kz_stage_mm_inverse(kz_ctx* ctx, kz_batch& bt, int dstCap) {
  const int B = bt.B;
  if ((int64_t)dstCap > bt.stride) dstCap = (int)bt.stride;
  KZ_LAUNCH(ctx, KID_MM_INV, k_mm_inv, dim3(B), dim3(64), bt.buf[bt.cur], bt.buf[bt.cur ^ 1], bt.stride, bt.d_len, bt.d_len2, bt.d_flag, dstCap);
  KZ_HIP(hipGetLastError());
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}
