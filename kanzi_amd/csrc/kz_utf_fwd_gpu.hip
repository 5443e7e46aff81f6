// kz_utf_fwd_gpu.hip -- UTFCodec.forward (K/transform/UTFCodec.java:68-221) on the device, for the blocks the device TEXT forward
// declined with the "dataType" UTF8 (TextCodec's detectType found valid UTF-8: UTFCodec then skips its own validation, :100).
//
// The reference walks the block's code points one by one (i += SIZES[lead >> 4], :136-165): the units of a code point are
// consumed whatever they hold, so "is byte i the first unit of a code point" is the state of a four-state machine (units still to
// skip) run from `start` -- a composition of per-byte maps {0..3} -> {0..3}, i.e. a scan: k_uf_map composes the maps of 4 KiB tiles,
// k_uf_tiles runs the tile maps along the block, and every later pass knows each tile's entry state.  Then
//   k_uf_count  occurrences per code point: a direct table of 3 x 65536 counters per block (1-, 2- and 3-unit code points have 16
//               payload bits under their size tag; the ASCII and two-unit counters are gathered in LDS per tile first), the
//               reference's checks of the third unit (:140-145);
//   k_uf_syms   one workgroup per block: the distinct code points, sorted by (count, key) descending in LDS (the reference sorts
//               ascending and reads backwards, :176-195), the map written behind the 4-byte header, the alias of every code point
//               stored back into the table (one byte for the 128 most frequent, two above), the reference's three early exits
//               (no symbol / 3 n + 6 >= 0.9 count / estimate >= 0.9 count);
//   k_uf_emit<false>, k_uf_scan, k_uf_emit<true>   sizes per tile, offsets, the alias bytes; the final "shorter than 0.9 count" test.
// What this form does not do goes to the host stage with the block untouched: four-unit code points (their 21 payload bits would
// need the reference's 2^22-entry table), more than 16384 distinct code points (the sort's LDS), blocks whose "dataType" is
// UNDEFINED (UTFCodec then validates with its own pair statistics, :317-430; such blocks almost always fail it).
#include "kz_device.h"
#include "kz_internal.h"

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

#define UF_TILE 4096
#define UF_TABLE (3 * 65536)
#define UF_MAXSYM 16384
#define UF_MIN_BLOCK 1024

struct UtfFwd {
  const int32_t* ord;       // [B] block -> index among the taken blocks, or -1
  u32* tileF;               // [A][maxTiles] composed map of each tile
  u8* tileIn;               // [A][maxTiles] entry state of each tile
  int32_t* tileSum;         // [A][maxTiles] alias bytes of each tile, then their exclusive sums
  u32* table;               // [A][UF_TABLE] counters, then aliases
  int32_t* info;            // [A][8]: 0 start, 1 adjust, 2 error / unsupported, 3 symbols, 4 verdict (1 go on, 2 declined, 0 host), 5 output length
  int maxTiles;
};

// a map {0..3} -> {0..3} with an error bit per entry state: bits 2e+1..2e = state left in when entered in e, bit 8+e = error
__device__ __forceinline__ u32 uf_compose(u32 f, u32 g) {          // f, then g
  u32 r = 0;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const u32 m = (f >> (2 * e)) & 3u;
    r |= ((g >> (2 * m)) & 3u) << (2 * e);
    r |= ((((f >> (8 + e)) | (g >> (8 + m))) & 1u)) << (8 + e);
  }
  return r;
}
#define UF_IDENT 0xE4u                                              // 3 2 1 0
__device__ __forceinline__ int uf_units(u32 b) { return (int)((0x4322000011111111ULL >> (4 * (b >> 4))) & 15ULL); }   // SIZES (:32)
// the map of one byte at position pos: outside [start, body) no code point starts (units still owed are consumed)
__device__ __forceinline__ u32 uf_byte_map(u32 b, int pos, int start, int body) {
  if (pos < start) return UF_IDENT;
  u32 f = 0x90u;                                                    // entered in 1, 2, 3: one unit less (2 1 0 ?)
  if (pos >= body) return f;                                        // entered in 0: stays 0
  const int u = uf_units(b);
  if (u == 0) return f | 0x100u;                                    // not a first unit: the reference stops (pack returns 0, :161)
  return f | (u32)(u - 1);
}
// the maps of this thread's 16 bytes composed; bytes beyond the block's end are the identity
__device__ __forceinline__ u32 uf_thread_map(const u8* p, int pos0, int n, int start, int body) {
  u32 f = UF_IDENT;
  for (int k = 0; k < 16; k++) { const int pos = pos0 + k; if (pos < n) f = uf_compose(f, uf_byte_map(p[k], pos, start, body)); }
  return f;
}
// exclusive scan of maps over the 256 threads of a tile; *total = the tile's map
__device__ __forceinline__ u32 uf_wg_scan(u32 f, u32* lds, u32* total) {
  const int lane = kz_lane(), wave = threadIdx.x >> 6;
  u32 inc = f;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const u32 t = (u32)__shfl_up((int)inc, d, 64); if (lane >= d) inc = uf_compose(t, inc); }
  __syncthreads();
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  u32 pre = UF_IDENT, all = UF_IDENT;
  for (int w = 0; w < 4; w++) { if (w < wave) pre = uf_compose(pre, lds[w]); all = uf_compose(all, lds[w]); }
  *total = all;
  u32 ex = (u32)__shfl_up((int)inc, 1, 64);
  if (lane == 0) ex = UF_IDENT;
  return uf_compose(pre, ex);
}

__global__ __launch_bounds__(64) void k_uf_init(const u8* __restrict__ src, int64_t stride, const int32_t* __restrict__ d_len, UtfFwd G, int B) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const int a = G.ord[b];
  if (a < 0) return;
  const u8* s = src + (int64_t)b * stride;
  const int n = d_len[b];
  int32_t* info = G.info + (int64_t)a * 8;
  int start = 0;
  if (n >= UF_MIN_BLOCK) {
    if (s[0] == 0xEF && s[1] == 0xBB && s[2] == 0xBF) start = 3;                                        // byte order mark (:106-110)
    else while (start < 4) { const u32 c = s[start]; const int ok = c < 0x80 ? 1 : (c < 0xC2 ? 0 : (c < 0xF5 ? 1 : 0)); if (ok) break; start++; }   // LEN_SEQ == 0 (:111-114)
  }
  info[0] = start; info[1] = 0; info[2] = n < UF_MIN_BLOCK ? 2 : 0; info[3] = 0; info[4] = 0; info[5] = 0;
}

__global__ __launch_bounds__(256) void k_uf_map(const u8* __restrict__ src, int64_t stride, const int32_t* __restrict__ d_len, UtfFwd G, int B) {
  const int b = blockIdx.y, a = G.ord[b];
  if (a < 0) return;
  const int n = d_len[b], t = blockIdx.x;
  if (t * UF_TILE >= n) return;
  const int32_t* info = G.info + (int64_t)a * 8;
  if (info[2]) return;
  const int start = info[0], body = n - 4;
  __shared__ u32 lds[8];
  const int pos0 = t * UF_TILE + (int)threadIdx.x * 16;
  u32 f = UF_IDENT;
  if (pos0 < n) f = uf_thread_map(src + (int64_t)b * stride + pos0, pos0, n, start, body);
  u32 total;
  uf_wg_scan(f, lds, &total);
  if (threadIdx.x == 0) G.tileF[(int64_t)a * G.maxTiles + t] = total;
}

__global__ __launch_bounds__(64) void k_uf_tiles(const int32_t* __restrict__ d_len, UtfFwd G, int B) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const int a = G.ord[b];
  if (a < 0) return;
  int32_t* info = G.info + (int64_t)a * 8;
  if (info[2]) return;
  const int n = d_len[b], nt = (n + UF_TILE - 1) / UF_TILE;
  u32 s = 0, err = 0;
  for (int t = 0; t < nt; t++) {
    G.tileIn[(int64_t)a * G.maxTiles + t] = (u8)s;
    const u32 f = G.tileF[(int64_t)a * G.maxTiles + t];
    err |= (f >> (8 + s)) & 1u;
    s = (f >> (2 * s)) & 3u;
  }
  if (err) info[2] = 1;                                             // a byte that cannot start a code point where one must start: UTF declines
}

// the table slot of the code point at p (u units): size tag << 16 | its 16 payload bits (= the low 16 bits of the reference's key, :436-455)
__device__ __forceinline__ u32 uf_slot(const u8* p, int u) {
  if (u == 1) return p[0];
  if (u == 2) return (1u << 16) | ((u32)p[0] << 8) | p[1];
  return (2u << 16) | ((u32)(p[0] & 0x0F) << 12) | ((u32)(p[1] & 0x3F) << 6) | (u32)(p[2] & 0x3F);
}

// PASS 0: count, 1: sizes, 2: write
template <int PASS>
__global__ __launch_bounds__(256) void k_uf_pass(const u8* __restrict__ src, u8* __restrict__ dst, int64_t stride, const int32_t* __restrict__ d_len, UtfFwd G, int B) {
  const int b = blockIdx.y, a = G.ord[b];
  if (a < 0) return;
  const int n = d_len[b], t = blockIdx.x;
  if (t * UF_TILE >= n) return;
  int32_t* info = G.info + (int64_t)a * 8;
  if (info[2] || (PASS > 0 && info[4] != 1)) return;
  const int start = info[0], body = n - 4;
  __shared__ u32 lds[8];
  __shared__ u32 bins[256 + 2048];                                  // PASS 0: ASCII counters, two-unit counters (32 first units x 64 second units)
  __shared__ u8 tile[UF_TILE + 16];
  const u8* s = src + (int64_t)b * stride;
  const int base = t * UF_TILE;
  for (int i = threadIdx.x * 16; i < UF_TILE + 16; i += 256 * 16) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (base + i < n) v = *(const uint4*)(s + base + i);            // (slots are 256-byte aligned and padded: the read stays inside the slot)
    *(uint4*)(tile + i) = v;
  }
  if (PASS == 0) for (int i = threadIdx.x; i < 256 + 2048; i += 256) bins[i] = 0;
  __syncthreads();
  const int off0 = (int)threadIdx.x * 16, pos0 = base + off0;
  u32 f = UF_IDENT;
  if (pos0 < n) f = uf_thread_map(tile + off0, pos0, n, start, body);
  u32 total;
  const u32 ex = uf_wg_scan(f, lds, &total);
  const u32 tin = G.tileIn[(int64_t)a * G.maxTiles + t];
  u32 st = (ex >> (2 * tin)) & 3u;                                  // units still owed when this thread's first byte comes
  u32* table = G.table + (int64_t)a * UF_TABLE;
  u32 mySize = 0;
  u32 al[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool bad = false, four = false;
  for (int k = 0; k < 16; k++) {
    const int pos = pos0 + k;
    if (pos >= n) break;
    if (pos == body && PASS == 0) info[1] = (int32_t)st;            // how far the last code point reaches into the four tail bytes (:207)
    if (pos < start) continue;
    if (st > 0) { st--; continue; }
    if (pos >= body) continue;
    const u8* p = tile + off0 + k;
    const int u = uf_units(p[0]);
    if (u == 0) { bad = true; break; }
    st = (u32)(u - 1);
    if (u == 4) { four = true; continue; }
    if (u == 3 && (p[2] < 0x80 || p[2] > 0xBF)) bad = true;         // :140-141
    const u32 slot = uf_slot(p, u);
    if (PASS == 0) {
      if (u == 1) atomicAdd(&bins[slot], 1u);
      else if (u == 2 && (p[1] & 0xC0u) == 0x80u) atomicAdd(&bins[256 + (((u32)p[0] - 0xC0u) << 6) + (p[1] & 63u)], 1u);   // (first unit 0xC0..0xDF)
      else atomicAdd(&table[slot], 1u);                             // three units; two units whose second is no continuation byte (never validated: :100)
    } else {
      const u32 alias = table[slot];
      al[k] = alias | 0x80000000u;
      mySize += 1u + (alias >> 16);
    }
  }
  if (PASS == 0) {
    if (bad) info[2] = 1;
    if (four) atomicOr((int*)&info[2], 2);
    __syncthreads();
    for (int i = threadIdx.x; i < 256 + 2048; i += 256) {
      const u32 c = bins[i];
      if (c) atomicAdd(&table[i < 256 ? (u32)i : ((1u << 16) | ((0xC0u + (((u32)i - 256u) >> 6)) << 8) | (0x80u + (((u32)i - 256u) & 63u)))], c);
    }
    return;
  }
  __shared__ u32 ls[32];
  u32 tileTotal;
  const u32 exs = kz_wg_excl_sum(mySize, ls, &tileTotal);
  if (PASS == 1) { if (threadIdx.x == 0) G.tileSum[(int64_t)a * G.maxTiles + t] = (int32_t)tileTotal; return; }
  u8* o = dst + (int64_t)b * stride + 4 + 3 * info[3] + start + G.tileSum[(int64_t)a * G.maxTiles + t] + exs;
  for (int k = 0; k < 16; k++) {
    if (!(al[k] & 0x80000000u)) continue;
    const u32 alias = al[k];
    *o++ = (u8)alias;
    if ((alias >> 16) & 1u) *o++ = (u8)(alias >> 8);
  }
}

// one workgroup per block: the distinct code points, their order, the map, the aliases
__global__ __launch_bounds__(1024) void k_uf_syms(u8* __restrict__ dst, int64_t stride, const int32_t* __restrict__ d_len, UtfFwd G, int B) {
  const int b = blockIdx.x, a = G.ord[b];
  if (a < 0) return;
  int32_t* info = G.info + (int64_t)a * 8;
  if (info[2]) { if (threadIdx.x == 0) info[4] = (info[2] & 1) ? 2 : 0; return; }       // error: UTF declines; four-unit code points: host stage
  const int n = d_len[b];
  __shared__ u64 key[UF_MAXSYM];
  __shared__ u32 cnt;
  u32* table = G.table + (int64_t)a * UF_TABLE;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < UF_TABLE; i += 1024) {
    const u32 c = table[i];
    if (c) {
      const u32 at = atomicAdd(&cnt, 1u);
      const u32 k22 = (((u32)i >> 16) << 19) | ((u32)i & 0xFFFFu);
      if (at < UF_MAXSYM) key[at] = ((u64)c << 22) | k22;
    }
  }
  __syncthreads();
  const int ns = (int)cnt;
  const int maxTarget = n - n / 10;
  if (ns > UF_MAXSYM) { if (threadIdx.x == 0) info[4] = 0; return; }                     // the host stage (it also knows the 32768 limit)
  if (ns == 0 || 3 * ns + 6 >= maxTarget) { if (threadIdx.x == 0) info[4] = 2; return; }   // :169-170
  int P = 1;
  while (P < ns) P <<= 1;
  for (int i = ns + (int)threadIdx.x; i < P; i += 1024) key[i] = 0;
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1)                                  // bitonic sort, descending (keys are distinct: the order is total)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < P; i += 1024) {
        const int l = i ^ j;
        if (l > i) {
          const u64 x = key[i], y = key[l];
          const bool up = (i & k) == 0;
          if (up ? (x < y) : (x > y)) { key[i] = y; key[l] = x; }
        }
      }
      __syncthreads();
    }
  // the map behind the header (:183-190), the aliases (:191), the size estimate (:187-195)
  u8* o = dst + (int64_t)b * stride;
  u32 est = 0;
  for (int r = threadIdx.x; r < ns; r += 1024) {
    const u64 kv = key[r];
    const u32 k22 = (u32)kv & 0x3FFFFFu, fr = (u32)(kv >> 22);
    o[4 + 3 * r] = (u8)(k22 >> 16); o[5 + 3 * r] = (u8)(k22 >> 8); o[6 + 3 * r] = (u8)k22;
    est += r < 128 ? fr : 2u * fr;
    table[((k22 >> 19) << 16) | (k22 & 0xFFFFu)] = r < 128 ? (u32)r : (0x10080u | (((u32)r << 1) & 0xFF00u) | ((u32)r & 0x7Fu));
  }
  __shared__ u32 ls[32];
  u32 total;
  kz_wg_excl_sum(est, ls, &total);
  if (threadIdx.x == 0) {
    o[2] = (u8)(ns >> 8); o[3] = (u8)ns;
    info[3] = ns;
    // estimate starts at the header's 4 bytes + 6 (:185); an output that cannot end below maxTarget is declined before it is written
    // (map + aliases may pass the slot: kz_text.hip)
    info[4] = ((long long)total + 10 >= (long long)maxTarget || (long long)4 + 3 * ns + info[0] + total + 1 >= (long long)maxTarget) ? 2 : 1;
  }
}

__global__ __launch_bounds__(256) void k_uf_scan(const u8* __restrict__ src, u8* __restrict__ dst, int64_t stride, const int32_t* __restrict__ d_len, UtfFwd G, int B) {
  const int b = blockIdx.x, a = G.ord[b];
  if (a < 0) return;
  int32_t* info = G.info + (int64_t)a * 8;
  if (info[4] != 1) return;
  const int n = d_len[b];
  const int nt = (n + UF_TILE - 1) / UF_TILE;
  const int per = (nt + 255) / 256;
  int32_t* ts = G.tileSum + (int64_t)a * G.maxTiles;
  __shared__ u32 lds[32];
  u32 run = 0;
  for (int i = 0; i < per; i++) { const int t = threadIdx.x * per + i; if (t < nt) run += (u32)ts[t]; }
  u32 total;
  u32 ex = kz_wg_excl_sum(run, lds, &total);
  for (int i = 0; i < per; i++) { const int t = threadIdx.x * per + i; if (t < nt) { const u32 v = (u32)ts[t]; ts[t] = (int32_t)ex; ex += v; } }
  if (threadIdx.x == 0) {
    const int start = info[0], adjust = info[1], ns = info[3];
    const u8* s = src + (int64_t)b * stride;
    u8* o = dst + (int64_t)b * stride;
    o[0] = (u8)start; o[1] = (u8)adjust;                            // :206-207
    for (int i = 0; i < start; i++) o[4 + 3 * ns + i] = s[i];        // the bytes in front of the first code point (:199-200)
    int at = 4 + 3 * ns + start + (int)total;
    for (int i = n - 4 + adjust; i < n; i++) o[at++] = s[i];        // the tail (:210-211)
    const int maxTarget = n - n / 10;
    info[5] = at;
    if (at >= maxTarget) info[4] = 2;                               // :214
  }
}

__global__ __launch_bounds__(256) void k_uf_copy_back(const u8* __restrict__ src, u8* __restrict__ dst, int64_t stride, const int32_t* __restrict__ len, const int32_t* __restrict__ cond) {
  const int b = blockIdx.y;
  if (!cond[b]) return;
  const int n16 = (len[b] + 15) >> 4;
  const uint4* s = (const uint4*)(src + (int64_t)b * stride);
  uint4* d = (uint4*)(dst + (int64_t)b * stride);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) d[i] = s[i];
}

size_t kz_utf_fwd_gpu_scratch(int B, int maxLen) {
  const size_t tiles = (size_t)maxLen / UF_TILE + 2;
  return (size_t)B * (tiles * 9 + 64 + (size_t)UF_TABLE * 4 + 32 + 12) + (1 << 16);
}

// take[b] != 0: the block (in bt.buf[cur], "dataType" UTF8, TEXT declined) goes through UTFCodec.forward here.  done[b] = 1: UTF
// applied, the block's slot holds its output, bt.h_len / bt.d_len are updated (skip bit and "dataType" = UTF8 are the caller's);
// done[b] = 2: UTF declines by the reference's rules (block untouched, "dataType" UTF8); 0: not decided here (host stage).
int kz_utf_fwd_gpu(kz_ctx* ctx, kz_batch& bt, const std::vector<int32_t>& take, std::vector<int32_t>& done) {
  const int B = bt.B;
  done.assign(B, 0);
  std::vector<int32_t> ord(B, -1);
  int A = 0, maxLen = 0;
  for (int b = 0; b < B; b++) if (take[b]) { ord[b] = A++; maxLen = std::max(maxLen, bt.h_len[b]); }
  if (A == 0) return 0;
  kz_arena_guard guard{ctx, ctx->arenaTop};
  UtfFwd G;
  G.maxTiles = maxLen / UF_TILE + 2;
  int32_t* dOrd = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  int32_t* dOut = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  int32_t* dCond = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  G.tileF = (u32*)kz_arena_alloc(ctx, (size_t)A * G.maxTiles * 4);
  G.tileSum = (int32_t*)kz_arena_alloc(ctx, (size_t)A * G.maxTiles * 4);
  G.tileIn = (u8*)kz_arena_alloc(ctx, (size_t)A * G.maxTiles);
  G.table = (u32*)kz_arena_alloc(ctx, (size_t)A * UF_TABLE * 4);
  G.info = (int32_t*)kz_arena_alloc(ctx, (size_t)A * 8 * 4);
  if (!dOrd || !dOut || !dCond || !G.tileF || !G.tileSum || !G.tileIn || !G.table || !G.info) return 0;     // no room: the host stage takes them all
  G.ord = dOrd;
  hipStream_t st = ctx->stream;
  KZ_HIP(hipMemcpyAsync(dOrd, ord.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
  KZ_HIP(hipMemsetAsync(G.table, 0, (size_t)A * UF_TABLE * 4, st));
  KZ_HIP(hipMemsetAsync(G.tileSum, 0, (size_t)A * G.maxTiles * 4, st));
  const u8* src = bt.buf[bt.cur]; u8* dst = bt.buf[bt.cur ^ 1];
  const dim3 tiles(G.maxTiles, B);
  KZ_LAUNCH(ctx, KID_UTF_FWD, k_uf_init, dim3((B + 63) / 64), dim3(64), src, bt.stride, bt.d_len, G, B);
  KZ_LAUNCH(ctx, KID_UTF_FWD, k_uf_map, tiles, dim3(256), src, bt.stride, bt.d_len, G, B);
  KZ_LAUNCH(ctx, KID_UTF_FWD, k_uf_tiles, dim3((B + 63) / 64), dim3(64), bt.d_len, G, B);
  KZ_LAUNCH(ctx, KID_UTF_FWD, k_uf_pass<0>, tiles, dim3(256), src, dst, bt.stride, bt.d_len, G, B);
  KZ_LAUNCH(ctx, KID_UTF_FWD, k_uf_syms, dim3(B), dim3(1024), dst, bt.stride, bt.d_len, G, B);
  KZ_LAUNCH(ctx, KID_UTF_FWD, k_uf_pass<1>, tiles, dim3(256), src, dst, bt.stride, bt.d_len, G, B);
  KZ_LAUNCH(ctx, KID_UTF_FWD, k_uf_scan, dim3(B), dim3(256), src, dst, bt.stride, bt.d_len, G, B);
  KZ_LAUNCH(ctx, KID_UTF_FWD, k_uf_pass<2>, tiles, dim3(256), src, dst, bt.stride, bt.d_len, G, B);
  std::vector<int32_t> info((size_t)A * 8);
  KZ_HIP(hipMemcpyAsync(info.data(), G.info, (size_t)A * 8 * 4, hipMemcpyDeviceToHost, st));
  KZ_HIP(kz_stream_sync(ctx, st));
  std::vector<int32_t> cond(B, 0), newLen(bt.h_len);
  int nDone = 0, nDecl = 0;
  for (int b = 0; b < B; b++) {
    if (ord[b] < 0) continue;
    const int32_t* I = &info[(size_t)ord[b] * 8];
    if (I[4] == 1) { cond[b] = 1; newLen[b] = I[5]; done[b] = 1; nDone++; }
    else if (I[4] == 2) { done[b] = 2; nDecl++; }
  }
  if (nDone) {
    KZ_HIP(hipMemcpyAsync(dCond, cond.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    KZ_HIP(hipMemcpyAsync(dOut, newLen.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_uf_copy_back, dim3(64, B), dim3(256), 0, st, bt.buf[bt.cur ^ 1], bt.buf[bt.cur], bt.stride, dOut, dCond);
    for (int b = 0; b < B; b++) bt.h_len[b] = newLen[b];
    KZ_HIP(hipMemcpyAsync(bt.d_len, bt.h_len.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    KZ_HIP(kz_stream_sync(ctx, st));
  }
  KZ_HIP(hipGetLastError());
  if (ctx->sw.textGpuTrace) fprintf(stderr, "[utffwd] took %d blocks, finished %d, declined %d\n", A, nDone, nDecl);
  return 0;
}
