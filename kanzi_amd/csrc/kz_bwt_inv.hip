// kz_bwt_inv.hip -- inverse BWT for a batch of blocks on gfx950.
//
// Replaces K/transform/BWTBlockCodec.java:138-213 (header parse + checks) and
// K/transform/BWT.java:245-374 (inverseMergeTPSI).  Same data structure as the reference -- a packed
// link array data[j] = (next << 8) | byte built by a stable counting sort of the BWT bytes
// (BWT.java:264-293) -- but built in parallel: LDS tile histograms, a per-symbol scan over tiles,
// and a ballot match-any stable scatter.  The reference then runs 8 pointer-chasing walkers from the
// 8 primary indexes (BWT.java:295-368); a pointer chase is latency bound (one HBM round trip per
// byte), so here every block gets THOUSANDS of walkers: one per grid point j*S of the link array plus
// the text head.  Every walker follows its segment ONCE (until the next grid point), counting its steps and
// recording its bytes into pooled 256-byte chunks; a per-block LDS pointer-jumping pass turns the segment
// lengths into text offsets and a copy kernel moves the chunks to their place (a second walk would cost
// another cache line per byte).  That stitching needs what every well-formed block has: one path from the text head
// through all n links, with the 8 primary indexes sitting ckSize steps apart on it.  k_bwti_resolve checks exactly
// that; a block that fails the check (corrupted input) is marked BI_SUSPECT and redone by k_bwti_literal, which runs
// the reference's 8 walkers literally (BWT.java:295-368, including its dummy link 0xFF behind row 0), so the bytes
// and the verdict match the reference on ANY input -- slowly, but only for blocks no encoder produced.
// Blocks of 2^24-1 bytes and more (round 5; the reference switches to inverseBiPSIv2 above 8 MiB, BWT.java:384-544, with the same
// output) take the same kernels with 8-byte links ((next << 8) | byte with a 32-bit next): template parameter WIDE.  A pair
// table like biPSIv2's would not halve the fetches here: with thousands of walkers that stop at grid ROWS, a two-step link splits
// the text into an even and an odd path and the walkers started on the grid cover both (n hops in total again).
#include "kz_device.h"
#include "kz_internal.h"
#define BI_LD(p) __builtin_nontemporal_load(p)   // link loads: every line is used for one 4-byte link per visit

typedef uint32_t u32;
typedef uint64_t u64;
typedef uint8_t u8;

#define BI_ITEMS 16
#define BI_TILE (KZ_WG * BI_ITEMS)

struct BwtInv {
  const int32_t* ord; // [B] dense index of the block among the blocks of this call that have data (the [A] arrays below)
  void* data;       // [A][NS] links: u32 (next << 8 | byte, next < 2^24-1) or, for batches with a block of 2^24-1 bytes or more, u64
  int wide;
  u32* tileHist;    // [A][T][256]
  u32* bucket;      // [B][256]
  int32_t* n;       // [B] payload length
  int32_t* hdr;     // [B] header size
  int32_t* prim;    // [B][8] primary indexes (as stored + 1)
  int32_t* status;  // [B]
  u32* segLen;      // [A][GS]  walker segment lengths
  int32_t* segNext; // [A][GS]  next segment id on the text path, -1 = end of text
  u32* segOff;      // [A][GS]  text offset of each segment
  u8* pool;         // [A][maxChunks * BI_CH] bytes recorded by the walkers, in chunks of BI_CH bytes
  uint2* chunkMeta; // [A][maxChunks] (walker, sequence number inside its segment)
  u32* chunkCount;  // [B] chunks handed out
  int maxChunks;
  int64_t NS; int T;
  int logS;         // grid spacing = 1 << logS
  int GS;           // walker stride per block (grid points + 1)
};
#define BI_END 0xFFFFFFu
// link formats: make(next, byte), next(link), END = the next of the row behind the last text byte
template <bool WIDE> struct BiLink;
template <> struct BiLink<false> {
  typedef u32 T;
  static constexpr u32 END = BI_END;
  static __device__ __forceinline__ T make(u32 nx, u32 c) { return (nx << 8) | c; }
  static __device__ __forceinline__ u32 next(T v) { return v >> 8; }
};
template <> struct BiLink<true> {
  typedef u64 T;
  static constexpr u32 END = 0xFFFFFFFFu;
  static __device__ __forceinline__ T make(u32 nx, u32 c) { return ((u64)nx << 8) | (u64)c; }
  static __device__ __forceinline__ u32 next(T v) { return (u32)(v >> 8); }
};
#define BI_CH 128          // bytes per recording chunk (segments average 256 bytes since round 5)
#define BI_SUSPECT 1       // status: not stitched, k_bwti_literal decides
#define BI_HEADS 8         // walkers G..G+7 start at the primary indexes (only head 0 records bytes)

// dense scratch index of every block that has data (one workgroup; B <= KZ_MAX_BATCH)
__global__ __launch_bounds__(1024) void k_bwti_ord(const int32_t* __restrict__ d_len, int32_t* __restrict__ ord, int B) {
  __shared__ uint32_t lds[32];
  uint32_t carry = 0;
  for (int base = 0; base < B; base += 1024) {
    const int b = base + (int)threadIdx.x;
    const uint32_t a = (b < B && d_len[b] > 0) ? 1u : 0u;
    uint32_t total;
    const uint32_t ex = kz_wg_excl_sum(a, lds, &total);
    if (b < B) ord[b] = a ? (int32_t)(carry + ex) : 0;
    carry += total;
    __syncthreads();
  }
}
__global__ void k_bwti_parse(const u8* __restrict__ src, int64_t stride, const int32_t* __restrict__ d_len, BwtInv V, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int blockSize = d_len[b];
  const u8* s = src + (int64_t)b * stride;
  int status = 0, n = 0, headerSize = 0;
  for (int k = 0; k < 8; k++) V.prim[b * 8 + k] = 0;
  if (blockSize > 0) {
    const u8 mode = s[0];
    const int logNbChunks = (mode >> 2) & 7;
    const int pIndexSize = (mode & 3) + 1;
    const int chunks = 1 << logNbChunks;
    headerSize = 1 + chunks * pIndexSize;
    if (blockSize < headerSize || chunks > 8) status = -KZ_ERR_PROCESS_BLOCK;          // :155-156
    else {
      n = blockSize - headerSize;
      if (chunks != ((n < 256) ? 1 : 8)) status = -KZ_ERR_PROCESS_BLOCK;               // :158-159
      else if (!V.wide && n >= (1 << 24) - 1) status = -KZ_ERR_BLOCK_SIZE;
      else {
        int pos = 1;
        for (int i = 0; i < chunks; i++) {
          long long pi = 0;
          for (int k = 0; k < pIndexSize; k++) pi = (pi << 8) | s[pos++];
          if (pi >= 0x7FFFFFFFLL) { status = -KZ_ERR_PROCESS_BLOCK; break; }
          V.prim[b * 8 + i] = (int32_t)pi + 1;
        }
        if (status == 0 && n >= 2) {
          const int p0 = V.prim[b * 8];
          if (p0 <= 0 || p0 > n) status = -KZ_ERR_PROCESS_BLOCK;                        // BWT.java:261
          if (chunks == 8) for (int k = 0; k < 8; k++) { const int t = V.prim[b * 8 + k] - 1; if (t < 0 || t >= n) status = -KZ_ERR_PROCESS_BLOCK; }  // :305-311
        }
      }
    }
  }
  V.n[b] = (status == 0) ? n : 0;
  V.hdr[b] = headerSize;
  V.status[b] = status;
}

// tile histogram: a thread takes 16 consecutive bytes (one unaligned 16-byte load; the payload starts behind a 25-byte header) and
// adds every RUN of equal bytes once -- after a BWT most bytes sit in runs, so a thread issues a handful of LDS atomics instead of
// the row's eight match-any ballots (round 5: the stage runs beside the other classes' RANK inverses and is issue bound there)
typedef u32 bi_u32x4 __attribute__((ext_vector_type(4)));
typedef bi_u32x4 __attribute__((aligned(1))) bi_u32x4_unaligned;
__global__ __launch_bounds__(KZ_WG) void k_bwti_hist(const u8* __restrict__ src, int64_t stride, BwtInv V) {
  const int b = blockIdx.y, tile = blockIdx.x;
  const int n = V.n[b];
  if ((int64_t)tile * BI_TILE >= n) return;
  __shared__ u32 hist[256];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const u8* s = src + (int64_t)b * stride + V.hdr[b];
  const int p = tile * BI_TILE + (int)threadIdx.x * 16;
  static_assert(BI_TILE == KZ_WG * 16, "one 16-byte piece per thread");
  if (p < n) {
    const int cnt = min(16, n - p);
    bi_u32x4 q = {0, 0, 0, 0};
    if (cnt == 16) q = *(const bi_u32x4_unaligned*)(s + p);              // (slots have >= 4 KiB of slack, but the tail may be poisoned: read it bytewise)
    else for (int j = 0; j < cnt; j++) { const u32 c = s[p + j]; const u32 w = (u32)j >> 2; const u32 sh = 8u * ((u32)j & 3u);
                                          if (w == 0) q.x |= c << sh; else if (w == 1) q.y |= c << sh; else if (w == 2) q.z |= c << sh; else q.w |= c << sh; }
    const u32 w[4] = {q.x, q.y, q.z, q.w};
    u32 runKey = w[0] & 0xFFu, runCnt = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const u32 c = (w[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
      if (j < cnt) {
        if (c == runKey) runCnt++;
        else { atomicAdd(&hist[runKey], runCnt); runKey = c; runCnt = 1; }
      }
    }
    if (runCnt) atomicAdd(&hist[runKey], runCnt);
  }
  __syncthreads();
  V.tileHist[((int64_t)V.ord[b] * V.T + tile) * 256 + threadIdx.x] = hist[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_bwti_scan(BwtInv V) {
  const int b = blockIdx.x;
  const int n = V.n[b];
  const int tiles = (n + BI_TILE - 1) / BI_TILE;
  __shared__ u32 lds[32];
  u32* h = V.tileHist + (int64_t)V.ord[b] * V.T * 256;
  u32 run = 0;
  for (int t = 0; t < tiles; t++) {
    const u32 v = h[(int64_t)t * 256 + threadIdx.x];
    h[(int64_t)t * 256 + threadIdx.x] = run;
    run += v;
  }
  u32 total;
  const u32 ex = kz_wg_excl_sum(run, lds, &total);
  V.bucket[b * 256 + threadIdx.x] = ex;
}

template <bool WIDE>
__global__ __launch_bounds__(KZ_WG) void k_bwti_scatter(const u8* __restrict__ src, int64_t stride, BwtInv V) {
  typedef BiLink<WIDE> LK;
  const int b = blockIdx.y, tile = blockIdx.x;
  const int n = V.n[b];
  if ((int64_t)tile * BI_TILE >= n) return;
  __shared__ u32 cnt[4][256];
  for (int i = threadIdx.x; i < 1024; i += KZ_WG) (&cnt[0][0])[i] = 0;
  __syncthreads();
  const u8* s = src + (int64_t)b * stride + V.hdr[b];
  typename LK::T* data = (typename LK::T*)V.data + (int64_t)V.ord[b] * V.NS;
  const int pIdx = V.prim[b * 8];
  const int wave = threadIdx.x >> 6, lane = kz_lane();
  const int base = tile * BI_TILE + wave * (64 * BI_ITEMS);
  const uint64_t lt = kz_lanemask_lt();
  u32 dr[BI_ITEMS];
#pragma unroll
  for (int r = 0; r < BI_ITEMS; r++) {
    const int idx = base + r * 64 + lane;
    const bool valid = idx < n;
    const u32 d = valid ? s[idx] : 0;
    // a row of one symbol (the usual case in the long runs behind a BWT) needs no match-any
    const uint64_t vm = kz_ballot(valid);
    const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)d);
    const bool uni = kz_ballot(valid && d == d0) == vm;
    const uint64_t peers = uni ? (valid ? vm : 0ULL) : kz_match8(d, valid);
    u32 pre = 0;
    if (valid) pre = cnt[wave][d];
    const u32 rnk = pre + (u32)__popcll(peers & lt);
    if (valid && (peers >> lane) == 1ULL) cnt[wave][d] = pre + (u32)__popcll(peers);
    dr[r] = d | (rnk << 8);
  }
  __syncthreads();
  {
    const int d = threadIdx.x;
    u32 runv = V.bucket[b * 256 + d] + V.tileHist[((int64_t)V.ord[b] * V.T + tile) * 256 + d];
#pragma unroll
    for (int w = 0; w < 4; w++) { const u32 t = cnt[w][d]; cnt[w][d] = runv; runv += t; }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < BI_ITEMS; r++) {
    const int i = base + r * 64 + lane;
    if (i < n) {
      const u32 c = dr[r] & 0xFF;
      const u32 pos = cnt[wave][c] + (dr[r] >> 8);
      // BWT.java:273-293: i<pIdx -> (i-1)<<8|c ; else i<<8|c.  i==0 is the row of the last text byte:
      // the reference stores a dummy link (0xFF00|c); an explicit END marker is stored here instead.
      data[pos] = LK::make((i == 0) ? LK::END : (u32)(i < pIdx ? i - 1 : i), c);
    }
  }
}

// The walk (done ONCE): walker w starts at grid point w*S (w < G) or at the text head t0 (w == G) and follows
// the links to the next grid point / END, counting its steps.  Where its bytes belong in the text is only known
// after k_bwti_resolve, so they are recorded into chunks of BI_CH bytes taken from a per-block pool (8 bytes
// per store; chunkMeta = (walker, sequence number)); k_bwti_copy then moves every chunk to its place with
// coalesced accesses.  A second walk would cost another cache line per byte.
template <bool WIDE>
__global__ __launch_bounds__(64) void k_bwti_walk1(BwtInv V, int b0) {
  typedef BiLink<WIDE> LK;
  const int b = blockIdx.y + b0;
  const int n = V.n[b];
  if (n < 2 || V.status[b] != 0) return;
  const int S = 1 << V.logS;
  const int G = (n + S - 1) >> V.logS;
  const int w = blockIdx.x * 64 + threadIdx.x;
  if (w >= G + BI_HEADS) return;
  const int head = w - G;                                       // >= 0: starts at primary index `head`
  if (head > 0 && n < 256) { V.segLen[(int64_t)V.ord[b] * V.GS + w] = 0; V.segNext[(int64_t)V.ord[b] * V.GS + w] = -1; return; }   // one primary index only
  const bool rec = head <= 0;
  const typename LK::T* data = (const typename LK::T*)V.data + (int64_t)V.ord[b] * V.NS;
  u8* pool = V.pool + (int64_t)V.ord[b] * V.maxChunks * BI_CH;
  uint2* meta = V.chunkMeta + (int64_t)V.ord[b] * V.maxChunks;
  u32 t = (head < 0) ? (u32)w << V.logS : (u32)(V.prim[b * 8 + head] - 1);
  u32 steps = 0;
  int nxt = -2;
  unsigned long long acc = 0;
  u32 fill = BI_CH, seq = 0;              // bytes stored in the current chunk (BI_CH: none allocated yet)
  u8* cp = nullptr;
  bool full = false;
  // Segment lengths are geometric with mean S; a walk this long without meeting a grid point is not a text path.
  const u32 cap = min((u32)n, max(1u << 20, (u32)S << 5));           // (32 x the mean: e^-32 of the segments of a well-formed block)
  while (steps <= cap) {
    if (t >= (u32)n) break;                                   // corrupt link
    const typename LK::T ptr = BI_LD(&data[t]);
    acc |= (unsigned long long)((u32)ptr & 0xFFu) << (8 * (steps & 7u));
    steps++;
    if (rec && (steps & 7u) == 0) {
      if (fill == BI_CH) {
        const u32 id = atomicAdd(&V.chunkCount[b], 1u);
        if (id >= (u32)V.maxChunks) { full = true; break; }
        meta[id] = make_uint2((u32)w, seq++);
        cp = pool + (size_t)id * BI_CH;
        fill = 0;
      }
      *(unsigned long long*)(cp + fill) = acc;
      fill += 8; acc = 0;
    }
    t = LK::next(ptr);
    if (t == LK::END) { nxt = -1; break; }
    if ((t & (u32)(S - 1)) == 0) { nxt = (int)(t >> V.logS); break; }
  }
  if (rec && (steps & 7u) != 0 && !full && nxt != -2) {       // partial tail
    if (fill == BI_CH) {
      const u32 id = atomicAdd(&V.chunkCount[b], 1u);
      if (id >= (u32)V.maxChunks) full = true;
      else { meta[id] = make_uint2((u32)w, seq++); cp = pool + (size_t)id * BI_CH; fill = 0; }
    }
    if (!full) *(unsigned long long*)(cp + fill) = acc;
  }
  if (nxt == -2 || full) { atomicCAS(&V.status[b], 0, BI_SUSPECT); nxt = -1; }
  V.segLen[(int64_t)V.ord[b] * V.GS + w] = steps;
  V.segNext[(int64_t)V.ord[b] * V.GS + w] = nxt;
}

// per block: the segments' text offsets = suffix sums of the segment lengths along the chain head -> ... -> END.
// List ranking by a ruling set (round 5: up to 16 K segments of ~256 bytes): every 16th grid segment and every head is a
// SPLITTER; a thread walks from its splitter to the next one summing lengths (~16 dependent loads from the block's 200 KB of
// segment tables, L2 resident), the ~1000 splitters are ranked by pointer jumping in LDS (11 rounds), and a second walk hands
// every segment its offset.  (Plain pointer jumping over all segments needs them in LDS: a 131 KB, 1024-thread workgroup per
// block took 20 ms per class beside the other classes' RANK inverses; this form keeps 6 KB.)
#define BI_MAXSEG 16448
#define BI_RES_WG 256
#define BI_SPL 16
#define BI_MAXSPL ((BI_MAXSEG + BI_SPL - 1) / BI_SPL + BI_HEADS + 4)
#define BI_RES_ITEMS ((BI_MAXSPL + BI_RES_WG - 1) / BI_RES_WG)
#define BI_NONE 0xFFFFu
__global__ __launch_bounds__(BI_RES_WG) void k_bwti_resolve(BwtInv V, int b0) {
  const int b = blockIdx.x + b0;
  const int n = V.n[b];
  if (n < 2 || V.status[b] != 0) return;
  const int S = 1 << V.logS;
  const int G = (n + S - 1) >> V.logS;
  const int M = G + BI_HEADS;
  __shared__ u32 spT[BI_MAXSPL];            // splitter: bytes of its sublist, then bytes from its start to the end of the text
  __shared__ uint16_t spN[BI_MAXSPL];       // splitter: next splitter (index into the splitter list), BI_NONE = end of text
  __shared__ int bad;
  const int64_t o = (int64_t)V.ord[b] * V.GS;
  const u32* __restrict__ segLen = V.segLen + o;
  const int32_t* __restrict__ segNext = V.segNext + o;
  u32* __restrict__ segOff = V.segOff + o;
  if (threadIdx.x == 0) bad = 0;
  for (int i = threadIdx.x; i < M; i += BI_RES_WG) segOff[i] = 0xFFFFFFFFu;   // segments no splitter reaches are not on the text path
  __syncthreads();
  const int NG = (G + BI_SPL - 1) / BI_SPL;              // splitters among the grid segments: 0, 16, 32, ...
  const int NSP = NG + BI_HEADS;                         // ... and the heads G .. G+7
  // 1. sublist sums
  for (int k = threadIdx.x; k < NSP; k += BI_RES_WG) {
    int v = k < NG ? k * BI_SPL : G + (k - NG);
    u32 sum = 0, next = BI_NONE;
    for (int steps = 0;; steps++) {
      sum += segLen[v];
      const int nx = segNext[v];
      if (nx < 0 || nx >= M) break;
      if (nx >= G || (nx % BI_SPL) == 0) { next = nx >= G ? (u32)(NG + (nx - G)) : (u32)(nx / BI_SPL); break; }
      if (steps > M) { bad = 1; break; }                  // a cycle that holds no splitter: not a text path
      v = nx;
    }
    spT[k] = sum; spN[k] = (uint16_t)next;
  }
  __syncthreads();
  // 2. suffix sums over the splitters (pointer jumping)
  int rounds = 1;
  while ((1 << rounds) < NSP) rounds++;
  for (int round = 0; round < rounds; round++) {
    u32 t2[BI_RES_ITEMS]; u32 n2[BI_RES_ITEMS];
#pragma unroll
    for (int q = 0; q < BI_RES_ITEMS; q++) {
      const int k = (int)threadIdx.x + q * BI_RES_WG;
      if (k < NSP) {
        const u32 nx = spN[k];
        t2[q] = nx != BI_NONE ? spT[k] + spT[nx] : spT[k];
        n2[q] = nx != BI_NONE ? (u32)spN[nx] : (u32)BI_NONE;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < BI_RES_ITEMS; q++) {
      const int k = (int)threadIdx.x + q * BI_RES_WG;
      if (k < NSP) { spT[k] = t2[q]; spN[k] = (uint16_t)n2[q]; }
    }
    __syncthreads();
  }
  // 3. every segment of a sublist: its text offset.  (Heads 1..7 start inside another splitter's sublist -- the grid segments
  //    behind them are that splitter's to number; every writer of a shared segment would store the same value anyway.)
  for (int k = threadIdx.x; k < NSP; k += BI_RES_WG) {
    int v = k < NG ? k * BI_SPL : G + (k - NG);
    u32 running = spT[k];                                 // bytes from the start of v to the end of the text
    for (int steps = 0;; steps++) {
      segOff[v] = running <= (u32)n ? (u32)n - running : 0xFFFFFFFFu;
      if (k > NG) break;                                  // heads 1..7: only their own offset (for the check below)
      running -= segLen[v];
      const int nx = segNext[v];
      if (nx < 0 || nx >= G || (nx % BI_SPL) == 0 || steps > M) break;
      v = nx;
    }
  }
  // Stitching is right when head 0 reaches END after exactly n steps (then every link is on that path) and, with 8 primary
  // indexes, head k sits k*ckSize steps into it (BWT.java:296-314); otherwise the literal walkers take over.
  if (threadIdx.x < BI_HEADS) {
    const int k = threadIdx.x;
    const bool atEnd = spN[NG + k] == BI_NONE;             // its chain of splitters ends at END (not in a cycle)
    const u32 Rk = spT[NG + k];
    bool good = true;
    if (k == 0) good = (Rk == (u32)n && atEnd && !bad);
    else if (n >= 256) {
      const u32 ckSize = (u32)(((n & 7) == 0) ? (n >> 3) : (n >> 3) + 1);
      good = (atEnd && Rk <= (u32)n && (u32)n - Rk == (u32)k * ckSize);
    }
    if (!good) atomicCAS(&V.status[b], 0, BI_SUSPECT);
  }
}

// move every recorded chunk to its place in the text: BI_CH / 4 lanes per chunk, 4 bytes per lane
typedef u32 __attribute__((aligned(1))) bi_u32_unaligned;
#define BI_CPL (BI_CH / 4)                 // lanes per chunk
#define BI_CPW (256 / BI_CPL)              // chunks per workgroup
__global__ __launch_bounds__(256) void k_bwti_copy(u8* __restrict__ dst, int64_t stride, BwtInv V, int b0) {
  const int b = blockIdx.y + b0;
  const int n = V.n[b];
  if (n < 2 || V.status[b] != 0) return;
  const u32 id = blockIdx.x * BI_CPW + (threadIdx.x / BI_CPL);
  if (id >= V.chunkCount[b] || id >= (u32)V.maxChunks) return;
  const int lane = threadIdx.x % BI_CPL;
  const uint2 m = V.chunkMeta[(int64_t)V.ord[b] * V.maxChunks + id];
  const u32 len = V.segLen[(int64_t)V.ord[b] * V.GS + m.x];
  const u32 off = V.segOff[(int64_t)V.ord[b] * V.GS + m.x];
  if (off == 0xFFFFFFFFu || (unsigned long long)off + len > (unsigned long long)n) return;   // not on the text path
  const u32 start = m.y * BI_CH;
  if (start >= len) return;
  const u32 cnt = min((u32)BI_CH, len - start);
  const u8* src = V.pool + ((int64_t)V.ord[b] * V.maxChunks + id) * BI_CH;
  u8* d = dst + (int64_t)b * stride + off + start;
  const u32 k = 4u * (u32)lane;
  if (k + 4 <= cnt) *(bi_u32_unaligned*)(d + k) = *(const u32*)(src + k);
  else for (u32 q = k; q < cnt; q++) d[q] = src[q];
}

// The reference's walkers, literally (BWT.java:295-368): lane k follows the links from primary index k for its
// share of the text; the link behind row 0 is the reference's dummy 0xFF (stored as BI_END here).  Only for blocks
// k_bwti_resolve would not stitch.
template <bool WIDE>
__global__ __launch_bounds__(64) void k_bwti_literal(u8* __restrict__ dst, int64_t stride, BwtInv V) {
  typedef BiLink<WIDE> LK;
  const int b = blockIdx.x;
  if (V.status[b] != BI_SUSPECT) return;
  const int n = V.n[b];
  const int lane = threadIdx.x;
  const typename LK::T* data = (const typename LK::T*)V.data + (int64_t)V.ord[b] * V.NS;
  u8* o = dst + (int64_t)b * stride;
  bool fail = false;
  const int walkers = (n < 256) ? 1 : 8;
  if (lane < walkers) {
    const int ckSize = (walkers == 1) ? n : (((n & 7) == 0) ? (n >> 3) : (n >> 3) + 1);
    const int steps = (lane < 7) ? ckSize : n - 7 * ckSize;
    u32 t = (u32)(V.prim[b * 8 + lane] - 1);
    u8* q = o + (int64_t)lane * ckSize;
    for (int i = 0; i < steps; i++) {
      if (t >= (u32)n) { fail = true; break; }                 // Java: data[t] out of range throws
      const typename LK::T ptr = data[t];
      q[i] = (u8)ptr;
      t = LK::next(ptr);
      if (t == LK::END) t = 0xFF;
    }
  }
  const bool anyFail = kz_ballot(fail) != 0;
  if (lane == 0) V.status[b] = anyFail ? -KZ_ERR_PROCESS_BLOCK : 0;
}

__global__ void k_bwti_fin(const u8* __restrict__ src, u8* __restrict__ dst, int64_t stride, BwtInv V,
                           int32_t* __restrict__ d_len2, int32_t* __restrict__ d_flag, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int n = V.n[b];
  const bool ok = V.status[b] == 0;
  d_len2[b] = ok ? n : 0;
  d_flag[b] = ok ? 1 : 0;
  if (ok && n == 1) dst[(int64_t)b * stride] = src[(int64_t)b * stride + V.hdr[b]];      // BWT.java:174-177 mirror
}

size_t kz_bwt_inverse_scratch(int B, int maxN) {
  const int64_t NS = (int64_t)kz_align((size_t)maxN + 64, BI_TILE);
  const int T = (int)(NS / BI_TILE);
  const size_t maxChunks = (size_t)NS / BI_CH + BI_MAXSEG + 8;
  return (size_t)B * ((size_t)NS * (maxN >= (1 << 24) - 1 ? 8 : 4) + (size_t)T * 1024 + 1024 + 64 * 4 + (size_t)BI_MAXSEG * 12 + maxChunks * (BI_CH + 8) + 512) + 16384;
}

int kz_stage_bwt_inverse(kz_ctx* ctx, kz_batch& bt) {
  const int B = bt.B;
  int maxN = 0;
  for (int b = 0; b < B; b++) if (bt.h_len[b] > maxN) maxN = bt.h_len[b];
  BwtInv V;
  V.NS = (int64_t)kz_align((size_t)(maxN > 0 ? maxN : 1), BI_TILE);
  V.T = (int)(V.NS / BI_TILE);
  // the large arrays only exist for the blocks that have data: a lengths-masked view of a batch (kz_api.hip: one cost class
  // of the decoder's overlapped schedule) takes its share of the scratch, and the classes' stages can be in flight together
  int A = 0;
  for (int b = 0; b < B; b++) if (bt.h_len[b] > 0) A++;
  if (A < 1) A = 1;
  int32_t* d_ord = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  V.ord = d_ord;
  V.wide = maxN >= (1 << 24) - 1 ? 1 : 0;                            // (maxN counts the header: a block this long may hold n >= 2^24-1)
  V.data = kz_arena_alloc(ctx, (size_t)V.NS * A * (V.wide ? 8 : 4));
  V.tileHist = (u32*)kz_arena_alloc(ctx, (size_t)V.T * A * 1024);
  V.bucket = (u32*)kz_arena_alloc(ctx, (size_t)B * 1024);
  V.n = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  V.hdr = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  V.prim = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 32);
  V.status = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  V.logS = 6;
  while (((maxN + (1 << V.logS) - 1) >> V.logS) > BI_MAXSEG - BI_HEADS - 8) V.logS++;
  V.GS = ((maxN + (1 << V.logS) - 1) >> V.logS) + BI_HEADS;
  V.segLen = (u32*)kz_arena_alloc(ctx, (size_t)A * V.GS * 4);
  V.segNext = (int32_t*)kz_arena_alloc(ctx, (size_t)A * V.GS * 4);
  V.segOff = (u32*)kz_arena_alloc(ctx, (size_t)A * V.GS * 4);
  V.maxChunks = (int)(V.NS / BI_CH) + V.GS + 4;
  V.pool = (u8*)kz_arena_alloc(ctx, (size_t)A * V.maxChunks * BI_CH);
  V.chunkMeta = (uint2*)kz_arena_alloc(ctx, (size_t)A * V.maxChunks * sizeof(uint2));
  V.chunkCount = (u32*)kz_arena_alloc(ctx, (size_t)B * 4);
  if (!V.status || !V.data || !V.segOff || !V.chunkCount || !V.pool || !V.chunkMeta || !d_ord) { snprintf(ctx->err, sizeof(ctx->err), "bwt_inverse: arena overflow"); return -KZ_ERR_DEVICE; }
  hipStream_t st = ctx->stream;
  const u8* src = bt.buf[bt.cur];
  u8* dst = bt.buf[bt.cur ^ 1];
  KZ_HIP(hipMemsetAsync(V.chunkCount, 0, (size_t)B * 4, st));
  KZ_LAUNCH(ctx, KID_BWTI_PARSE, k_bwti_ord, dim3(1), dim3(1024), bt.d_len, d_ord, B);
  KZ_LAUNCH(ctx, KID_BWTI_PARSE, k_bwti_parse, dim3((B + 63) / 64), dim3(64), src, bt.stride, bt.d_len, V, B);
  const int tiles = (maxN + BI_TILE - 1) / BI_TILE;
  if (tiles > 0) {
    KZ_LAUNCH(ctx, KID_BWTI_HIST, k_bwti_hist, dim3(tiles, B), dim3(KZ_WG), src, bt.stride, V);
    KZ_LAUNCH(ctx, KID_BWTI_SCAN, k_bwti_scan, dim3(B), dim3(256), V);
    if (V.wide) { KZ_LAUNCH(ctx, KID_BWTI_SCATTER, k_bwti_scatter<true>, dim3(tiles, B), dim3(KZ_WG), src, bt.stride, V); }
    else { KZ_LAUNCH(ctx, KID_BWTI_SCATTER, k_bwti_scatter<false>, dim3(tiles, B), dim3(KZ_WG), src, bt.stride, V); }
  }
  if (maxN >= 2) {
    // The walks are dependent random 4-byte loads over a 4n-byte link array per block: bandwidth bound at one
    // cache line per step once enough walkers are in flight.  Walking the batch in small groups (so that the
    // group's link arrays stay in the 256 MB Infinity Cache) was measured 4-7x SLOWER: fewer walkers in
    // flight (and segment lengths are geometric, so most lanes of a wave idle behind its longest segment).
    // The whole batch is walked at once.
    const int group = B;
    for (int b0 = 0; b0 < B; b0 += group) {
      const int nb = (B - b0 < group) ? B - b0 : group;
      if (V.wide) { KZ_LAUNCH(ctx, KID_BWTI_WALK1, k_bwti_walk1<true>, dim3((V.GS + 63) / 64, nb), dim3(64), V, b0); }
      else { KZ_LAUNCH(ctx, KID_BWTI_WALK1, k_bwti_walk1<false>, dim3((V.GS + 63) / 64, nb), dim3(64), V, b0); }
      KZ_LAUNCH(ctx, KID_BWTI_RESOLVE, k_bwti_resolve, dim3(nb), dim3(BI_RES_WG), V, b0);
      KZ_LAUNCH(ctx, KID_BWTI_COPY, k_bwti_copy, dim3((V.maxChunks + BI_CPW - 1) / BI_CPW, nb), dim3(256), dst, bt.stride, V, b0);
    }
    if (V.wide) { KZ_LAUNCH(ctx, KID_BWTI_LITERAL, k_bwti_literal<true>, dim3(B), dim3(64), dst, bt.stride, V); }
    else { KZ_LAUNCH(ctx, KID_BWTI_LITERAL, k_bwti_literal<false>, dim3(B), dim3(64), dst, bt.stride, V); }
  }
  KZ_LAUNCH(ctx, KID_BWTI_FIN, k_bwti_fin, dim3((B + 255) / 256), dim3(256), src, dst, bt.stride, V, bt.d_len2, bt.d_flag, B);
  KZ_HIP(hipGetLastError());
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}
