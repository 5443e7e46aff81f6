// kz_internal.h -- host-side context + stage interfaces (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <set>
#include <mutex>
#include <thread>
#include "../../include/kanzi_hip.h"

enum KzKernelId { KID_ANS_ENC_CHUNK, KID_ANS_ENC_SCAN, KID_ANS_ENC_CONCAT, KID_ANS_DEC_INDEX, KID_ANS_DEC_CHUNK, KID_ANS_DEC_FIN, KID_MASK_LEN, KID_PASSTHROUGH, KID_FRAME_PREPARE, KID_COPY_BYTES, KID_FRAME_DECIDE, KID_FRAME_HEADER, KID_FRAME_PARSE, KID_COPY_PAYLOAD, KID_BWT_INIT, KID_RADIX_HIST, KID_RADIX_SCAN, KID_RADIX_SCATTER, KID_SEG_REDUCE, KID_SEG_SCAN, KID_SEG_APPLY, KID_LIVE_EMIT, KID_BWT_EMIT, KID_BWTI_PARSE, KID_BWTI_HIST, KID_BWTI_SCAN, KID_BWTI_SCATTER, KID_BWTI_WALK1, KID_BWTI_RESOLVE, KID_BWTI_COPY, KID_BWTI_LITERAL, KID_BWTI_FIN, KID_SBRT_LAST2, KID_SBRT_SCAN, KID_SBRT_REPLAY, KID_COPY_LEN, KID_SBRT_INVERSE, KID_ZRLT_F1, KID_ZRLT_F2, KID_ZRLT_F3, KID_ZRLT_FFIN, KID_ZRLT_I1, KID_ZRLT_I2, KID_ZRLT_I3, KID_ZRLT_IFIN, KID_HUF_ENC_CHUNK, KID_HUF_DEC_INDEX, KID_HUF_DEC_CHUNK, KID_HUF_DEC_FIN, KID_FPAQ_ENC, KID_FPAQ_PACK, KID_FPAQ_DEC, KID_SRT_HIST, KID_SRT_PREP, KID_SRT_SCATTER, KID_SRT_INV, KID_LZ_FWD, KID_LZ_INV, KID_XXHASH, KID_BLOCK_MAGIC, KID_MM_ANALYZE, KID_MM_EMIT, KID_MM_CHECK, KID_MM_INV, KID_ALIAS_ANALYZE, KID_ALIAS_HIST1, KID_ALIAS_SELECT, KID_ALIAS_EMIT, KID_ALIAS_INV, KID_SKIP_DECIDE, KID_MSD_HIST, KID_MSD_SCAN, KID_MSD_SCATTER, KID_BUCKET_SORT, KID_BUCKET_COUNT, KID_BUCKET_COUNT_S, KID_TR_HIST16, KID_TR_ASSIGN, KID_TR_COUNT, KID_TR_SCATTER, KID_TR_SORT, KID_TEXT_INV, KID_UTF_INV, KID_TEXT_FWD, KID_TEXT_WALK, KID_UTF_FWD, KID_COUNT };
#define KZ_KERNEL_NAMES { "k_ans_enc_chunk", "k_ans_enc_scan", "k_ans_enc_concat", "k_ans_dec_index", "k_ans_dec_chunk", "k_ans_dec_fin", "k_mask_len", "k_passthrough", "k_frame_prepare", "k_copy_bytes", "k_frame_decide", "k_frame_header", "k_frame_parse", "k_copy_payload", "k_bwt_init", "k_radix_hist", "k_radix_scan", "k_radix_scatter", "k_seg_reduce", "k_seg_scan", "k_seg_apply", "k_live_emit", "k_bwt_emit", "k_bwti_parse", "k_bwti_hist", "k_bwti_scan", "k_bwti_scatter", "k_bwti_walk1", "k_bwti_resolve", "k_bwti_copy", "k_bwti_literal", "k_bwti_fin", "k_sbrt_last2", "k_sbrt_scan", "k_sbrt_replay", "k_copy_len", "k_sbrt_inverse", "k_zrlt_f1", "k_zrlt_f2", "k_zrlt_f3", "k_zrlt_ffin", "k_zrlt_i1", "k_zrlt_i2", "k_zrlt_i3", "k_zrlt_ifin", "k_huf_enc_chunk", "k_huf_dec_index", "k_huf_dec_chunk", "k_huf_dec_fin", "k_fpaq_enc", "k_fpaq_pack", "k_fpaq_dec", "k_srt_hist", "k_srt_prep", "k_srt_scatter", "k_srt_inv", "k_lz_fwd", "k_lz_inv", "k_xxhash", "k_block_magic", "k_mm_analyze", "k_mm_emit", "k_mm_check", "k_mm_inv", "k_alias_analyze", "k_alias_hist1", "k_alias_select", "k_alias_emit", "k_alias_inv", "k_skip_decide", "k_msd_hist", "k_msd_scan", "k_msd_scatter", "k_bucket_sort", "k_bucket_count", "k_bucket_count_s", "k_tr_hist16", "k_tr_assign", "k_tr_count", "k_tr_scatter", "k_tr_sort", "k_text_inv", "k_utf_inv", "k_text_fwd", "k_text_walk", "k_utf_fwd" }
// Environment switches (diagnostics, A/B runs, the tests' forced schedules; none is needed for normal use).  Read ONCE, when the
// context is created (kz_switches_read, kz_api.hip); kz_ctx_reload_switches re-reads them for a live context (tests, A/B tools).
// Nothing on a call's path calls getenv.
struct kz_switches {
  int blockingWaits = -1;        // KZ_BLOCKING_WAITS: -1 unset (decided from the CPU quota), 0 / 1
  int textGpu = -1;              // KZ_TEXT_GPU: -1 unset, 0 host, 1 rows x 3 waves, 2 serial walk, 3 rows x 1 wave
  int textGpuMin = 512;          // KZ_TEXT_GPU_MIN
  int textFwdGpu = -1;           // KZ_TEXT_FWD_GPU: -1 unset, 0 host, 1 any batch
  int textFwdGpuMin = 256;       // KZ_TEXT_FWD_GPU_MIN
  int utfGpu = 1;                // KZ_UTF_GPU=0: UTF inverse on the host
  int utfFwdGpu = -1;            // KZ_UTF_FWD_GPU: -1 unset (with the device TEXT forward), 0 host, 1 any batch
  int textGpuTrace = 0;          // KZ_TEXT_GPU_TRACE
  int fuseMinBlocks = 32;        // KZ_FUSE_MIN_BLOCKS: smaller batches run the decoder's stages one after the other
  int overlapClasses = 3;        // KZ_OVERLAP_CLASSES
  int wideQueues = -1;           // KZ_WIDE_QUEUES: -1 = measured once per context
  int noSFirst = 0;              // KZ_NO_SFIRST
  int traceSched = 0;            // KZ_TRACE_SCHED
  int tracePipe = 0;             // KZ_TRACE_PIPE
  int hostChunk = 0;             // KZ_HOST_CHUNK (0: by the batch)
  int hostChunkDec = 512;        // KZ_HOST_CHUNK_DEC
  int hostInvStaged = -1;        // KZ_HOST_INV_STAGED: -1 unset
  int streamChunk = 0;           // KZ_STREAM_CHUNK (0: default)
  int streamSerial = 0;          // KZ_STREAM_SERIAL
  int bwtTrie = -1, bwtTrieWin = -1, bwtBuckets = -1, bwtDmax = -1, bwtRetire = -1;   // KZ_BWT_TRIE / _TRIEWIN / _BUCKETS / _DMAX / _RETIRE (-1 unset)
  int bwtLazyRank = -1;         // KZ_BWT_LAZYRANK=0: round 0 stores every rank (A/B; see k_tr_sort)
  int bwtTrace = 0;              // KZ_BWT_TRACE
  int bwtTestTrieOverflow = -1;  // KZ_BWT_TEST_TRIE_OVERFLOW=<round> (tests)
  int fpaqForce = 0;             // KZ_FPAQ_FORCE: 0 unset, 1 wave, 2 lane
  int sbrtForm = -1;             // KZ_SBRT_FORM: -1 default, 0 = the 32-bit list forms of rounds 2-5 (A/B)
};
void kz_switches_read(kz_switches& s);   // kz_api.hip

struct KzPending { hipEvent_t e0, e1; int id; };
struct kz_ctx {
  int device = 0;
  kz_switches sw;                // environment switches as read at creation
  hipStream_t stream = nullptr;
  hipStream_t side[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};     // side streams of the overlapped RANK-inverse / BWT-inverse schedule (created on first use)
  hipEvent_t evJoin[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int wideQueues = -1;           // main + three side streams run side by side (-1: not measured yet)
  int wideNow = 0;               // what the current call uses (sw.wideQueues overrides the measurement)
  // grow-only device arena, bump-allocated per API call
  uint8_t* arena = nullptr;
  size_t arenaCap = 0, arenaTop = 0;
  // pinned host staging for small read-backs
  int32_t* hpin = nullptr;       // hpinInts ints, grown by kz_hpin_reserve
  size_t hpinInts = 0;
  char err[512] = {0};
  // last-call stage timings (ms) for bench/roofline (hipEvent based)
  float stageMs[KZ_MAX_STAGES] = {0};
  int64_t stageAlgBytes[KZ_MAX_STAGES] = {0};
  int nStages = 0;
  int checksum = 0;              // 0 none, 1 XXHash32, 2 XXHash64 (ctx map key "checksum")
  int skipBlocks = 0;            // ctx map key "skipBlocks": store incompressible-looking blocks as copy blocks
  int dataType = 0;              // ctx map key "dataType" for the single-block calls (Global.DataType, KZ_DT_*)
  int blockSize = 4 * 1024 * 1024;   // ctx map key "blockSize" (TEXT sizes its hash map by it)
  bool blockSizeSet = false;     // kz_ctx_set_block_size was called: chains with TEXT refuse to guess
  int entropy = KZ_E_NONE;       // ctx map key "entropy" for the single-block calls (TEXT: TextCodec1 / TextCodec2)
  int numCUs = 256;              // compute units of the device (placement of the serial-per-block kernels)
  long long* d_endBits = nullptr; // optional [B] device array: bit position behind each block's entropy payload (kz_entropy_decode)
  bool timing = false;
  // per-kernel event timing (bench roofline): events recorded on ctx->stream around every launch
  bool ktiming = false;
  std::vector<KzPending> pending;
  std::vector<hipEvent_t> evPool;
  double kMs[KID_COUNT] = {0};
  double kMaxMs[KID_COUNT] = {0};     // longest single launch (launches of one kernel on several streams overlap)
  long long kLaunches[KID_COUNT] = {0};
  // asynchronous batches (kz_submit_* / kz_wait): one worker thread per context runs the queued calls in order
  std::thread worker;
  std::mutex qmu;
  std::condition_variable qcv;
  std::deque<std::pair<int64_t, std::function<int32_t()>>> queue;
  std::map<int64_t, int32_t> finished;
  std::set<int64_t> collected;   // ids kz_wait has handed out (>= collectedBelow; everything below is collected too)
  int64_t collectedBelow = 1;
  int64_t nextJob = 1;
  bool stopWorker = false;
  // staging of the host-buffer stream calls (kz_compress / kz_decompress, kz_stream.hip): grow-only, two slots each
  struct Stage { uint8_t* p = nullptr; size_t cap = 0; };
  Stage pinIn[2], pinOut[2];       // pinned host memory (hipHostMalloc)
  Stage devIn[2], devOut[2];       // HBM outside the arena (the arena is reset by every batched call)
  Stage hsIn[2], hsOut[2];         // pinned: the host-stage pipeline's copy of a chunk's blocks / the stages' outputs (kz_api.hip)
  Stage tfIn, tfOut[2];           // pinned: the device TEXT forward's host passes (kz_encode_blocks_pre).  Their own buffers and
  hipStream_t tfCopy = nullptr;    // stream: kz_compress's upload thread fills hsOut[k & 1] / drains on copyDown for the NEXT chunk meanwhile
  hipStream_t copyUp = nullptr, copyDown = nullptr;
  Stage hiAux[2];                  // HBM: per-block lengths / masks of the decoder's host-inverse chunks (kz_api.hip)
  hipStream_t hiStream[2] = {nullptr, nullptr};   // their gather / scatter kernels run beside the main stream's next chunk
  // waits of the context's own thread: 1 = block on an event (hipEventBlockingSync) instead of hipStreamSynchronize's spin loop.
  // Chosen at creation: on when the process is CPU-quota limited (a spinning waiter burns a whole CPU of the quota that the
  // host stages -- or the other ranks of the node -- could use), KZ_BLOCKING_WAITS=0/1 overrides.
  int blockingWaits = 0;
  hipEvent_t evBlock = nullptr;
};

// scratch taken from the context's arena inside one function: given back on every way out (KZ_HIP returns early)
struct kz_arena_guard { kz_ctx* c; size_t mark; ~kz_arena_guard() { c->arenaTop = mark; } };

#define KZ_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  snprintf(ctx->err, sizeof(ctx->err), "%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
  return -KZ_ERR_DEVICE; } } while (0)

// wait for everything queued on `st` (the context's thread only)
static inline hipError_t kz_stream_sync(kz_ctx* ctx, hipStream_t st) {
  if (!ctx->blockingWaits) return hipStreamSynchronize(st);
  if (!ctx->evBlock) { hipError_t e = hipEventCreateWithFlags(&ctx->evBlock, hipEventBlockingSync | hipEventDisableTiming); if (e != hipSuccess) return e; }
  hipError_t e = hipEventRecord(ctx->evBlock, st);
  if (e != hipSuccess) return e;
  return hipEventSynchronize(ctx->evBlock);
}

#define KZ_MAX_PACKED_BLOCK ((1 << 24) - 257)   // largest block the packed inverse BWT / RANK kernels take (longer ones: their wide forms)
#define KZ_MAX_BLOCK ((1 << 30) + 1057)         // a 1 GiB block (BWT.java:59, the stream header's limit) plus the longest stage header
#define KZ_MAX_BATCH 65535        // blocks per launch: the batched kernels carry the block index in gridDim.y (<= 65535)
int kz_hpin_reserve(kz_ctx* ctx, size_t ints);
static inline size_t kz_align(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Reserve `total` bytes of arena (may reallocate: only call before any sub-allocation).
int kz_arena_reserve(kz_ctx* ctx, size_t total);
void* kz_arena_alloc(kz_ctx* ctx, size_t bytes);   // 256-byte aligned; nullptr on overflow

// A batch of independent blocks resident in HBM: block b lives at base + b*stride.
struct kz_batch {
  int B = 0;            // blocks in the batch
  int maxN = 0;         // largest original block length
  int64_t stride = 0;   // bytes between blocks in each ping-pong buffer
  uint8_t* buf[2] = {nullptr, nullptr};
  int cur = 0;          // which buffer holds the current data
  int32_t* d_len = nullptr;     // [B] current length per block (device)
  int32_t* d_len2 = nullptr;    // [B] next length (stages write here, then swap)
  int32_t* d_flag = nullptr;    // [B] per-stage applied flag (device)
  int32_t* d_dtype = nullptr;   // [B] Global.DataType of each block: the reference's per-block context entry "dataType"
  std::vector<int32_t> h_len;   // host mirror of d_len
  int slotRot = 0;              // rotation of the SIMD assignment in kz_place_blocks (views launched side by side start on different SIMDs)
  int prio = 0;                 // issue priority of the serial-per-block kernels launched for this view (overlapped decoder schedule)
  std::vector<int32_t> h_cost;  // optional per-block cost hint for serial-per-block stages (length at the previous stage's input)
};

struct KzPlacement { int wpg = 1; int R = 1; std::vector<int> G, off; const int32_t* d_order = nullptr; };
int kz_place_blocks(kz_ctx*, const kz_batch&, KzPlacement&);     // kz_sbrt.hip: cost-aware placement of one-wave-per-block kernels

// ---- stages (each works on the whole batch; returns 0 or -KZ_ERR_*) ----
// forward: reads batch.buf[cur] / d_len, writes buf[cur^1] / d_len and d_flag[b]=1 applied, 0 declined
int kz_stage_bwt_forward(kz_ctx*, kz_batch&);
int kz_stage_bwt_inverse(kz_ctx*, kz_batch&);
int kz_stage_sbrt_forward(kz_ctx*, kz_batch&, int mode);
int kz_stage_sbrt_inverse(kz_ctx*, kz_batch&, int mode);
int kz_sbrt_ranks(kz_ctx*, const uint8_t* src, uint8_t* dst, int64_t stride, const int32_t* d_len, int B, int maxN, int mode);
int kz_stage_srt_forward(kz_ctx*, kz_batch&);
int kz_stage_srt_inverse(kz_ctx*, kz_batch&);
int kz_stage_lz_forward(kz_ctx*, kz_batch&, int extra);
int kz_stage_mm_forward(kz_ctx*, kz_batch&);                    // FSDCodec (kz_mm.hip); reads and writes batch.d_dtype
int kz_stage_mm_inverse(kz_ctx*, kz_batch&, int dstCap);
size_t kz_mm_scratch(int B, int maxN);
int kz_stage_alias_forward(kz_ctx*, kz_batch&, int onlyDNA);    // AliasCodec = PACK / DNA (kz_alias.hip); reads and writes batch.d_dtype
int kz_stage_alias_inverse(kz_ctx*, kz_batch&, int dstCap);
size_t kz_alias_scratch(int B, int maxN, bool decode);
int kz_block_data_types(kz_ctx*, kz_batch&, int init, bool sniff);
int kz_skip_block_flags(kz_ctx*, kz_batch&, int32_t* d_skip);   // the writer's "skipBlocks" test per block (kz_mm.hip)   // batch.d_dtype = init, or the writer's Magic tag per block
int kz_stage_lz_inverse(kz_ctx*, kz_batch&, int extra, int dstCap);
int kz_stage_fpaq_encode(kz_ctx*, kz_batch&, uint8_t* out, int64_t outStride, const int32_t* d_hdrBytes, int64_t* d_bits);
int kz_stage_fpaq_decode(kz_ctx*, kz_batch&, const uint8_t* in, int64_t inStride, const int64_t* d_bitOff, const int64_t* d_bitEnd);
int kz_block_hashes(kz_ctx*, const uint8_t* data, int64_t stride, const int32_t* d_len, int B, int kind, unsigned long long* d_hash);
size_t kz_srt_scratch(int B, int maxN);
size_t kz_fpaq_scratch(int B, int maxN);
size_t kz_lz_scratch(int B, int maxN);
int kz_stage_zrlt_forward(kz_ctx*, kz_batch&);
int kz_stage_zrlt_inverse(kz_ctx*, kz_batch&, int dstCap);
// entropy: reads buf[cur]; writes per-block bitstring at out + b*outStride starting at byte offset
// d_hdrBytes[b]; d_bits[b] = payload bits produced
int kz_stage_ans0_encode(kz_ctx*, kz_batch&, uint8_t* out, int64_t outStride, const int32_t* d_hdrBytes, int64_t* d_bits);
// decode: in = bitstrings (payload starts at bit offset d_bitOff[b]); count d_len[b]; writes buf[cur^1]
int kz_stage_ans0_decode(kz_ctx*, kz_batch&, const uint8_t* in, int64_t inStride, const int64_t* d_bitOff, const int64_t* d_bitEnd);
int kz_stage_huffman_encode(kz_ctx*, kz_batch&, uint8_t* out, int64_t outStride, const int32_t* d_hdrBytes, int64_t* d_bits);
int kz_stage_huffman_decode(kz_ctx*, kz_batch&, const uint8_t* in, int64_t inStride, const int64_t* d_bitOff, const int64_t* d_bitEnd);

size_t kz_bwt_forward_scratch(int B, int maxN);
size_t kz_bwt_inverse_scratch(int B, int maxN);
size_t kz_sbrt_scratch(int B, int maxN);
size_t kz_zrlt_scratch(int B, int maxN);
size_t kz_ans_scratch(int B, int maxN);

// per-kernel timing
hipEvent_t kz_ev(kz_ctx* ctx);
void kz_ktimer_flush(kz_ctx* ctx);     // call after the stream has been synchronized
#define KZ_LAUNCH(ctx, kid, kernel, grid, block, ...) do { \
  hipEvent_t a_ = nullptr, b_ = nullptr; \
  if ((ctx)->ktiming) { a_ = kz_ev(ctx); b_ = kz_ev(ctx); (void)hipEventRecord(a_, (ctx)->stream); } \
  hipLaunchKernelGGL(kernel, grid, block, 0, (ctx)->stream, __VA_ARGS__); \
  if ((ctx)->ktiming) { (void)hipEventRecord(b_, (ctx)->stream); (ctx)->pending.push_back({a_, b_, kid}); } } while (0)

// ---- host (CPU) stages in front of the GPU chain: TEXT and UTF (kz_text.hip) ----
bool kz_is_host_transform(int type);
int kz_host_block_data_type(const uint8_t* p, int n, int init);
int kz_host_transform_forward(int type, int entropyType, int blockSize, int* dataType, const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced);
int kz_host_transform_inverse(int type, int blockSize, const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced);
// kz_text_gpu.hip: the TEXT inverse of blocks that are still in HBM (done[b] = 1 for the blocks it finished; the others: host stage)
size_t kz_text_gpu_scratch_per_block(int blockSize);
// kz_text_fwd_gpu.hip: the TEXT forward (TextCodec2) of blocks that sit in HBM (done[b] = 1 for the blocks it finished; the others: host stage)
size_t kz_text_fwd_gpu_scratch(int B, int blockSize, int maxLen);
// kz_utf_fwd_gpu.hip: UTFCodec.forward of the blocks TEXT declined with "dataType" UTF8 (done[b]: 1 applied, 2 declined by the reference's rules, 0 host stage)
int kz_utf_fwd_gpu(kz_ctx* ctx, kz_batch& bt, const std::vector<int32_t>& take, std::vector<int32_t>& done);
bool kz_text_fwd_gpu_applies(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType, int nBlocks);   // kz_api.hip
struct TextFwdJob;
TextFwdJob* kz_text_fwd_gpu_new();
void kz_text_fwd_gpu_free(TextFwdJob*);
int kz_text_fwd_gpu_classify(kz_ctx* ctx, kz_batch& bt, int blockSize, const std::vector<int32_t>& take, std::vector<int32_t>& keeps,
                             std::vector<int32_t>& declined, TextFwdJob& J);
int kz_text_fwd_gpu_launch(kz_ctx* ctx, kz_batch& bt, TextFwdJob& J);
int kz_text_fwd_gpu_finish(kz_ctx* ctx, kz_batch& bt, TextFwdJob& J, std::vector<int32_t>& done);
size_t kz_utf_gpu_scratch_per_block(int maxLen);
int kz_stage_utf_inverse_gpu(kz_ctx* ctx, kz_batch& bt, int dstCap, const std::vector<int32_t>& take, std::vector<int32_t>& done);
int kz_stage_text_inverse_gpu(kz_ctx* ctx, kz_batch& bt, int blockSize, int dstCap, bool variant1, const std::vector<int32_t>& take, std::vector<int32_t>& done, int form);
// run fn(i) for i in [0, n) on host threads (blocks are independent)
void kz_parallel_for(int n, int maxThreads, void (*fn)(int, void*), void* arg);   // kz_host.hip: persistent pool
int kz_usable_cpus();
struct HostPre;
int kz_stage_reserve(kz_ctx* ctx, kz_ctx::Stage& s, size_t need, bool pinned);   // grow-only staging buffer (kz_host.hip)
int32_t kz_encode_blocks_pre(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType, int32_t blockSize, const uint8_t* in, int64_t inStride,
                             const int32_t* lengths, int32_t nBlocks, uint8_t* out, int64_t outStride, kz_block_result* results, int32_t memKind, const HostPre* pre);
// the host stages (TEXT / UTF) of nBlocks blocks in host memory, ahead of kz_encode_blocks_pre (null when the chain has none or they
// cannot be run ahead); kz_host_pre_free releases it
HostPre* kz_host_prestage(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType, int32_t blockSize, const uint8_t* hsrc, int64_t hstride,
                          const int32_t* lengths, int32_t nBlocks, int slotId);   // slotId 0/1: the outputs go to the context's pinned hsOut[slotId]
void kz_host_pre_free(HostPre* p);

// timing helpers
void kz_stage_begin(kz_ctx*, hipEvent_t* e0);
void kz_stage_end(kz_ctx*, hipEvent_t e0, int stageId, int64_t algBytes);
