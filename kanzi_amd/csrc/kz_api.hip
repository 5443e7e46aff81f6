// kz_api.hip -- C-ABI entry points (include/kanzi_hip.h), the per-batch pipeline driver that plays
// the role of K/transform/Sequence.java + the codec span of EncodingTask.encodeBlock /
// DecodingTask.decodeBlock, the block-header kernels, and the host-side .knz stream framing.
#include <dirent.h>
#include "kz_device.h"
#include "kz_internal.h"
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <memory>
#include <thread>
#include <chrono>
#include <sched.h>

typedef uint32_t u32;
typedef uint8_t u8;
typedef unsigned long long u64;

// =================================================================================================
// context / arena
extern "C" int32_t kz_abi_version(void) { return KZ_ABI_VERSION; }

// The decoder's wide schedule keeps four streams busy side by side; HIP's default is four hardware queues for ALL streams of
// the process.  The library does NOT touch the process environment: the application exports GPU_MAX_HW_QUEUES=8 before its
// first HIP call (bench.py, tests/conftest.py and the Python binding do; a JVM is started with it in its environment,
// INTEGRATION.md 3) or gets the measured three-stream fallback (overlap_streams).
extern "C" kz_ctx* kz_ctx_create(int32_t deviceId) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return nullptr;   // fail loudly: no CPU fallback
  if (deviceId < 0 || deviceId >= ndev) return nullptr;
  if (hipSetDevice(deviceId) != hipSuccess) return nullptr;
  kz_ctx* ctx = new kz_ctx();
  ctx->device = deviceId;
  { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, deviceId) == hipSuccess && cus > 0) ctx->numCUs = cus; }
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return nullptr; }
  if (hipHostMalloc((void**)&ctx->hpin, 1 << 20, hipHostMallocDefault) != hipSuccess) { hipStreamDestroy(ctx->stream); delete ctx; return nullptr; }
  ctx->hpinInts = (1 << 20) / 4;
  kz_switches_read(ctx->sw);
  {
    const int hw = (int)std::thread::hardware_concurrency();
    ctx->blockingWaits = ctx->sw.blockingWaits >= 0 ? ctx->sw.blockingWaits : ((hw > 0 && kz_usable_cpus() < hw) ? 1 : 0);
  }
  return ctx;
}
// every environment switch of the library, in one place (kz_internal.h: kz_switches)
void kz_switches_read(kz_switches& s) {
  s = kz_switches();
  auto num = [](const char* name, int unset) { const char* e = getenv(name); return e ? atoi(e) : unset; };
  auto flag = [](const char* name) { return getenv(name) != nullptr ? 1 : 0; };
  auto digit = [](const char* name, int lo, int hi, int unset) { const char* e = getenv(name); return (e && e[0] >= '0' + lo && e[0] <= '0' + hi) ? e[0] - '0' : unset; };
  { const char* e = getenv("KZ_BLOCKING_WAITS"); s.blockingWaits = e ? (atoi(e) ? 1 : 0) : -1; }
  s.textGpu = digit("KZ_TEXT_GPU", 0, 3, -1);
  s.textGpuMin = num("KZ_TEXT_GPU_MIN", 512);
  s.textFwdGpu = digit("KZ_TEXT_FWD_GPU", 0, 1, -1);
  s.textFwdGpuMin = num("KZ_TEXT_FWD_GPU_MIN", 256);
  s.utfGpu = digit("KZ_UTF_GPU", 0, 0, 1);
  s.utfFwdGpu = digit("KZ_UTF_FWD_GPU", 0, 1, -1);
  s.textGpuTrace = flag("KZ_TEXT_GPU_TRACE");
  s.fuseMinBlocks = num("KZ_FUSE_MIN_BLOCKS", 32);
  s.overlapClasses = num("KZ_OVERLAP_CLASSES", 3);
  { const char* e = getenv("KZ_WIDE_QUEUES"); s.wideQueues = e ? (atoi(e) ? 1 : 0) : -1; }
  s.noSFirst = flag("KZ_NO_SFIRST");
  s.traceSched = flag("KZ_TRACE_SCHED");
  s.tracePipe = flag("KZ_TRACE_PIPE");
  s.hostChunk = num("KZ_HOST_CHUNK", 0);
  { const int v = num("KZ_HOST_CHUNK_DEC", 512); s.hostChunkDec = v < 8 ? 8 : v; }
  { const char* e = getenv("KZ_HOST_INV_STAGED"); s.hostInvStaged = e ? atoi(e) : -1; }
  s.streamChunk = num("KZ_STREAM_CHUNK", 0);
  s.streamSerial = flag("KZ_STREAM_SERIAL");
  s.bwtTrie = num("KZ_BWT_TRIE", -1); s.bwtTrieWin = num("KZ_BWT_TRIEWIN", -1); s.bwtBuckets = num("KZ_BWT_BUCKETS", -1);
  s.bwtDmax = num("KZ_BWT_DMAX", -1); s.bwtRetire = num("KZ_BWT_RETIRE", -1); s.bwtLazyRank = num("KZ_BWT_LAZYRANK", -1);
  s.bwtTrace = flag("KZ_BWT_TRACE");
  s.bwtTestTrieOverflow = digit("KZ_BWT_TEST_TRIE_OVERFLOW", 0, 9, -1);
  { const char* f = getenv("KZ_FPAQ_FORCE"); s.fpaqForce = f ? ((f[0] == 'w' || f[0] == 'W' || f[0] == '1') ? 1 : ((f[0] == 'l' || f[0] == 'L' || f[0] == '2') ? 2 : 0)) : 0; }
  s.sbrtForm = num("KZ_SBRT_FORM", -1);
}
extern "C" void kz_ctx_reload_switches(kz_ctx* ctx) {
  if (!ctx) return;
  kz_switches_read(ctx->sw);
  if (ctx->sw.blockingWaits >= 0) ctx->blockingWaits = ctx->sw.blockingWaits;
}
extern "C" void kz_ctx_destroy(kz_ctx* ctx) {
  if (!ctx) return;
  if (ctx->worker.joinable()) {                                     // queued batches finish first
    { std::lock_guard<std::mutex> g(ctx->qmu); ctx->stopWorker = true; }
    ctx->qcv.notify_all();
    ctx->worker.join();
  }
  hipSetDevice(ctx->device);
  if (ctx->arena) hipFree(ctx->arena);
  for (int i = 0; i < 5; i++) { if (ctx->side[i]) hipStreamDestroy(ctx->side[i]); if (ctx->evJoin[i]) hipEventDestroy(ctx->evJoin[i]); }
  if (ctx->hpin) hipHostFree(ctx->hpin);
  for (int i = 0; i < 2; i++) {
    if (ctx->pinIn[i].p) hipHostFree(ctx->pinIn[i].p);
    if (ctx->pinOut[i].p) hipHostFree(ctx->pinOut[i].p);
    if (ctx->hsIn[i].p) hipHostFree(ctx->hsIn[i].p);
    if (ctx->hsOut[i].p) hipHostFree(ctx->hsOut[i].p);
    if (ctx->tfOut[i].p) hipHostFree(ctx->tfOut[i].p);
    if (ctx->hiAux[i].p) hipFree(ctx->hiAux[i].p);
    if (ctx->hiStream[i]) hipStreamDestroy(ctx->hiStream[i]);
    if (ctx->devIn[i].p) hipFree(ctx->devIn[i].p);
    if (ctx->devOut[i].p) hipFree(ctx->devOut[i].p);
  }
  if (ctx->tfIn.p) hipHostFree(ctx->tfIn.p);
  if (ctx->tfCopy) hipStreamDestroy(ctx->tfCopy);
  if (ctx->evBlock) hipEventDestroy(ctx->evBlock);
  if (ctx->copyUp) hipStreamDestroy(ctx->copyUp);
  if (ctx->copyDown) hipStreamDestroy(ctx->copyDown);
  if (ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
}
extern "C" int32_t kz_ctx_set_checksum(kz_ctx* ctx, int32_t bits) {
  if (!ctx || (bits != 0 && bits != 32 && bits != 64)) return -KZ_ERR_INVALID_PARAM;
  ctx->checksum = bits == 32 ? 1 : (bits == 64 ? 2 : 0);
  return 0;
}
// ctx map key "skipBlocks" (the reference CLI's --skip): blocks that look incompressible are stored as copy blocks
extern "C" int32_t kz_ctx_set_skip_blocks(kz_ctx* ctx, int32_t on) {
  if (!ctx) return -KZ_ERR_INVALID_PARAM;
  ctx->skipBlocks = on ? 1 : 0;
  return 0;
}
// ctx map key "dataType" (Global.DataType): what a transform instance built with this context would find / leave there
extern "C" int32_t kz_ctx_set_data_type(kz_ctx* ctx, int32_t dataType) {
  if (!ctx || dataType < KZ_DT_UNDEFINED || dataType > KZ_DT_UTF8) return -KZ_ERR_INVALID_PARAM;
  ctx->dataType = dataType;
  return 0;
}
// ctx map keys "blockSize" and "entropy" as TEXT reads them (TextCodec.java:561-575,1068-1081; TransformFactory.java:275-286)
extern "C" int32_t kz_ctx_set_block_size(kz_ctx* ctx, int32_t blockSize) {
  if (!ctx || blockSize < 1024 || blockSize > (1 << 30) || (blockSize & 15)) return -KZ_ERR_BLOCK_SIZE;   // CompressedOutputStream.java:165-174
  ctx->blockSize = blockSize;
  ctx->blockSizeSet = true;
  return 0;
}
extern "C" int32_t kz_ctx_set_entropy(kz_ctx* ctx, uint32_t entropyType) {
  // TPAQX (9) is refused: the reference gives TEXT one more hash bit under it (TextCodec.java:561-575 extraPerf), which the host
  // stage does not model, and no entropy coder in scope is TPAQX
  if (!ctx || entropyType >= 9 || entropyType == 3) return -KZ_ERR_INVALID_CODEC;
  ctx->entropy = (int)entropyType;
  return 0;
}
// every kz_ctx_set_* value back to what a fresh context has (a caller that leaves through an error path need not remember which
// entries of its "map" it had set)
extern "C" int32_t kz_ctx_reset(kz_ctx* ctx) {
  if (!ctx) return -KZ_ERR_INVALID_PARAM;
  ctx->checksum = 0; ctx->skipBlocks = 0; ctx->dataType = KZ_DT_UNDEFINED;
  ctx->blockSize = 4 * 1024 * 1024; ctx->blockSizeSet = false; ctx->entropy = KZ_E_NONE;
  return 0;
}
extern "C" int32_t kz_ctx_get_data_type(kz_ctx* ctx) { return ctx ? ctx->dataType : -KZ_ERR_INVALID_PARAM; }
// N ranks on one host (one process per GPU): keep the process's host threads (TEXT / UTF stages, bit assembly, staging copies)
// on the CPUs next to its GPU.  Reads /sys/bus/pci/devices/<bdf>/local_cpulist; returns the number of CPUs pinned to, 0 when
// the topology is not visible (nothing changed), <0 on error.  Threads created afterwards inherit the mask.
extern "C" int32_t kz_pin_to_device_numa(int32_t deviceId) {
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), deviceId) != hipSuccess) return -KZ_ERR_DEVICE;
  for (char* p = bdf; *p; p++) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');
  char path[160];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bdf);
  FILE* f = fopen(path, "r");
  if (!f) return 0;
  char line[4096] = {0};
  const bool got = fgets(line, sizeof(line), f) != nullptr;
  fclose(f);
  if (!got) return 0;
  cpu_set_t set;
  CPU_ZERO(&set);
  int count = 0;
  for (char* p = line; *p;) {                                       // "0-31,64-95"
    char* e = nullptr;
    const long a = strtol(p, &e, 10);
    if (e == p) break;
    long b = a;
    p = e;
    if (*p == '-') { b = strtol(p + 1, &e, 10); p = e; }
    for (long c = a; c <= b && c < CPU_SETSIZE; c++) { CPU_SET((int)c, &set); count++; }
    if (*p == ',') p++;
  }
  if (count == 0) return 0;
  // every thread the process has at this moment (sched_setaffinity(0, ..) alone would pin the caller and the threads it creates
  // later, leaving a context's job worker or an already running host pool where they were: ADVICE r3); threads created afterwards
  // inherit their creator's mask
  int pinned = 0;
  if (DIR* dir = opendir("/proc/self/task")) {
    while (struct dirent* de = readdir(dir)) {
      const long tid = strtol(de->d_name, nullptr, 10);
      if (tid > 0 && sched_setaffinity((pid_t)tid, sizeof(set), &set) == 0) pinned++;
    }
    closedir(dir);
  }
  if (pinned == 0 && sched_setaffinity(0, sizeof(set), &set) != 0) return 0;
  return count;
}
extern "C" const char* kz_last_error(kz_ctx* ctx) { return ctx ? ctx->err : "null context"; }
extern "C" void* kz_ctx_stream(kz_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

// pinned read-back buffer: the batched calls read up to 9 int arrays of B entries back per stage (run_stage: 2B,
// kz_decode_blocks: 4B + 5B); grown here, never indexed beyond hpinInts
int kz_hpin_reserve(kz_ctx* ctx, size_t ints) {
  if (ints <= ctx->hpinInts) return 0;
  KZ_HIP(kz_stream_sync(ctx, ctx->stream));
  if (ctx->hpin) { hipHostFree(ctx->hpin); ctx->hpin = nullptr; ctx->hpinInts = 0; }
  ints = kz_align(ints + (ints >> 2), 1 << 16);
  KZ_HIP(hipHostMalloc((void**)&ctx->hpin, ints * 4, hipHostMallocDefault));
  ctx->hpinInts = ints;
  return 0;
}

// host threads of the TEXT / UTF stages.  The reference caps its jobs at 64 (BlockCompressor.java:199-203) on hosts of a few
// dozen cores; a GPU box brings hundreds, and these stages are what the GPU waits for in the level-exact chains: up to 128
// (every thread keeps a dictionary, stage buffers and a small alias map: a few MiB)
#define KZ_HOST_STAGE_THREADS 128
int kz_arena_reserve(kz_ctx* ctx, size_t total) {
  ctx->arenaTop = 0;
  if (total <= ctx->arenaCap) return 0;
  if (ctx->arena) { hipStreamSynchronize(ctx->stream); hipFree(ctx->arena); ctx->arena = nullptr; ctx->arenaCap = 0; }
  total = kz_align(total + (total >> 3), 1 << 20);
  KZ_HIP(hipMalloc((void**)&ctx->arena, total));
  ctx->arenaCap = total;
  return 0;
}
void* kz_arena_alloc(kz_ctx* ctx, size_t bytes) {
  size_t top = kz_align(ctx->arenaTop, 256);
  if (top + bytes > ctx->arenaCap) return nullptr;
  ctx->arenaTop = top + bytes;
  return ctx->arena + top;
}

// ---- timing --------------------------------------------------------------------------------------
extern "C" void kz_set_timing(kz_ctx* ctx, int32_t enable) { ctx->timing = enable != 0; }
extern "C" int32_t kz_get_stage_count(kz_ctx*) { return KZ_MAX_STAGES; }
extern "C" float kz_get_stage_ms(kz_ctx* ctx, int32_t s) { return (s >= 0 && s < KZ_MAX_STAGES) ? ctx->stageMs[s] : 0.f; }
extern "C" int64_t kz_get_stage_alg_bytes(kz_ctx* ctx, int32_t s) { return (s >= 0 && s < KZ_MAX_STAGES) ? ctx->stageAlgBytes[s] : 0; }
extern "C" void kz_reset_timing(kz_ctx* ctx) { for (int i = 0; i < KZ_MAX_STAGES; i++) { ctx->stageMs[i] = 0; ctx->stageAlgBytes[i] = 0; } }

hipEvent_t kz_ev(kz_ctx* ctx) {
  if (!ctx->evPool.empty()) { hipEvent_t e = ctx->evPool.back(); ctx->evPool.pop_back(); return e; }
  hipEvent_t e; hipEventCreate(&e); return e;
}
void kz_ktimer_flush(kz_ctx* ctx) {
  for (auto& p : ctx->pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) { ctx->kMs[p.id] += ms; ctx->kLaunches[p.id]++; if (ms > ctx->kMaxMs[p.id]) ctx->kMaxMs[p.id] = ms; }
    ctx->evPool.push_back(p.e0); ctx->evPool.push_back(p.e1);
  }
  ctx->pending.clear();
}
extern "C" void kz_set_kernel_timing(kz_ctx* ctx, int32_t enable) { ctx->ktiming = enable != 0; }
extern "C" int32_t kz_get_kernel_count(void) { return KID_COUNT; }
extern "C" const char* kz_get_kernel_name(int32_t id) { static const char* n[] = KZ_KERNEL_NAMES; return (id >= 0 && id < KID_COUNT) ? n[id] : ""; }
extern "C" double kz_get_kernel_ms(kz_ctx* ctx, int32_t id) { return (id >= 0 && id < KID_COUNT) ? ctx->kMs[id] : 0.0; }
extern "C" double kz_get_kernel_max_ms(kz_ctx* ctx, int32_t id) { return (id >= 0 && id < KID_COUNT) ? ctx->kMaxMs[id] : 0.0; }
extern "C" int64_t kz_get_kernel_launches(kz_ctx* ctx, int32_t id) { return (id >= 0 && id < KID_COUNT) ? ctx->kLaunches[id] : 0; }
extern "C" void kz_reset_kernel_timing(kz_ctx* ctx) { for (int i = 0; i < KID_COUNT; i++) { ctx->kMs[i] = 0; ctx->kMaxMs[i] = 0; ctx->kLaunches[i] = 0; } }

void kz_stage_begin(kz_ctx* ctx, hipEvent_t* e0) {
  *e0 = nullptr;
  if (!ctx->timing) return;
  hipEventCreate(e0);
  hipEventRecord(*e0, ctx->stream);
}
void kz_stage_end(kz_ctx* ctx, hipEvent_t e0, int stageId, int64_t algBytes) {
  if (!ctx->timing || !e0) return;
  hipEvent_t e1; hipEventCreate(&e1);
  hipEventRecord(e1, ctx->stream);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  ctx->stageMs[stageId] += ms; ctx->stageAlgBytes[stageId] += algBytes;
  hipEventDestroy(e0); hipEventDestroy(e1);
}

// =================================================================================================
// ids / sizes
extern "C" int32_t kz_transform_max_encoded_len(uint32_t type, int32_t n) {
  switch (type) {
    case KZ_T_BWT: return n + 33;                                    // BWTBlockCodec.java:40,222
    case KZ_T_SRT: return n + 1024;                                  // SRT.java:30,365
    case KZ_T_LZ: case KZ_T_LZX: return ((n <= 1024) ? n + 16 : n + (n / 64)) + 2;   // LZCodec.java:961-964
    case KZ_T_MM: return n + std::max(64, n >> 4);                   // FSDCodec.java:320-323
    case KZ_T_PACK: case KZ_T_DNA: return n + 1024;                  // AliasCodec.java:472-475
    case KZ_T_UTF: return n + 8192;                                  // UTFCodec.java:308-310
    default: return n;                                               // ZRLT.java:243, SBRT.java:224
  }
}
extern "C" uint64_t kz_transform_type(const int32_t* types, int32_t nb) {   // TransformFactory.java:132-158
  uint64_t t = 0;
  for (int i = 0; i < 8; i++) t = (t << 6) | (uint64_t)((i < nb) ? (types[i] & 0x3F) : 0);
  return t;
}
static int split_types(uint64_t tt, int* types) {                    // TransformFactory.java:240-266
  int nbtr = 0;
  for (int i = 0; i < 8; i++) if (((tt >> (42 - 6 * i)) & 0x3F) != KZ_T_NONE) nbtr++;
  if (nbtr == 0) nbtr = 1;
  int k = 0;
  for (int i = 0; i < nbtr; i++) { int t = (int)((tt >> (42 - 6 * i)) & 0x3F); if (t != KZ_T_NONE || i == 0) types[k++] = t; }
  return k;
}
static bool transform_supported(int t) { return t == KZ_T_NONE || t == KZ_T_TEXT || t == KZ_T_UTF || t == KZ_T_BWT || t == KZ_T_RANK || t == KZ_T_MTFT || t == KZ_T_ZRLT || t == KZ_T_SRT || t == KZ_T_LZ || t == KZ_T_LZX || t == KZ_T_MM || t == KZ_T_PACK || t == KZ_T_DNA; }
static bool entropy_supported(int e) { return e == KZ_E_NONE || e == KZ_E_ANS0 || e == KZ_E_HUFFMAN || e == KZ_E_FPAQ; }
static int seq_max_len(const int* types, int nb, int n) {             // Sequence.java:215-226
  int req = n;
  for (int i = 0; i < nb; i++) req = std::max(req, kz_transform_max_encoded_len((uint32_t)types[i], req));
  return req;
}
extern "C" int64_t kz_max_block_stream_bytes(int32_t n) { return (int64_t)kz_align((size_t)n + (size_t)(n >> 3) + 1024, 256); }

// =================================================================================================
// small device helpers
__global__ void k_mask_len(const int32_t* __restrict__ len, const int32_t* __restrict__ mask, int32_t* __restrict__ out, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) out[b] = mask[b] ? len[b] : 0;
}
// pass-through for blocks whose stage did not run (mask==0) or declined (flag==0)
__global__ void k_passthrough(const u8* __restrict__ src, u8* __restrict__ dst, int64_t stride, const int32_t* __restrict__ lenOld,
                              int32_t* __restrict__ lenNew, const int32_t* __restrict__ mask, const int32_t* __restrict__ flag,
                              int32_t* __restrict__ applied, const int32_t* __restrict__ part) {
  const int b = blockIdx.y;
  if (part && !part[b]) {                                           // not this pass's business: the slot is left alone
    if (blockIdx.x == 0 && threadIdx.x == 0) { applied[b] = 0; lenNew[b] = lenOld[b]; }
    return;
  }
  const bool ran = mask[b] != 0 && flag[b] > 0;                     // flag < 0: the reference's transform would have thrown
  if (blockIdx.x == 0 && threadIdx.x == 0) { applied[b] = ran ? 1 : ((mask[b] != 0 && flag[b] < 0) ? -1 : 0); if (!ran) lenNew[b] = lenOld[b]; }
  if (ran) return;
  const int n = lenOld[b];
  const u8* s = src + (int64_t)b * stride;
  u8* d = dst + (int64_t)b * stride;
  for (int i = (blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += gridDim.x * blockDim.x * 16) {
    if (i + 16 <= n) *(uint4*)(d + i) = *(const uint4*)(s + i);
    else for (int k = i; k < n; k++) d[k] = s[k];
  }
}
__global__ void k_copy_bytes(const u8* __restrict__ src, int64_t sstride, u8* __restrict__ dst, int64_t dstride,
                             const int32_t* __restrict__ len, const int32_t* __restrict__ dstOff, const int32_t* __restrict__ cond) {
  const int b = blockIdx.y;
  if (cond && !cond[b]) return;
  const int n = len[b];
  const u8* s = src + (int64_t)b * sstride;
  u8* d = dst + (int64_t)b * dstride + (dstOff ? dstOff[b] : 0);
  const int t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  if (((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0) {            // both 16-byte aligned (uniform per block): 16 bytes per lane
    const int n16 = n >> 4;
    for (int i = t; i < n16; i += nt) ((uint4*)d)[i] = ((const uint4*)s)[i];
    for (int i = (n16 << 4) + t; i < n; i += nt) d[i] = s[i];
  } else {
    for (int i = t; i < n; i += nt) d[i] = s[i];
  }
}

__device__ __forceinline__ u32 kz_mix32(u32 c, u32 h, u32 v) {        // CompressedOutputStream.java:89-93
  c ^= h * ~v;
  c = (c << 13) | (c >> 19);
  return c * 5u + 0x52DCE729u;
}
__device__ __forceinline__ u8 kz_block_cksum(u32 mode, u32 hsf, u32 postLen, u64 written) {   // :977-985
  const u32 HASH = 0x1E35A7BDu;
  u32 c = HASH * 0x01030507u;
  c = kz_mix32(c, HASH, mode);
  c = kz_mix32(c, HASH, hsf);
  c = kz_mix32(c, HASH, postLen);
  c = kz_mix32(c, HASH, (u32)(written >> 32));
  c = kz_mix32(c, HASH, (u32)written);
  c = (c >> 23) ^ (c >> 3);
  return (u8)c;
}

struct FrameEnc {
  int32_t* postLen;      // = batch d_len after the chain
  int32_t* skipFlags;    // [B]
  int32_t* hdrBytes;     // [B]
  int32_t* isCopy;       // [B] small block (<= 15 bytes): raw copy block
  int32_t* fallback;     // [B]
  int64_t* bits;         // [B] entropy payload bits -> final W
  u64* hash;             // [B] block checksum (XXHash32/64 of the original block)
  int chk;               // 0 none, 1 = 32 bit, 2 = 64 bit
  int nbFunctions;
};

// before entropy: header size per block (CompressedOutputStream.java:825-826, :861-896)
__global__ void k_frame_prepare(FrameEnc F, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int postLen = F.postLen[b];
  const int dataSize = (postLen < 256) ? 1 : (kz_ilog2((u32)postLen) >> 3) + 1;
  const bool two = !F.isCopy[b] && F.nbFunctions > 4;
  F.hdrBytes[b] = 1 + (two ? 1 : 0) + dataSize + 1 + (F.chk == 1 ? 4 : (F.chk == 2 ? 8 : 0));
}
// after entropy: decide the raw "transformed copy" fallback (:926-973)
__global__ void k_frame_decide(FrameEnc F, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int64_t written = 8LL * F.hdrBytes[b] + F.bits[b];
  const int64_t entropyBytes = (written + 7) >> 3;
  F.fallback[b] = (!F.isCopy[b] && (int64_t)F.postLen[b] < entropyBytes) ? 1 : 0;
}
__global__ void k_frame_header(FrameEnc F, u8* __restrict__ out, int64_t outStride, kz_block_result* __restrict__ res, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  u8* o = out + (int64_t)b * outStride;
  const int postLen = F.postLen[b];
  const u32 skipFlags = (u32)F.skipFlags[b] & 0xFF;
  const int nb = F.nbFunctions;
  const int dataSize = (postLen < 256) ? 1 : (kz_ilog2((u32)postLen) >> 3) + 1;
  u32 mode = F.isCopy[b] ? 0x80u : 0u;
  mode |= (u32)(((dataSize - 1) & 3) << 5);
  u32 hsf = skipFlags;
  int idx = 0;
  int64_t written;
  if (postLen == 0) { res[b].bits = 0; res[b].length = 0; res[b].status = 0; res[b].skipFlags = 0xFF; res[b].mode = 0; return; }
  if (F.fallback[b]) {
    const u32 copyMode = (mode | (nb <= 4 ? (skipFlags >> 4) : 0x10u)) | 0x80u | 0x10u;     // :933-934
    o[idx++] = (u8)copyMode;
    if (nb > 4) { o[idx++] = (u8)skipFlags; hsf = skipFlags; } else hsf = ((copyMode << 4) | 0x0F) & 0xFF;
    mode = copyMode;
    written = 8LL * (F.hdrBytes[b] + postLen);
  } else {
    if ((mode & 0x80) || nb <= 4) {
      mode |= (skipFlags >> 4);
      hsf = (mode & 0x80) ? 0u : (((mode << 4) | 0x0F) & 0xFF);
      o[idx++] = (u8)mode;
    } else {
      mode |= 0x10;
      o[idx++] = (u8)mode; o[idx++] = (u8)skipFlags;
    }
    written = 8LL * F.hdrBytes[b] + F.bits[b];
  }
  for (int k = dataSize - 1; k >= 0; k--) o[idx++] = (u8)((u32)postLen >> (8 * k));
  o[idx++] = kz_block_cksum(mode & 0xFF, hsf, (u32)postLen, (u64)written);
  if (F.chk) {                                               // CompressedOutputStream.java:887-891
    const int nby = (F.chk == 1) ? 4 : 8;
    const u64 hv = F.hash[b];
    for (int k = nby - 1; k >= 0; k--) o[idx++] = (u8)(hv >> (8 * k));
  }
  res[b].bits = written; res[b].length = postLen; res[b].status = 0;
  res[b].skipFlags = (u8)skipFlags; res[b].mode = (u8)mode;
}

// =================================================================================================
// batch plumbing
struct Pipe {
  kz_batch bt;
  int32_t* d_mask = nullptr;      // [B] stage applies to this block
  int32_t* d_applied = nullptr;   // [B]
  int32_t* d_lenSave = nullptr;   // [B]
  int32_t* d_one = nullptr;       // [B] all ones
};

static int sync_lengths(kz_ctx* ctx, kz_batch& bt) {
  KZ_HIP(hipMemcpyAsync(ctx->hpin, bt.d_len, (size_t)bt.B * 4, hipMemcpyDeviceToHost, ctx->stream));
  KZ_HIP(kz_stream_sync(ctx, ctx->stream));
  for (int b = 0; b < bt.B; b++) bt.h_len[b] = ctx->hpin[b];
  return 0;
}

// run one stage on the blocks selected by h_mask; others (and blocks where the transform declines)
// pass through unchanged.  On return h_applied[b] tells which blocks the transform was applied to.
template <typename F>
static int run_stage(kz_ctx* ctx, Pipe& P, const std::vector<int32_t>& h_mask, std::vector<int32_t>& h_applied, F stage, const int32_t* d_part = nullptr) {
  kz_batch& bt = P.bt;
  const int B = bt.B;
  hipStream_t st = ctx->stream;
  KZ_HIP(hipMemcpyAsync(P.d_mask, h_mask.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
  // save true lengths, run on masked lengths
  KZ_HIP(hipMemcpyAsync(P.d_lenSave, bt.d_len, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
  KZ_LAUNCH(ctx, KID_MASK_LEN, k_mask_len, dim3((B + 255) / 256), dim3(256), P.d_lenSave, P.d_mask, bt.d_len, B);
  std::vector<int32_t> saved = bt.h_len;
  for (int b = 0; b < B; b++) if (!h_mask[b]) bt.h_len[b] = 0;
  const u8* srcBefore = bt.buf[bt.cur];
  int rc = stage(bt);
  if (rc) return rc;
  u8* dstAfter = bt.buf[bt.cur];
  KZ_LAUNCH(ctx, KID_PASSTHROUGH, k_passthrough, dim3(64, B), dim3(256), srcBefore, dstAfter, bt.stride, P.d_lenSave, bt.d_len, P.d_mask, bt.d_flag, P.d_applied, d_part);
  KZ_HIP(hipMemcpyAsync(ctx->hpin + B, P.d_applied, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  rc = sync_lengths(ctx, bt);
  if (rc) return rc;
  h_applied.resize(B);
  for (int b = 0; b < B; b++) h_applied[b] = ctx->hpin[B + b];
  (void)saved;
  return 0;
}

static size_t kz_arena_budget_probe() {
  const char* e = getenv("KZ_ARENA_BUDGET_GB");
  double gb = e ? atof(e) : 96.0;
  size_t freeB = 0, totalB = 0;
  if (hipMemGetInfo(&freeB, &totalB) == hipSuccess && freeB > 0) gb = std::min(gb, (double)freeB / (1 << 30) * 0.6);
  if (gb < 1.0) gb = 1.0;
  return (size_t)(gb * (double)(1 << 30));
}
static size_t kz_arena_budget() {
  static const size_t budget = kz_arena_budget_probe();      // initialised once, also when contexts live on several threads
  return budget;
}
// stage scratch is bump-allocated after the ping-pong buffers; each stage allocates in turn, so the
// arena must hold the SUM over the stages the chain actually runs
struct ChainSpec { int types[8]; int nb; int entropy; };
// The forward BWT (suffix sort) needs ~43 B of scratch per input byte, an order of magnitude more than any
// other stage.  It therefore sorts the batch in groups that fit half the arena budget and reuses the scratch,
// while every other stage (and the serial-per-block entropy coders in particular) sees the whole batch.
static int bwt_group_blocks(int B, int maxLen) {
  const size_t per = kz_bwt_forward_scratch(1, maxLen);
  const int g = (int)std::max<size_t>(1, (kz_arena_budget() / 2) / per);
  if (g >= B) return B;
  const int groups = (B + g - 1) / g;                                // equal groups: 2 048 blocks at 341 per group used to end with a
  return (B + groups - 1) / groups;                                  // seventh group of TWO blocks paying every round's launches
}
static size_t pipeline_scratch(int B, int maxLen, bool decode, const ChainSpec& C) {
  size_t s = 65536;
  for (int i = 0; i < C.nb; i++) {
    switch (C.types[i]) {
      case KZ_T_BWT: s += decode ? kz_bwt_inverse_scratch(B, maxLen) : kz_bwt_forward_scratch(bwt_group_blocks(B, maxLen), maxLen); break;
      case KZ_T_RANK: case KZ_T_MTFT: s += kz_sbrt_scratch(B, maxLen); break;
      case KZ_T_ZRLT: s += kz_zrlt_scratch(B, maxLen); break;
      case KZ_T_SRT: s += decode ? 4096 : kz_srt_scratch(B, maxLen); break;
      case KZ_T_LZ: case KZ_T_LZX: s += decode ? 4096 : kz_lz_scratch(B, maxLen); break;
      case KZ_T_MM: s += kz_mm_scratch(B, maxLen); break;
      case KZ_T_PACK: case KZ_T_DNA: s += kz_alias_scratch(B, maxLen, decode); break;
      default: break;
    }
  }
  if (C.entropy == KZ_E_ANS0 || C.entropy == KZ_E_HUFFMAN)
    s += decode ? (size_t)B * ((size_t)(maxLen / 16384 + 4) * 8 + 64) + 65536 : kz_ans_scratch(B, maxLen);
  else if (C.entropy == KZ_E_FPAQ)
    s += decode ? 4096 : kz_fpaq_scratch(B, maxLen);
  return s;
}

static int pipe_setup(kz_ctx* ctx, Pipe& P, int B, int maxLen, int64_t extraBytes, bool decode, const ChainSpec& C) {
  kz_batch& bt = P.bt;
  bt.B = B; bt.maxN = maxLen;
  bt.stride = (int64_t)kz_align((size_t)maxLen + 4096, 256);
  const size_t fixed = (size_t)bt.stride * B * 2 + (size_t)B * 4 * 16 + 65536 + (size_t)extraBytes;
  int rc = kz_arena_reserve(ctx, fixed + pipeline_scratch(B, maxLen, decode, C) + (1 << 20));
  if (rc) return rc;
  rc = kz_hpin_reserve(ctx, (size_t)B * 9 + 64);
  if (rc) return rc;
  bt.buf[0] = (u8*)kz_arena_alloc(ctx, (size_t)bt.stride * B);
  bt.buf[1] = (u8*)kz_arena_alloc(ctx, (size_t)bt.stride * B);
  bt.d_len = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  bt.d_len2 = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  bt.d_flag = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  bt.d_dtype = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  P.d_mask = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  P.d_applied = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  P.d_lenSave = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  bt.cur = 0;
  bt.h_len.assign(B, 0);
  if (!P.d_lenSave || !bt.d_dtype) { snprintf(ctx->err, sizeof(ctx->err), "pipe_setup: arena overflow"); return -KZ_ERR_DEVICE; }
  KZ_HIP(hipMemsetAsync(bt.d_dtype, 0, (size_t)B * 4, ctx->stream));      // Global.DataType.UNDEFINED
  return 0;
}

static int bwt_forward_grouped(kz_ctx* ctx, kz_batch& bt) {
  const int B = bt.B;
  const int G = bwt_group_blocks(B, bt.maxN);
  if (G >= B) return kz_stage_bwt_forward(ctx, bt);
  const size_t mark = ctx->arenaTop;
  for (int b0 = 0; b0 < B; b0 += G) {
    const int cnt = std::min(G, B - b0);
    kz_batch v;                                   // a view of blocks [b0, b0+cnt)
    v.B = cnt; v.maxN = bt.maxN; v.stride = bt.stride; v.cur = bt.cur;
    v.buf[0] = bt.buf[0] + (int64_t)b0 * bt.stride; v.buf[1] = bt.buf[1] + (int64_t)b0 * bt.stride;
    v.d_len = bt.d_len + b0; v.d_len2 = bt.d_len2 + b0; v.d_flag = bt.d_flag + b0;
    v.h_len.assign(bt.h_len.begin() + b0, bt.h_len.begin() + b0 + cnt);
    ctx->arenaTop = mark;                         // the groups run back to back on one stream: reuse the scratch
    const int rc = kz_stage_bwt_forward(ctx, v);
    if (rc) return rc;
  }
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}

static int run_transform_stage(kz_ctx* ctx, kz_batch& bt, int type, bool forward, int dstCap) {
  switch (type) {
    case KZ_T_BWT: return forward ? bwt_forward_grouped(ctx, bt) : kz_stage_bwt_inverse(ctx, bt);
    case KZ_T_RANK: return forward ? kz_stage_sbrt_forward(ctx, bt, 2) : kz_stage_sbrt_inverse(ctx, bt, 2);
    case KZ_T_MTFT: return forward ? kz_stage_sbrt_forward(ctx, bt, 1) : kz_stage_sbrt_inverse(ctx, bt, 1);
    case KZ_T_ZRLT: return forward ? kz_stage_zrlt_forward(ctx, bt) : kz_stage_zrlt_inverse(ctx, bt, dstCap);
    case KZ_T_SRT: return forward ? kz_stage_srt_forward(ctx, bt) : kz_stage_srt_inverse(ctx, bt);
    case KZ_T_LZ: return forward ? kz_stage_lz_forward(ctx, bt, 0) : kz_stage_lz_inverse(ctx, bt, 0, dstCap);
    case KZ_T_LZX: return forward ? kz_stage_lz_forward(ctx, bt, 1) : kz_stage_lz_inverse(ctx, bt, 1, dstCap);
    case KZ_T_MM: return forward ? kz_stage_mm_forward(ctx, bt) : kz_stage_mm_inverse(ctx, bt, dstCap);
    case KZ_T_PACK: return forward ? kz_stage_alias_forward(ctx, bt, 0) : kz_stage_alias_inverse(ctx, bt, dstCap);
    case KZ_T_DNA: return forward ? kz_stage_alias_forward(ctx, bt, 1) : kz_stage_alias_inverse(ctx, bt, dstCap);   // TransformFactory.java:341-343
    default: snprintf(ctx->err, sizeof(ctx->err), "transform %d has no HIP stage", type); return -KZ_ERR_INVALID_CODEC;
  }
}
static int stage_id(int type, bool forward) {
  switch (type) {
    case KZ_T_BWT: return forward ? KZ_STAGE_BWT_FWD : KZ_STAGE_BWT_INV;
    case KZ_T_RANK: case KZ_T_MTFT: return forward ? KZ_STAGE_SBRT_FWD : KZ_STAGE_SBRT_INV;
    case KZ_T_SRT: return forward ? KZ_STAGE_SRT_FWD : KZ_STAGE_SRT_INV;
    case KZ_T_LZ: case KZ_T_LZX: return forward ? KZ_STAGE_LZ_FWD : KZ_STAGE_LZ_INV;
    default: return forward ? KZ_STAGE_ZRLT_FWD : KZ_STAGE_ZRLT_INV;
  }
}

// =================================================================================================
// host (CPU) stages: TEXT and UTF lead the chains of the reference's levels 3, 5 and 6.  They are sequential dictionary coders
// (kz_text.hip); the batched calls run them per block on host threads, in front of the GPU stages when encoding and behind
// them when decoding, and carry their skip flags and the block's "dataType" entry through like Sequence does.
static int host_prefix(kz_ctx* ctx, const int* types, int nb) {          // number of leading host stages, or <0
  int hp = 0;
  while (hp < nb && kz_is_host_transform(types[hp])) hp++;
  for (int i = hp; i < nb; i++)
    if (kz_is_host_transform(types[i])) {
      snprintf(ctx->err, sizeof(ctx->err), "TEXT / UTF are built as host stages in front of the GPU stages only (as in the reference's levels)");
      return -KZ_ERR_INVALID_CODEC;
    }
  return hp;
}
// Result of the host stages (TEXT, UTF) of a set of blocks, computed host -> host so that it can run on a helper thread while the
// GPU codes the previous set: per block its length, skip flag bits and "dataType" entry, and -- when a stage applied -- the
// transformed bytes (slot b of `store`).
struct HostPre {
  std::vector<int32_t> outLen, skip, dtype;
  std::vector<uint8_t> changed;
  std::unique_ptr<uint8_t[]> own;                   // the slots' memory when the caller gave none
  uint8_t* store = nullptr;
  bool pinned = false;                              // store is pinned host memory (a kernel can read it: one gather instead of a copy per block)
  int64_t slot = 0;
  const uint8_t* data(int b) const { return store + (int64_t)b * slot; }
};
struct HostFwd {
  const int* types; int hp; int entropy; int cap; int blockSize;
  const uint8_t* hsrc; int64_t hstride;             // the blocks in host memory
  const int32_t* lengths; const int32_t* copy;      // copy: blocks the chain does not apply to (null: those of <= 15 bytes)
  HostPre* P;
  const uint8_t* const* ptrs = nullptr;             // a LIST of blocks instead (block k at ptrs[k]; hsrc / hstride unused)
  const int32_t* first = nullptr;                   // list form: the stages in front of first[k] are done with block k (they declined:
  const int32_t* dt0 = nullptr;                     //            data untouched, skip bits set) and left the "dataType" dt0[k]
};
// blocks that went through a TEXT / UTF stage on a HOST thread since the last reset (process-wide; bench.py reports them next to the
// stage's wall time, which includes the device forms' kernels): [0] forward, [1] inverse
static std::atomic<int64_t> g_hostStageBlocks[2];
extern "C" int64_t kz_host_stage_blocks(int32_t inverse, int32_t reset) {
  std::atomic<int64_t>& c = g_hostStageBlocks[inverse ? 1 : 0];
  return reset ? c.exchange(0) : c.load();
}
static void host_forward_block(int b, void* arg) {
  HostFwd& H = *(HostFwd*)arg;
  HostPre& P = *H.P;
  const int n = H.lengths[b];
  P.outLen[b] = n;
  P.dtype[b] = KZ_DT_UNDEFINED;
  P.skip[b] = 0xFF;
  P.changed[b] = 0;
  if (n == 0 || (H.copy ? H.copy[b] != 0 : n <= 15)) return;
  g_hostStageBlocks[0]++;
  static thread_local std::vector<uint8_t> bufA;
  if ((int)bufA.size() < H.cap + 64) bufA.resize((size_t)H.cap + 64);
  const uint8_t* const origin = H.ptrs ? H.ptrs[b] : H.hsrc + (int64_t)b * H.hstride;
  const uint8_t* cur = origin;
  const int i0 = H.first ? H.first[b] : 0;
  int dt = i0 > 0 ? H.dt0[b] : kz_host_block_data_type(cur, n, KZ_DT_UNDEFINED);   // CompressedOutputStream.java:795-804
  int len = n;
  uint8_t* mine = P.store + (int64_t)b * P.slot;
  uint8_t* out = mine;                                                  // ping-pong between the block's slot and a scratch buffer
  for (int i = i0; i < H.hp; i++) {
    int produced = 0;
    if (!kz_host_transform_forward(H.types[i], H.entropy, H.blockSize, &dt, cur, len, out, H.cap, &produced)) continue;   // declined: data untouched
    P.skip[b] &= ~(1 << (7 - i));
    cur = out; len = produced;
    out = (out == mine) ? bufA.data() : mine;
  }
  P.dtype[b] = dt;
  if (cur != origin) {                                                  // a stage applied
    if (cur != mine) memcpy(mine, cur, (size_t)len);
    P.changed[b] = 1;
    P.outLen[b] = len;
  }
}
// runs the chain's host stages over B blocks in host memory
static int64_t host_pre_slot(int cap) { return (int64_t)kz_align((size_t)cap + 64, 64); }
// store: B slots of host_pre_slot(cap) bytes for the stages' outputs (pinned staging of the pipelines), or null
static void host_prestage(const int* types, int hp, int entropy, int blockSize, int cap, const uint8_t* hsrc, int64_t hstride,
                          const int32_t* lengths, const int32_t* copy, int B, HostPre& P, uint8_t* store = nullptr) {
  P.outLen.assign(B, 0); P.skip.assign(B, 0xFF); P.dtype.assign(B, 0); P.changed.assign(B, 0);
  P.slot = host_pre_slot(cap);
  if (store) { P.store = store; P.pinned = true; }
  else { P.own.reset(new uint8_t[(size_t)P.slot * (size_t)B + 64]); P.store = P.own.get(); }   // uninitialised: only the bytes a stage writes are touched
  HostFwd H;
  H.types = types; H.hp = hp; H.entropy = entropy; H.cap = cap; H.blockSize = blockSize;
  H.hsrc = hsrc; H.hstride = hstride; H.lengths = lengths; H.copy = copy; H.P = &P;
  kz_parallel_for(B, KZ_HOST_STAGE_THREADS, host_forward_block, &H);
}
// the same over a list of blocks (block k at ptrs[k], lengths[k] bytes): the results are numbered like the list
// store: K slots of host_pre_slot(cap) bytes (pinned: the per-block copies back to HBM are then real DMA), or null
static void host_prestage_list(const int* types, int hp, int entropy, int blockSize, int cap, const std::vector<const uint8_t*>& ptrs,
                               const std::vector<int32_t>& lengths, HostPre& P, uint8_t* store = nullptr,
                               const std::vector<int32_t>* first = nullptr, const std::vector<int32_t>* dt0 = nullptr) {
  const int K = (int)ptrs.size();
  P.outLen.assign(K, 0); P.skip.assign(K, 0xFF); P.dtype.assign(K, 0); P.changed.assign(K, 0);
  P.slot = host_pre_slot(cap);
  if (store) P.store = store;                          // (P.pinned stays false: that flag selects the gather kernel, which indexes slots by block)
  else { P.own.reset(new uint8_t[(size_t)P.slot * (size_t)std::max(K, 1) + 64]); P.store = P.own.get(); }
  if (K == 0) return;
  std::vector<int32_t> none(K, 0);
  HostFwd H;
  H.types = types; H.hp = hp; H.entropy = entropy; H.cap = cap; H.blockSize = blockSize;
  H.hsrc = nullptr; H.hstride = 0; H.lengths = lengths.data(); H.copy = none.data(); H.P = &P; H.ptrs = ptrs.data();
  if (first && dt0) { H.first = first->data(); H.dt0 = dt0->data(); }
  kz_parallel_for(K, KZ_HOST_STAGE_THREADS, host_forward_block, &H);
}
HostPre* kz_host_prestage(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType, int32_t blockSize, const uint8_t* hsrc, int64_t hstride,
                          const int32_t* lengths, int32_t nBlocks, int slotId) {
  int types[8];
  const int nb = split_types(transformType, types);
  int hp = 0;
  while (hp < nb && kz_is_host_transform(types[hp])) hp++;
  if (hp == 0 || ctx->skipBlocks || nBlocks <= 0) return nullptr;   // ("skipBlocks": the copy decision comes from the device first)
  int maxN = 0;
  for (int b = 0; b < nBlocks; b++) maxN = std::max(maxN, lengths[b]);
  HostPre* P = new HostPre();
  const int cap = seq_max_len(types, nb, maxN);
  uint8_t* store = nullptr;
  if (slotId >= 0 && slotId < 2 && hipSetDevice(ctx->device) == hipSuccess &&
      kz_stage_reserve(ctx, ctx->hsOut[slotId], (size_t)host_pre_slot(cap) * (size_t)nBlocks + 64, true) == 0) store = ctx->hsOut[slotId].p;
  host_prestage(types, hp, (int)entropyType, blockSize, cap, hsrc, hstride, lengths, nullptr, nBlocks, *P, store);
  return P;
}
void kz_host_pre_free(HostPre* p) { delete p; }
struct HostInv {
  int device; const int* types; int hp; int blockSize; int cap;
  uint8_t* dbuf; int64_t dstride; int64_t slotCap; bool hostMem;        // the blocks' slots: in HBM (copied over and back) or in host memory
  int32_t* len; const int32_t* skip; int32_t* status; std::atomic<int> fail{0};
  uint8_t* pin = nullptr; int64_t pinSlot = 0;                          // staged form: the chunk's blocks were gathered into pinned slots
};
static void host_inverse_block(int b, void* arg) {
  HostInv& H = *(HostInv*)arg;
  int len = H.len[b];
  if (H.status[b] || len <= 0) return;
  bool any = false;
  for (int i = 0; i < H.hp; i++) any |= !(H.skip[b] & (1 << (7 - i)));
  if (!any) return;
  g_hostStageBlocks[1]++;
  static thread_local std::vector<uint8_t> bufA, bufB;
  const size_t need = (size_t)std::max(H.cap, len) + 64;
  if (bufA.size() < need) { bufA.resize(need); bufB.resize(need); }
  uint8_t* slot = H.pin ? H.pin + (int64_t)b * H.pinSlot : H.dbuf + (int64_t)b * H.dstride;
  const bool hostSlot = H.hostMem || H.pin != nullptr;
  uint8_t* cur = bufA.data();
  uint8_t* out = bufB.data();
  if (H.pin) { cur = slot; out = bufA.data(); }                         // read the pinned slot in place, first output to a private buffer
  else if (H.hostMem) memcpy(bufA.data(), slot, (size_t)len);
  else if (hipSetDevice(H.device) != hipSuccess || hipMemcpy(bufA.data(), slot, (size_t)len, hipMemcpyDeviceToHost) != hipSuccess) { H.fail = 1; return; }
  for (int i = H.hp - 1; i >= 0; i--) {                                 // Sequence.inverse: last applied first
    if (H.skip[b] & (1 << (7 - i))) continue;
    int produced = 0;
    if (!kz_host_transform_inverse(H.types[i], H.blockSize, cur, len, out, H.cap, &produced)) { H.status[b] = -KZ_ERR_PROCESS_BLOCK; H.len[b] = 0; return; }
    len = produced;
    if (cur == slot) { cur = out; out = bufB.data(); }                  // (staged form) the slot itself is never an output buffer
    else std::swap(cur, out);
  }
  if ((int64_t)len > H.slotCap) { H.status[b] = -KZ_ERR_PROCESS_BLOCK; H.len[b] = 0; return; }   // more than a block: "incorrectly decompressed"
  if (hostSlot) { if (cur != slot) memcpy(slot, cur, (size_t)len); }
  else if (hipMemcpy(slot, cur, (size_t)len, hipMemcpyHostToDevice) != hipSuccess) { H.fail = 1; return; }
  H.len[b] = len;
}
// The host inverse stages of one chunk whose blocks live in HBM: ONE gather kernel brings the blocks a host stage applies to into
// pinned slots (exact lengths), the host pool decodes them there, ONE scatter kernel puts the results back.  Runs on its own stream
// beside the main stream's next chunk.  (Before: a synchronous D2H and H2D copy per block.)
struct HostInvSub { HostInv* H; int b0; };
static void host_inverse_block_sub(int i, void* arg) { HostInvSub& S = *(HostInvSub*)arg; host_inverse_block(S.b0 + i, S.H); }
static void host_inverse_chunk_staged(kz_ctx* ctx, HostInv* H, int cnt, int ring) {
  // d_aux: [0, cnt) lengths in, [cnt, 2 cnt) mask in, [2 cnt, 3 cnt) lengths out, [3 cnt, 4 cnt) mask out
  std::vector<int32_t> hm((size_t)cnt * 4);
  int any = 0;
  for (int b = 0; b < cnt; b++) {
    bool need = !H->status[b] && H->len[b] > 0;
    if (need) { bool a = false; for (int i = 0; i < H->hp; i++) a |= !(H->skip[b] & (1 << (7 - i))); need = a; }
    hm[b] = H->len[b]; hm[cnt + b] = need ? 1 : 0; any |= need ? 1 : 0;
  }
  if (!any) return;
  hipStream_t hs = ctx->hiStream[ring];
  int32_t* d_aux = (int32_t*)ctx->hiAux[ring].p;
  if (hipSetDevice(ctx->device) != hipSuccess ||
      hipMemcpyAsync(d_aux, hm.data(), (size_t)cnt * 8, hipMemcpyHostToDevice, hs) != hipSuccess) { H->fail = 1; return; }
  // sub-chunks: the gather of sub-chunk j + 1 and the scatter of sub-chunk j - 1 run under the host stages of sub-chunk j
  const int SUB = 64;
  const int nsub = (cnt + SUB - 1) / SUB;
  std::vector<hipEvent_t> ev((size_t)nsub, nullptr);
  auto gather = [&](int j) {
    const int j0 = j * SUB, c = std::min(SUB, cnt - j0);
    hipLaunchKernelGGL(k_copy_bytes, dim3(32, c), dim3(256), 0, hs, H->dbuf + (int64_t)j0 * H->dstride, H->dstride, H->pin + (int64_t)j0 * H->pinSlot, H->pinSlot,
                       d_aux + j0, (const int32_t*)nullptr, d_aux + cnt + j0);
    if (hipEventCreateWithFlags(&ev[j], hipEventDisableTiming) != hipSuccess || hipEventRecord(ev[j], hs) != hipSuccess) H->fail = 1;
  };
  gather(0);
  for (int j = 0; j < nsub && !H->fail; j++) {
    const int j0 = j * SUB, c = std::min(SUB, cnt - j0);
    if (j + 1 < nsub) gather(j + 1);
    if (H->fail || hipEventSynchronize(ev[j]) != hipSuccess) { H->fail = 1; break; }
    HostInvSub S{H, j0};
    kz_parallel_for(c, KZ_HOST_STAGE_THREADS, host_inverse_block_sub, &S);
    if (H->fail) break;
    for (int b = j0; b < j0 + c; b++) { hm[2 * cnt + b] = H->len[b]; hm[3 * cnt + b] = (hm[cnt + b] && !H->status[b]) ? 1 : 0; }
    if (hipMemcpyAsync(d_aux + 2 * cnt + j0, hm.data() + 2 * cnt + j0, (size_t)c * 4, hipMemcpyHostToDevice, hs) != hipSuccess ||
        hipMemcpyAsync(d_aux + 3 * cnt + j0, hm.data() + 3 * cnt + j0, (size_t)c * 4, hipMemcpyHostToDevice, hs) != hipSuccess) { H->fail = 1; break; }
    hipLaunchKernelGGL(k_copy_bytes, dim3(32, c), dim3(256), 0, hs, H->pin + (int64_t)j0 * H->pinSlot, H->pinSlot, H->dbuf + (int64_t)j0 * H->dstride, H->dstride,
                       d_aux + 2 * cnt + j0, (const int32_t*)nullptr, d_aux + 3 * cnt + j0);
  }
  if (hipStreamSynchronize(hs) != hipSuccess) H->fail = 1;
  for (auto e : ev) if (e) hipEventDestroy(e);
}

// ---- decoder: RANK / MTFT inverse and BWT inverse of a large batch, overlapped -------------------------------------------
// The RANK inverse is serial per block (one wave per block, no HBM traffic to speak of) and its launch lasts as long as its
// slowest block; the BWT inverse behind it is bandwidth bound.  Run back to back the second waits for blocks the first has long
// finished.  Here the expensive blocks (cost hint = length at the previous stage's input ~ non-zero ranks) take the RANK
// inverse on a side stream while the cheap ones go through RANK inverse AND BWT inverse on the main stream; the expensive
// blocks' BWT inverse follows.  Blocks live in fixed slots of the two ping-pong buffers, so the groups are lengths-masked
// views of the same batch.  Only for batches where both stages apply to the same blocks.
__global__ void k_merge_group(const int32_t* __restrict__ in, const int32_t* __restrict__ len, const int32_t* __restrict__ flag, const int32_t* __restrict__ old,
                              int32_t* __restrict__ lenOut, int32_t* __restrict__ applied, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B || !in[b]) return;                                     // other blocks: lenOut (the batch's own lengths) and applied stay
  lenOut[b] = flag[b] > 0 ? len[b] : old[b];
  applied[b] = flag[b] > 0 ? 1 : 0;
}
// the arrays of one group copied into those of a view that holds several groups (their BWT inverses run as one: overlap_finish)
__global__ void k_gather_group(const int32_t* __restrict__ in, const int32_t* __restrict__ len, const int32_t* __restrict__ flag, const int32_t* __restrict__ old,
                               int32_t* __restrict__ mIn, int32_t* __restrict__ mLen, int32_t* __restrict__ mFlag, int32_t* __restrict__ mOld, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B || !in[b]) return;
  mIn[b] = 1; mLen[b] = len[b]; mFlag[b] = flag[b]; mOld[b] = old[b];
}
// the TEXT inverse on the device (kz_text_gpu.hip) instead of the host stage: opt-in, KZ_TEXT_GPU=1 (read per call).  Its first
// form is one serial walk per block: 1.07 s for 1 536 text blocks of 4 MiB side by side -- and as long for 384 of them -- where 16
// host CPUs need 0.87 s under the GPU's next chunk; it pays only where the host has next to no CPUs for the process.
// TEXT inverse on the device (kz_text_gpu.hip).  KZ_TEXT_GPU: unset = the row form (three waves per block) for streams whose entropy
// coder selects TextCodec2 (every coder but FPAQ / the CM family, TextCodec.java:73-88) in batches of KZ_TEXT_GPU_MIN blocks or more
// (512: a block takes the kernel ~0.1 s however few there are, the host stage ~10 ms per block and thread), else the host stage;
// 0 = host stage only; 1 = row form, three waves; 2 = serial token walk (both codecs); 3 = row form, one wave (1-3: any batch).
static int text_gpu_form(const kz_ctx* ctx, uint32_t entropyType, int nBlocks) {
  if (ctx->sw.textGpu >= 0) return ctx->sw.textGpu;
  return (entropyType == KZ_E_FPAQ || nBlocks < ctx->sw.textGpuMin) ? 0 : 1;
}
static bool text_gpu_on(const kz_ctx* ctx, uint32_t entropyType, int nBlocks) { return text_gpu_form(ctx, entropyType, nBlocks) != 0; }
// TEXT forward on the device (kz_text_fwd_gpu.hip): TextCodec2 streams (every entropy coder but FPAQ) in batches of KZ_TEXT_FWD_GPU_MIN
// blocks (256) or more -- a block's dictionary walk takes the kernel ~0.1 s however few there are, so small batches stay on the host's
// chunk pipeline (measured with 16 host CPUs, level-exact -l 5 encode of 64 / 128 / 256 / 512 blocks: host 102 / 159 / 293 / 529 ms,
// device 157 / 188 / 247 / 379 ms).  KZ_TEXT_FWD_GPU=0: host stage, =1: any batch size.  (Read per call: the tests force both.)
static bool text_fwd_gpu_on(const kz_ctx* ctx, uint32_t entropyType, int nBlocks) {
  const bool type2 = entropyType == KZ_E_NONE || entropyType == KZ_E_ANS0 || entropyType == KZ_E_HUFFMAN;
  if (!type2) return false;
  if (ctx->sw.textFwdGpu >= 0) return ctx->sw.textFwdGpu == 1;
  return nBlocks >= ctx->sw.textFwdGpuMin;
}
// UTF inverse on the device (kz_text_gpu.hip: parallel inside a block, so for any batch); KZ_UTF_GPU=0 keeps it on the host
static bool utf_gpu_on(const kz_ctx* ctx) { return ctx->sw.utfGpu != 0; }

static int fuse_min_blocks(const kz_ctx* ctx) { return ctx->sw.fuseMinBlocks; }   // batches below this take the stages one after the other
// One cost class of the batch: a lengths-masked view of the same slots with its own length / flag arrays.
struct OverlapGroup {
  std::vector<int32_t> in;
  int32_t *d_in = nullptr, *len = nullptr, *len2 = nullptr, *flag = nullptr, *old = nullptr;
  int prio = 0;
  int ev = -1;                            // index of the side stream / join event of its RANK inverse (-1: on the main stream)
  double tPre = 0, tRank = 0, tBwt = 0;   // modelled seconds of the group's entropy + ZRLT pass, RANK inverse and BWT inverse
  kz_batch v;
};
// groups[0] goes on the main stream (the cheapest class), groups[1..] on the side streams, most expensive last
struct Overlap { std::vector<OverlapGroup> groups; std::vector<int> launchOrder, bwtOrder; int mainGroup = -1; int64_t started = 0;
                 std::vector<std::vector<int>> bwtCalls; };   // bwtCalls: the groups of bwtOrder whose BWT inverses run as ONE call (empty: one call per group)
#define KZ_OVERLAP_MAXG 5
// host side: classes by cost relative to the largest: >= 3/4 | >= 3/8 | >= 3/16 | >= 3/32 | the rest; classes of fewer than 8
// blocks join the class below (the cheapest one: the class above).  false when there is nothing to overlap.
static bool overlap_classify(const kz_ctx* ctx, int B, const std::vector<int32_t>& h_mask, const std::vector<int32_t>& cost, Overlap& O) {
  if (B < fuse_min_blocks(ctx)) return false;
  int64_t maxCost = 0;
  for (int b = 0; b < B; b++) if (h_mask[b]) maxCost = std::max<int64_t>(maxCost, cost[b]);
  int nClasses = ctx->sw.overlapClasses;                           // 3: main + two side streams fit HIP's default of 4 hardware queues
  nClasses = std::min(std::max(nClasses, 2), KZ_OVERLAP_MAXG);
  std::vector<int> cls(B, -1);
  int cnt[KZ_OVERLAP_MAXG] = {0};
  for (int b = 0; b < B; b++) {
    if (!h_mask[b]) continue;
    int c = nClasses - 1;                                          // most expensive
    int64_t lim = maxCost * 3;                                     // cost * 4 >= maxCost * 3, then halving
    while (c > 0 && (int64_t)cost[b] * 4 < lim) { c--; lim >>= 1; }
    cls[b] = c;
    cnt[c]++;
  }
  int to[KZ_OVERLAP_MAXG];
  for (int c = 0; c < KZ_OVERLAP_MAXG; c++) to[c] = c;
  for (int c = nClasses - 1; c >= 1; c--)
    if (cnt[c] > 0 && cnt[c] < 8) { cnt[c - 1] += cnt[c]; cnt[c] = 0; for (int k = 0; k < KZ_OVERLAP_MAXG; k++) if (to[k] == c) to[k] = c - 1; }
  if (cnt[0] > 0 && cnt[0] < 8) {                                  // the cheapest class is too small: the next one takes the main stream
    int up = -1;
    for (int c = 1; c < nClasses; c++) if (cnt[c] > 0) { up = c; break; }
    if (up > 0) { cnt[up] += cnt[0]; cnt[0] = 0; for (int k = 0; k < KZ_OVERLAP_MAXG; k++) if (to[k] == 0) to[k] = up; }
  }
  if (ctx->sw.traceSched) {
    int raw[KZ_OVERLAP_MAXG] = {0};
    for (int b = 0; b < B; b++) if (cls[b] >= 0) raw[cls[b]]++;
    fprintf(stderr, "[sched] classes (cheap..expensive) raw %d %d %d %d %d merged %d %d %d %d %d maxCost %lld\n", raw[0], raw[1], raw[2], raw[3], raw[4],
            cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], (long long)maxCost);
  }
  int order[KZ_OVERLAP_MAXG], n = 0;
  for (int c = 0; c < nClasses; c++) if (cnt[c] > 0) order[n++] = c;
  if (n < 2) return false;
  O.groups.resize(n);
  // issue priority of the classes' RANK-inverse waves (s_setprio 3 / 1 / 0).  Round 6: the class whose BWT inverse comes FIRST
  // gets the highest one -- the main stream has nothing to do until that class's last RANK chain has ended, and the long chains of the
  // other classes end under its BWT inverse anyway (bulk decode 595 -> 539 ms; with the 32-bit forms of round 5 the opposite order
  // was the faster one: the incompressible class's chain was then the critical path)
  for (int g = 0; g < n; g++) {
    O.groups[g].in.assign(B, 0);
    O.groups[g].prio = (g == n - 1) ? 2 : (g > 0 ? 1 : 0);          // (the wide schedule: overlap_plan re-assigns by the order of the BWT inverses)
  }
  for (int b = 0; b < B; b++) {
    if (cls[b] < 0) continue;
    const int c = to[cls[b]];
    for (int g = 0; g < n; g++) if (order[g] == c) O.groups[g].in[b] = 1;
  }
  return true;
}
static int overlap_streams(kz_ctx* ctx);
static int overlap_alloc(kz_ctx* ctx, int B, Overlap& O) {
  { const int qrc = overlap_streams(ctx); if (qrc) return qrc; }
  for (auto& G : O.groups) {
    int32_t* d = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4 * 5);
    if (!d) { snprintf(ctx->err, sizeof(ctx->err), "overlapped inverse: arena overflow"); return -KZ_ERR_DEVICE; }
    G.d_in = d; G.len = d + B; G.len2 = d + 2 * B; G.flag = d + 3 * B; G.old = d + 4 * B;
    KZ_HIP(hipMemcpyAsync(G.d_in, G.in.data(), (size_t)B * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  KZ_HIP(kz_stream_sync(ctx, ctx->stream));                        // pageable sources
  return 0;
}
static int overlap_view(kz_ctx* ctx, const kz_batch& bt, OverlapGroup& G) {
  const int B = bt.B;
  G.v = bt;
  G.v.d_len = G.len; G.v.d_len2 = G.len2; G.v.d_flag = G.flag;
  G.v.prio = G.prio;
  for (int b = 0; b < B; b++) if (!G.in[b]) G.v.h_len[b] = 0;
  // the view's device lengths come from the HOST mirror: a block that failed in an earlier stage has length 0 there while the
  // batch's device array still holds its old length, and the stages size their scratch by the host's count of non-empty blocks
  KZ_HIP(hipMemcpyAsync(G.len, G.v.h_len.data(), (size_t)B * 4, hipMemcpyHostToDevice, ctx->stream));
  KZ_HIP(hipMemcpyAsync(G.old, G.len, (size_t)B * 4, hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}
// Does this process get as many concurrent hardware queues as the wide schedule needs (main + three side streams)?  HIP
// multiplexes its streams over GPU_MAX_HW_QUEUES hardware queues (4 unless the variable says otherwise when the runtime
// starts: the APPLICATION exports GPU_MAX_HW_QUEUES=8 before its first HIP call, the library never touches the environment;
// a caller that forgets it gets the three-stream form below, which this measurement selects), and two streams on one queue run
// one after the other.  Measured once per context: four 2 ms spin kernels on the four streams take 2 ms or 4+.
__global__ void k_spin(long long ticks) {
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
static int overlap_streams(kz_ctx* ctx) {
  for (int i = 0; i < 5; i++) if (!ctx->side[i]) {
    KZ_HIP(hipStreamCreateWithFlags(&ctx->side[i], hipStreamNonBlocking));
    KZ_HIP(hipEventCreateWithFlags(&ctx->evJoin[i], hipEventDisableTiming));
  }
  if (ctx->sw.wideQueues >= 0) { ctx->wideNow = ctx->sw.wideQueues; return 0; }   // (the tests force both schedules)
  if (ctx->wideQueues < 0) {
    KZ_HIP(kz_stream_sync(ctx, ctx->stream));
    hipStream_t q[4] = {ctx->stream, ctx->side[0], ctx->side[1], ctx->side[2]};
    for (int r = 0; r < 2; r++) {                                   // first round: warm-up (queue creation, code load)
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 4; i++) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, q[i], (long long)200000);   // 100 MHz: 2 ms
      for (int i = 0; i < 4; i++) KZ_HIP(hipStreamSynchronize(q[i]));
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      ctx->wideQueues = ms < 3.5 ? 1 : 0;
    }
    if (ctx->sw.traceSched) fprintf(stderr, "[sched] four streams run %s\n", ctx->wideQueues ? "side by side" : "on shared hardware queues: three-stream schedule");
  }
  ctx->wideNow = ctx->wideQueues;
  return 0;
}
// Order of the groups.  Wide schedule (four concurrent hardware queues): every group's RANK inverse runs on its own side stream;
// the main stream prepares the groups one after the other (entropy decoding + ZRLT inverse, when the caller has not done them
// yet) and then runs the BWT inverses as the RANK inverses finish (HBM-bound kernels with huge grids do not share the GPU with
// another queue's kernels anyway: chains "RANK, BWT" per stream were measured, the BWT stages ran one after the other and
// started late).  The order of the preparation decides which serial chain is on the critical path; it is picked by trying all
// permutations against a three-constant model: RANK inverse = 115 ns per byte of the group's longest ZRLT-coded block, at most 64 ns
// per byte of the block (a serial chain per block; batches this large keep two waves per SIMD), BWT inverse = 40 ps per byte, preparation = 12 ps per
// entropy-coded byte (MI355X, profiles/).  Narrow schedule: two side streams, the cheapest group's RANK inverse on the main
// stream behind all the preparations.
static void overlap_plan(kz_ctx* ctx, Overlap& O, const std::vector<int32_t>& zlen, const std::vector<int32_t>& rawp, int blockSize, bool withPre) {
  const int n = (int)O.groups.size();
  O.mainGroup = -1;
  if (!ctx->wideNow || n > 3) {
    O.launchOrder.clear(); O.bwtOrder.clear();
    for (int g = n - 1; g >= 0; g--) O.launchOrder.push_back(g);
    for (int g = 0; g < n; g++) O.bwtOrder.push_back(g);
    O.mainGroup = 0;
    return;
  }
  for (auto& G : O.groups) {
    int64_t maxZ = 0, pre = 0, cnt = 0, raws = 0;
    for (size_t b = 0; b < G.in.size(); b++) if (G.in[b]) { maxZ = std::max<int64_t>(maxZ, zlen[b]); cnt++; raws += rawp[b] ? 1 : 0; pre += rawp[b] ? zlen[b] / 16 : zlen[b]; }
    // round 6: a block whose ranks are nearly all non-zero and mostly >= 64 (ZRLT-coded length ~ its own: incompressible data)
    // runs in the interleaved keyed rows of kz_sbrt_f64.h (blocks up to 8 MiB), rows without ranks >= 64 in the by-position keyed
    // rows.  Measured in the bulk batch (two waves per SIMD, the other classes' waves and the BWT inverses beside them); the class
    // of cheap blocks (three fifths of the batch) takes 1.65 x what its longest block suggests: the factor 1 + blocks / (2 x SIMDs'
    // worth) below.  (Tried and measured slower: the large class cut in two at its median cost on six streams, 643 - 805 ms against
    // 593; the longest chain first, 695 against 665.)
    const bool keyedForms = ctx->sw.sbrtForm != 0 && blockSize <= (1 << 23);
    const bool keyed = keyedForms && maxZ >= (int64_t)blockSize - (blockSize >> 5) && raws * 2 > cnt;   // incompressible (stored raw): ranks of every depth
    const double crowd = 1.0 + 0.55 * (double)cnt / (4.0 * (double)std::max(1, ctx->numCUs));
    // (with the keyed rows: incompressible class 464 ms = 95 ns per byte of the block, skewed bytes 331 ms = at most 65 ns per byte,
    // cheap blocks 232 ms = 86 ns per ZRLT-coded byte, each times the crowding factor)
    if (!keyedForms) G.tRank = 115e-9 * (double)maxZ * crowd;
    else G.tRank = (keyed ? 95e-9 * (double)blockSize : std::min(86e-9 * (double)maxZ, 65e-9 * (double)blockSize)) * crowd;
    G.tBwt = 40e-12 * (double)cnt * (double)blockSize;
    G.tPre = withPre ? 12e-12 * (double)pre : 0.0;
  }
  std::vector<int> perm(n), best;
  for (int i = 0; i < n; i++) perm[i] = i;
  double bestT = 1e30;
  do {
    double t = 0, endR[KZ_OVERLAP_MAXG];
    for (int i = 0; i < n; i++) { t += O.groups[perm[i]].tPre; endR[perm[i]] = t + O.groups[perm[i]].tRank; }
    std::vector<int> byEnd(perm);
    std::sort(byEnd.begin(), byEnd.end(), [&](int a, int b) { return endR[a] < endR[b]; });
    double tb = t;                                                  // the main stream is busy with the preparations until t
    for (int g : byEnd) tb = std::max(tb, endR[g]) + O.groups[g].tBwt;
    if (tb < bestT - 1e-9) { bestT = tb; best = perm; O.bwtOrder = byEnd; }
  } while (std::next_permutation(perm.begin(), perm.end()));
  O.launchOrder = best;
  {
    // Round 6: groups whose RANK chains are (by the model) over before the main stream gets to them have their BWT inverses run as
    // ONE call: the walk of 416 blocks takes 45 ms, of 832 about 80, and the per-call tails and launches are paid once
    // (the two trailing classes of the bulk batch: 2 x 60 -> 85 ms).
    double t = 0, endR[KZ_OVERLAP_MAXG];
    for (int i = 0; i < n; i++) { t += O.groups[O.launchOrder[i]].tPre; endR[O.launchOrder[i]] = t + O.groups[O.launchOrder[i]].tRank; }
    double tb = t;
    O.bwtCalls.clear();
    for (size_t i = 0; i < O.bwtOrder.size();) {
      std::vector<int> call(1, O.bwtOrder[i]);
      const double begin = std::max(tb, endR[O.bwtOrder[i]]);
      double dur = O.groups[O.bwtOrder[i]].tBwt;
      size_t k = i + 1;
      // (only chains over by then: taking in a chain that ends later -- tried with `<= begin + dur` -- holds the call back behind the longest
      // one: bulk decode 544 -> 587 ms)
      while (i > 0 && k < O.bwtOrder.size() && endR[O.bwtOrder[k]] <= begin) { call.push_back(O.bwtOrder[k]); dur += O.groups[O.bwtOrder[k]].tBwt; k++; }
      O.bwtCalls.push_back(call);
      tb = begin + dur;
      i = k;
    }
  }
  if ((int)zlen.size() > 4 * ctx->numCUs)                           // (batches of one wave per SIMD or less: the longest chain keeps the highest priority)
    for (size_t i = 0; i < O.bwtOrder.size(); i++) O.groups[O.bwtOrder[i]].prio = i == 0 ? 2 : (i == 1 ? 1 : 0);
  if (ctx->sw.traceSched) {
    fprintf(stderr, "[sched] plan %.0f ms: launch", bestT * 1e3);
    for (int g : O.launchOrder) fprintf(stderr, " %d(pre %.0f rank %.0f bwt %.0f)", g, O.groups[g].tPre * 1e3, O.groups[g].tRank * 1e3, O.groups[g].tBwt * 1e3);
    fprintf(stderr, " | bwt order"); for (int g : O.bwtOrder) fprintf(stderr, " %d", g); fprintf(stderr, "\n");
  }
}
// RANK inverse of group g: on the main stream for the narrow schedule's main group, else on side stream k (everything queued
// on the main stream so far is waited for)
static int overlap_start_rank(kz_ctx* ctx, const kz_batch& bt, int mode, Overlap& O, int g, int k) {
  OverlapGroup& G = O.groups[g];
  int rc = overlap_view(ctx, bt, G);
  if (rc) return rc;
  {                                                                 // the launches so far occupy the first SIMDs of every CU
    int64_t act = 0;
    for (int b = 0; b < bt.B; b++) act += G.in[b] ? 1 : 0;
    G.v.slotRot = (int)((O.started / std::max(1, (bt.B + 7) / 8)) & 3);
    O.started += act;
  }
  if (g == O.mainGroup) { G.ev = -1; return kz_stage_sbrt_inverse(ctx, G.v, mode); }
  G.ev = k;
  KZ_HIP(kz_stream_sync(ctx, ctx->stream));
  hipStream_t& sd = ctx->side[k];
  std::swap(ctx->stream, sd);
  rc = kz_stage_sbrt_inverse(ctx, G.v, mode);
  if (!rc) { hipError_t e = hipEventRecord(ctx->evJoin[k], ctx->stream); if (e != hipSuccess) rc = -KZ_ERR_DEVICE; }
  std::swap(ctx->stream, sd);
  return rc;
}
// the groups' BWT inverses on the main stream, in the planned order; merges lengths and flags
static int overlap_finish(kz_ctx* ctx, Pipe& P, Overlap& O, std::vector<int32_t>& h_applied) {
  kz_batch& bt = P.bt;
  const int B = bt.B;
  hipStream_t st = ctx->stream;
  int rc = 0;
  if (O.bwtCalls.empty()) for (int g : O.bwtOrder) O.bwtCalls.push_back(std::vector<int>(1, g));
  // a call of several groups works on a view of its own: the groups' masks, lengths, flags and old lengths gathered (allocated in
  // front of `mark`: the stages' scratch behind it is given back between the calls)
  struct Merged { OverlapGroup M; bool used = false; };
  std::vector<Merged> merged(O.bwtCalls.size());
  for (size_t c = 0; c < O.bwtCalls.size(); c++) {
    if (O.bwtCalls[c].size() < 2) continue;
    OverlapGroup& M = merged[c].M;
    int32_t* d = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4 * 5);
    if (!d) { snprintf(ctx->err, sizeof(ctx->err), "overlapped inverse: arena overflow"); return -KZ_ERR_DEVICE; }
    M.d_in = d; M.len = d + B; M.len2 = d + 2 * B; M.flag = d + 3 * B; M.old = d + 4 * B;
    merged[c].used = true;
  }
  const size_t mark = ctx->arenaTop;
  for (size_t c = 0; c < O.bwtCalls.size(); c++) {
    const std::vector<int>& call = O.bwtCalls[c];
    for (int g : call) if (O.groups[g].ev >= 0) KZ_HIP(hipStreamWaitEvent(st, ctx->evJoin[O.groups[g].ev], 0));
    ctx->arenaTop = mark;                                           // same stream, in order: the scratch is free again
    if (!merged[c].used) { rc = kz_stage_bwt_inverse(ctx, O.groups[call[0]].v); if (rc) return rc; continue; }
    OverlapGroup& M = merged[c].M;
    M.v = O.groups[call[0]].v;                                      // (same buffers, same `cur`: every view has run one stage)
    KZ_HIP(hipMemsetAsync(M.d_in, 0, (size_t)B * 4 * 5, st));
    for (size_t q = 0; q < call.size(); q++) {
      OverlapGroup& G = O.groups[call[q]];
      if (G.v.cur != M.v.cur) { snprintf(ctx->err, sizeof(ctx->err), "overlapped inverse: views out of step"); return -KZ_ERR_DEVICE; }
      if (q > 0) for (int b = 0; b < B; b++) if (G.in[b]) M.v.h_len[b] = G.v.h_len[b];
      hipLaunchKernelGGL(k_gather_group, dim3((B + 255) / 256), dim3(256), 0, st, G.d_in, G.v.d_len, G.v.d_flag, G.old, M.d_in, M.len, M.flag, M.old, B);
    }
    M.v.d_len = M.len; M.v.d_len2 = M.len2; M.v.d_flag = M.flag;
    rc = kz_stage_bwt_inverse(ctx, M.v);
    if (rc) return rc;
  }
  // every view is back in the buffer it started from (two stages each); the batch's own state is untouched except lengths
  KZ_HIP(hipMemsetAsync(P.d_applied, 0, (size_t)B * 4, st));
  for (size_t c = 0; c < O.bwtCalls.size(); c++) {
    if (merged[c].used) {
      OverlapGroup& M = merged[c].M;
      KZ_LAUNCH(ctx, KID_MASK_LEN, k_merge_group, dim3((B + 255) / 256), dim3(256), M.d_in, M.v.d_len, M.v.d_flag, M.old, bt.d_len, P.d_applied, B);
    } else {
      OverlapGroup& G = O.groups[O.bwtCalls[c][0]];
      KZ_LAUNCH(ctx, KID_MASK_LEN, k_merge_group, dim3((B + 255) / 256), dim3(256), G.d_in, G.v.d_len, G.v.d_flag, G.old, bt.d_len, P.d_applied, B);
    }
  }
  KZ_HIP(hipMemcpyAsync(ctx->hpin + B, P.d_applied, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  rc = sync_lengths(ctx, bt);
  if (rc) return rc;
  h_applied.resize(B);
  for (int b = 0; b < B; b++) h_applied[b] = ctx->hpin[B + b];
  return 0;
}
// returns 1 when it ran both stages (h_applied = blocks both were applied to), 0 when the schedule does not apply, <0 on error
static int overlapped_rank_bwt_inverse(kz_ctx* ctx, Pipe& P, int mode, const std::vector<int32_t>& h_mask, const std::vector<int32_t>& cost,
                                       std::vector<int32_t>& h_applied) {
  Overlap O;
  if (!overlap_classify(ctx, P.bt.B, h_mask, cost, O)) return 0;
  int rc = overlap_alloc(ctx, P.bt.B, O);
  if (rc) return rc;
  {
    std::vector<int32_t> none(P.bt.B, 0);
    overlap_plan(ctx, O, P.bt.h_len, none, P.bt.maxN, false);
  }
  for (size_t k = 0; k < O.launchOrder.size(); k++) { rc = overlap_start_rank(ctx, P.bt, mode, O, O.launchOrder[k], (int)k); if (rc) return rc; }
  rc = overlap_finish(ctx, P, O, h_applied);
  return rc ? rc : 1;
}

// does a batch of nBlocks blocks of this chain run its TEXT stage on the device (kz_stream.hip: such chunks are not pre-staged on the host)
bool kz_text_fwd_gpu_applies(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType, int nBlocks) {
  int types[8];
  const int nb = split_types(transformType, types);
  int hp = 0;
  while (hp < nb && kz_is_host_transform(types[hp])) hp++;
  return hp > 0 && !ctx->skipBlocks && types[0] == KZ_T_TEXT && (hp == 1 || types[1] == KZ_T_UTF) && text_fwd_gpu_on(ctx, entropyType, nBlocks);
}

// =================================================================================================
// encode
// blockSize = the stream's "blockSize" entry as TEXT reads it, fixed when the call was made (a queued job keeps the value of its
// submit time: later kz_ctx_set_block_size calls or kz_compress's scope do not reach it)
// pre (optional): the host stages' results for exactly these blocks, computed ahead (host_prestage) by the caller.
int32_t kz_encode_blocks_pre(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType, int32_t blockSize,
                             const uint8_t* in, int64_t inStride, const int32_t* lengths, int32_t nBlocks,
                             uint8_t* out, int64_t outStride, kz_block_result* results, int32_t memKind, const HostPre* pre);
// blocks per step of the encoder's host-stage pipeline: 256 for bulk batches; batches of a few dozen blocks (the silesia shape: 51)
// take thirds, so that the TEXT / UTF stages of all but the first third run under the GPU work of the third before
static int host_chunk_blocks(const kz_ctx* ctx, int B) {
  if (ctx->sw.hostChunk > 0) return ctx->sw.hostChunk < 8 ? 8 : ctx->sw.hostChunk;
  if (B >= 512) return 256;
  return std::max(12, (B + 2) / 3);
}
static int32_t encode_blocks_bs(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType, int32_t blockSize,
                                const uint8_t* in, int64_t inStride, const int32_t* lengths, int32_t nBlocks,
                                uint8_t* out, int64_t outStride, kz_block_result* results, int32_t memKind) {
  return kz_encode_blocks_pre(ctx, transformType, entropyType, blockSize, in, inStride, lengths, nBlocks, out, outStride, results, memKind, nullptr);
}
int32_t kz_encode_blocks_pre(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType, int32_t blockSize,
                             const uint8_t* in, int64_t inStride, const int32_t* lengths, int32_t nBlocks,
                             uint8_t* out, int64_t outStride, kz_block_result* results, int32_t memKind, const HostPre* pre) {
  if (!ctx) return -KZ_ERR_INVALID_PARAM;
  if (nBlocks <= 0) return 0;
  KZ_HIP(hipSetDevice(ctx->device));
  int types[8];
  const int nb = split_types(transformType, types);
  for (int i = 0; i < nb; i++) if (!transform_supported(types[i])) { snprintf(ctx->err, sizeof(ctx->err), "unsupported transform id %d", types[i]); return -KZ_ERR_INVALID_CODEC; }
  if (!entropy_supported((int)entropyType)) { snprintf(ctx->err, sizeof(ctx->err), "unsupported entropy id %u", entropyType); return -KZ_ERR_INVALID_CODEC; }
  const int B = nBlocks;
  const int hp = host_prefix(ctx, types, nb);
  if (hp < 0) return hp;
  int maxN = 0;
  for (int b = 0; b < B; b++) { if (lengths[b] < 0) return -KZ_ERR_INVALID_PARAM; maxN = std::max(maxN, lengths[b]); }
  const int maxLen = seq_max_len(types, nb, maxN);
  // blocks go up to the reference's 1 GiB (BWT.java:59); BWT / RANK / MTFT blocks of 2^24 bytes and more take the wide forms of the
  // inverse kernels (8-byte links in kz_bwt_inv.hip, the plain list in kz_sbrt.hip: round 5), slower but the same bytes
  if (maxLen > KZ_MAX_BLOCK) { snprintf(ctx->err, sizeof(ctx->err), "block of %d bytes: blocks go up to %d bytes", maxN, KZ_MAX_BLOCK - 1057); return -KZ_ERR_BLOCK_SIZE; }
  ChainSpec CS; CS.nb = nb; CS.entropy = (int)entropyType; for (int i = 0; i < nb; i++) CS.types[i] = types[i];
  {
    // bound the scratch arena: the suffix sort needs ~43 B per input byte, so very large batches are
    // processed as consecutive sub-batches (blocks are independent; results are identical)
    bool hasBwt = false;
    for (int i = 0; i < nb; i++) hasBwt |= (types[i] == KZ_T_BWT);
    ChainSpec noBwt = CS;
    for (int i = 0; i < noBwt.nb; i++) if (noBwt.types[i] == KZ_T_BWT) noBwt.types[i] = KZ_T_NONE;
    const size_t perBlock = pipeline_scratch(1, maxLen, false, noBwt) + (size_t)maxLen * 2 + 8192 + (size_t)(memKind == KZ_MEM_HOST ? outStride : 0) + (1 << 16);
    const size_t avail = hasBwt ? kz_arena_budget() / 2 : kz_arena_budget();      // the other half: suffix-sort groups
    const int maxB = (int)std::min<size_t>(KZ_MAX_BATCH, std::max<size_t>(1, avail / perBlock));   // grid.y carries the block index
    if (B > maxB) {
      for (int b0 = 0; b0 < B; b0 += maxB) {
        const int cnt = std::min(maxB, B - b0);
        HostPre view;                                                 // a pre-staged batch is split like any other: every sub-batch sees its
        if (pre) {                                                    // share of the host stages' results (the slots stay where they are)
          view.outLen.assign(pre->outLen.begin() + b0, pre->outLen.begin() + b0 + cnt);
          view.skip.assign(pre->skip.begin() + b0, pre->skip.begin() + b0 + cnt);
          view.dtype.assign(pre->dtype.begin() + b0, pre->dtype.begin() + b0 + cnt);
          view.changed.assign(pre->changed.begin() + b0, pre->changed.begin() + b0 + cnt);
          view.store = pre->store + (int64_t)b0 * pre->slot;
          view.slot = pre->slot;
          view.pinned = pre->pinned;
        }
        int rc = kz_encode_blocks_pre(ctx, transformType, entropyType, blockSize, in + (int64_t)b0 * inStride, inStride, lengths + b0, cnt,
                                      out + (int64_t)b0 * outStride, outStride, results + b0, memKind, pre ? &view : nullptr);
        if (rc) return rc;
      }
      return 0;
    }
  }
  // ---- chains led by TEXT / UTF on large batches: the host stages of chunk k+1 run on a helper thread (and the host pool) while
  //      the GPU codes chunk k.  Blocks are independent, so chunking changes nothing in the output.  (With "skipBlocks" the copy
  //      decision comes from the device and precedes the host stages: that case takes the one-pass path below.) ----
  const bool textFwdGpu = !pre && kz_text_fwd_gpu_applies(ctx, transformType, entropyType, B);   // ONE predicate (kz_stream.hip asks the same one)
  if (hp > 0 && !pre && !ctx->skipBlocks && B >= 2 * host_chunk_blocks(ctx, B) && !textFwdGpu) {
    const int CH = host_chunk_blocks(ctx, B);
    const int nch = (B + CH - 1) / CH;
    const bool hostIn = memKind == KZ_MEM_HOST;
    struct Chunk { HostPre P; int rc = 0; };
    std::vector<Chunk> ck(nch);
    const int dev = ctx->device;
    for (int q = 0; q < 2; q++) {                                   // pinned, kept by the context: no fresh pages per chunk, DMA both ways
      int rc = kz_stage_reserve(ctx, ctx->hsOut[q], (size_t)host_pre_slot(maxLen) * (size_t)CH + 64, true);
      if (!rc && !hostIn) rc = kz_stage_reserve(ctx, ctx->hsIn[q], (size_t)CH * (size_t)maxN + 64, true);
      if (rc) return rc;
    }
    const bool trace = ctx->sw.tracePipe != 0;
    const auto tStart = std::chrono::steady_clock::now();
    auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tStart).count(); };
    auto stage = [&](int k) {
      const double t0 = ms();
      const int b0 = k * CH, cnt = std::min(CH, B - b0);
      const uint8_t* hsrc = in + (int64_t)b0 * inStride;
      int64_t hstride = inStride;
      if (!hostIn) {                                                // device input: one copy back for the host stages (legacy stream:
        uint8_t* back = ctx->hsIn[k & 1].p;                         // it does not wait for the context's non-blocking stream)
        if (hipSetDevice(dev) != hipSuccess) { ck[k].rc = -KZ_ERR_DEVICE; return; }
        if (inStride == (int64_t)maxN) {
          if (hipMemcpy(back, hsrc, (size_t)cnt * (size_t)maxN, hipMemcpyDeviceToHost) != hipSuccess) { ck[k].rc = -KZ_ERR_DEVICE; return; }
        } else if (hipMemcpy2D(back, (size_t)maxN, hsrc, (size_t)inStride, (size_t)maxN, (size_t)cnt, hipMemcpyDeviceToHost) != hipSuccess) { ck[k].rc = -KZ_ERR_DEVICE; return; }
        hsrc = back; hstride = maxN;
      }
      const double t1 = ms();
      host_prestage(types, hp, (int)entropyType, blockSize, maxLen, hsrc, hstride, lengths + b0, nullptr, cnt, ck[k].P, ctx->hsOut[k & 1].p);
      if (trace) fprintf(stderr, "[pipe] stage %d: copy back %.0f ms, host stages %.0f ms (at %.0f)\n", k, t1 - t0, ms() - t1, ms());
    };
    stage(0);
    for (int k = 0; k < nch; k++) {
      std::thread ahead;
      if (k + 1 < nch) ahead = std::thread(stage, k + 1);
      const int b0 = k * CH, cnt = std::min(CH, B - b0);
      int rc = ck[k].rc;
      const double tg = ms();
      if (!rc) rc = kz_encode_blocks_pre(ctx, transformType, entropyType, blockSize, in + (int64_t)b0 * inStride, inStride, lengths + b0, cnt,
                                         out + (int64_t)b0 * outStride, outStride, results + b0, memKind, &ck[k].P);
      ck[k].P = HostPre();
      const double tj = ms();
      if (ahead.joinable()) ahead.join();
      if (trace) fprintf(stderr, "[pipe] chunk %d: gpu %.0f ms, waited %.0f ms for the next stage (at %.0f)\n", k, tj - tg, ms() - tj, ms());
      if (rc) { if (rc == -KZ_ERR_DEVICE && !ctx->err[0]) snprintf(ctx->err, sizeof(ctx->err), "host stage: copy from the device failed"); return rc; }
    }
    return 0;
  }
  const int64_t needOut = kz_max_block_stream_bytes(maxN);
  if (outStride < needOut || (outStride & 3)) { snprintf(ctx->err, sizeof(ctx->err), "outStride %lld < %lld or not a multiple of 4", (long long)outStride, (long long)needOut); return -KZ_ERR_INVALID_PARAM; }
  const bool host = memKind == KZ_MEM_HOST;
  Pipe P;
  int64_t extra = (host ? (int64_t)outStride * B : 0) + (int64_t)B * (sizeof(kz_block_result) + 64) + (int64_t)B * 32 + 8192;
  if (textFwdGpu) {                     // the device TEXT forward runs first and gives its scratch back: it needs the larger of the two, not the sum
    const int64_t tf = (int64_t)kz_text_fwd_gpu_scratch(B, blockSize, maxLen), rest = (int64_t)pipeline_scratch(B, maxLen, false, CS);
    if (tf > rest) extra += tf - rest;
  }
  int rc = pipe_setup(ctx, P, B, maxLen, extra, false, CS);
  if (rc) return rc;
  kz_batch& bt = P.bt;
  hipStream_t st = ctx->stream;
  u8* d_out = host ? (u8*)kz_arena_alloc(ctx, (size_t)outStride * B) : out;
  kz_block_result* d_res = (kz_block_result*)kz_arena_alloc(ctx, (size_t)B * sizeof(kz_block_result));
  FrameEnc F;
  F.skipFlags = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  F.hdrBytes = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  F.isCopy = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  F.fallback = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  F.bits = (int64_t*)kz_arena_alloc(ctx, (size_t)B * 8);
  F.hash = (u64*)kz_arena_alloc(ctx, (size_t)B * 8);
  F.chk = ctx->checksum;
  F.nbFunctions = nb;
  if (!F.bits || !d_res || !d_out || !F.hash) { snprintf(ctx->err, sizeof(ctx->err), "encode: arena overflow"); return -KZ_ERR_DEVICE; }

  // ---- load blocks into HBM (device input: one strided copy kernel instead of one memcpy per block) ----
  for (int b = 0; b < B; b++) bt.h_len[b] = lengths[b];
  KZ_HIP(hipMemcpyAsync(bt.d_len, lengths, (size_t)B * 4, hipMemcpyHostToDevice, st));
  if (host) {
    for (int b = 0; b < B; b++) {
      if (lengths[b] == 0) continue;
      KZ_HIP(hipMemcpyAsync(bt.buf[0] + (int64_t)b * bt.stride, in + (int64_t)b * inStride, (size_t)lengths[b], hipMemcpyHostToDevice, st));
    }
  } else {
    KZ_LAUNCH(ctx, KID_COPY_BYTES, k_copy_bytes, dim3(64, B), dim3(256), in, inStride, bt.buf[0], bt.stride, bt.d_len, (const int32_t*)nullptr, (const int32_t*)nullptr);
  }
  KZ_HIP(hipMemsetAsync(d_out, 0, (size_t)outStride * B, st));       // bit-concat ORs into zeroed words
  // the writer's per-block "dataType" tag from the first four bytes (CompressedOutputStream.java:795-804); the stages
  // that care (MM, LZ/LZX) read and update it on the device
  rc = kz_block_data_types(ctx, bt, KZ_DT_UNDEFINED, true);
  if (rc) return rc;
  if (F.chk) { rc = kz_block_hashes(ctx, bt.buf[0], bt.stride, bt.d_len, B, F.chk, F.hash); if (rc) return rc; }

  // ---- transform chain (Sequence.forward, K/transform/Sequence.java:56-127) ----
  std::vector<int32_t> h_copy(B), h_mask(B), h_applied, h_skip(B, 0xFF), h_threw(B, 0);
  for (int b = 0; b < B; b++) h_copy[b] = (lengths[b] <= 15) ? 1 : 0;          // CompressedOutputStream.java:764-767
  if (ctx->skipBlocks) {                                                         // :769-788
    rc = kz_skip_block_flags(ctx, bt, P.d_applied);
    if (rc) return rc;
    KZ_HIP(hipMemcpyAsync(ctx->hpin, P.d_applied, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    KZ_HIP(kz_stream_sync(ctx, st));
    for (int b = 0; b < B; b++) if (ctx->hpin[b]) h_copy[b] = 1;
  }
  if (hp > 0) {
    // host stages (TEXT, UTF): per block on host threads (ahead of this call when `pre` is given); a block a stage applied to is
    // replaced in HBM, its skip flag bits and its "dataType" entry (the writer's Magic tag, then whatever TEXT / UTF leave
    // there) go on to the GPU stages
    hipEvent_t e0; kz_stage_begin(ctx, &e0);
    HostPre mine;
    // TEXT forward on the device (kz_text_fwd_gpu.hip) for the blocks it keeps after its statistics and finishes: they are rewritten
    // in their slots (hashes and Magic tags were taken from the original bytes above, on the same stream).  The host stages run on
    // the blocks it does not keep WHILE its walk runs, then on the few it kept and could not finish.
    std::vector<int32_t> gpuDone(B, 0), textDeclined(B, -1), noHost(B, 0), utfDone(B, 0);
    std::vector<int> listOf(B, -1), listIdx(B, -1);                               // block -> (host pass, index in that pass's list)
    HostPre passP[2];
    if (textFwdGpu) {
      std::unique_ptr<TextFwdJob, void (*)(TextFwdJob*)> J(kz_text_fwd_gpu_new(), kz_text_fwd_gpu_free);
      std::vector<int32_t> take(B), keeps;
      for (int b = 0; b < B; b++) take[b] = (!h_copy[b] && lengths[b] > 0) ? 1 : 0;
      rc = kz_text_fwd_gpu_classify(ctx, bt, blockSize, take, keeps, textDeclined, *J);
      if (rc) return rc;
      // a block that is not text: TEXT declined it on the device and left its "dataType"; UTF (if the chain has it) looks at UNDEFINED
      // and UTF8 blocks only (UTFCodec.java:93-101), on the host, from that entry on; every other block is done with the host stages
      for (int b = 0; b < B; b++) if (textDeclined[b] >= 0 && (hp == 1 || (textDeclined[b] != KZ_DT_UNDEFINED && textDeclined[b] != KZ_DT_UTF8))) noHost[b] = 1;
      // UTF forward on the device for the blocks TEXT declined as UTF-8 (kz_utf_fwd_gpu.hip; KZ_UTF_FWD_GPU=0: host stage): what it
      // finishes or declines by the reference's rules needs no host stage
      if (hp == 2 && ctx->sw.utfFwdGpu != 0) {
        std::vector<int32_t> takeU(B, 0);
        int nU = 0;
        for (int b = 0; b < B; b++) if (take[b] && textDeclined[b] == KZ_DT_UTF8) { takeU[b] = 1; nU++; }
        if (nU) { rc = kz_utf_fwd_gpu(ctx, bt, takeU, utfDone); if (rc) return rc; }
        for (int b = 0; b < B; b++) if (utfDone[b]) noHost[b] = 1;
      }
      kz_ctx::Stage& stg = ctx->tfIn;                                              // pinned staging for device input (not hsIn: ADVICE r5)
      auto host_pass = [&](int pass, const std::vector<int>& blocks, bool overlap) -> int {
        std::vector<const uint8_t*> ptrs(blocks.size());
        std::vector<int32_t> lens(blocks.size()), firstStage(blocks.size(), 0), dtIn(blocks.size(), KZ_DT_UNDEFINED);
        if (!host && !blocks.empty()) {
          int r2 = kz_stage_reserve(ctx, stg, (size_t)blocks.size() * (size_t)maxN + 64, true);
          if (r2) return r2;
        }
        // the copies of a pass that runs beside the walk go on their own stream (the caller's input is not this context's to order:
        // k_copy_bytes read it on the main stream without waiting either), the walk is queued on the main one meanwhile
        if (overlap && !host && !ctx->tfCopy) KZ_HIP(hipStreamCreateWithFlags(&ctx->tfCopy, hipStreamNonBlocking));
        hipStream_t cs = (overlap && !host) ? ctx->tfCopy : st;
        for (size_t k = 0; k < blocks.size(); k++) {
          const int b = blocks[k];
          lens[k] = lengths[b];
          listOf[b] = pass; listIdx[b] = (int)k;
          if (textDeclined[b] >= 0) { firstStage[k] = 1; dtIn[k] = textDeclined[b]; }    // TEXT is done with it
          if (host) ptrs[k] = in + (int64_t)b * inStride;
          else {
            ptrs[k] = stg.p + k * (size_t)maxN;
            KZ_HIP(hipMemcpyAsync(stg.p + k * (size_t)maxN, in + (int64_t)b * inStride, (size_t)lengths[b], hipMemcpyDeviceToHost, cs));
          }
        }
        if (overlap) {
          int r2 = kz_text_fwd_gpu_launch(ctx, bt, *J);
          if (r2) return r2;
          if (!host) KZ_HIP(hipStreamSynchronize(cs));
        } else KZ_HIP(kz_stream_sync(ctx, st));
        uint8_t* store = nullptr;                                                  // pinned slots for the stages' outputs, when there is room for them
        if (!blocks.empty() && kz_stage_reserve(ctx, ctx->tfOut[pass], (size_t)host_pre_slot(maxLen) * blocks.size() + 64, true) == 0) store = ctx->tfOut[pass].p;
        const auto t0 = std::chrono::steady_clock::now();
        host_prestage_list(types, hp, (int)entropyType, blockSize, maxLen, ptrs, lens, passP[pass], store, &firstStage, &dtIn);
        if (ctx->sw.textGpuTrace)
          fprintf(stderr, "[textfwd] host pass %d: %d blocks, %.0f ms\n", pass, (int)blocks.size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        return 0;
      };
      std::vector<int> first, second;
      for (int b = 0; b < B; b++) if (take[b] && !keeps[b] && !noHost[b]) first.push_back(b);
      rc = host_pass(0, first, true);
      if (rc) return rc;
      rc = kz_text_fwd_gpu_finish(ctx, bt, *J, gpuDone);
      if (rc) return rc;
      for (int b = 0; b < B; b++) if (take[b] && keeps[b] && !gpuDone[b]) second.push_back(b);
      if (!second.empty()) { rc = host_pass(1, second, false); if (rc) return rc; }
    } else if (!pre) {
      std::unique_ptr<uint8_t[]> hostCopy;
      const uint8_t* hsrc = in;
      int64_t hstride = inStride;
      if (!host) {
        hostCopy.reset(new uint8_t[(size_t)B * (size_t)maxN + 64]);
        for (int b = 0; b < B; b++)
          if (lengths[b]) KZ_HIP(hipMemcpyAsync(hostCopy.get() + (size_t)b * maxN, in + (int64_t)b * inStride, (size_t)lengths[b], hipMemcpyDeviceToHost, st));
        hsrc = hostCopy.get(); hstride = maxN;
        KZ_HIP(kz_stream_sync(ctx, st));
      }
      host_prestage(types, hp, (int)entropyType, blockSize, maxLen, hsrc, hstride, lengths, h_copy.data(), B, mine);
      pre = &mine;
    }
    KZ_HIP(kz_stream_sync(ctx, st));                                            // the blocks are in HBM, hashed and tagged: slots may be rewritten
    std::vector<int32_t> dts(B, KZ_DT_UNDEFINED);
    for (int b = 0; b < B; b++) {
      h_mask[b] = 0;
      if (h_copy[b]) { bt.h_len[b] = lengths[b]; continue; }                     // (a pre-staged block of <= 15 bytes was left alone as well)
      if (utfDone[b] == 1) { h_skip[b] = 0xFF & ~(1 << 6); dts[b] = KZ_DT_UTF8; continue; }   // UTF applied on the device (length set there); TEXT declined it
      if (gpuDone[b]) { h_skip[b] = 0xFF & ~(1 << 7); dts[b] = KZ_DT_TEXT; continue; }   // TEXT applied on the device (length set there: TextCodec.java:667); UTF declines a block tagged TEXT (UTFCodec.java:93-101)
      const HostPre* q = pre;
      int k = b;
      if (textFwdGpu) {
        if (listOf[b] < 0) { bt.h_len[b] = lengths[b]; if (noHost[b]) dts[b] = textDeclined[b]; continue; }   // (noHost: every host stage declined, TEXT's entry stays)
        q = &passP[listOf[b]]; k = listIdx[b];
      }
      h_skip[b] = q->skip[k];
      dts[b] = q->dtype[k];
      bt.h_len[b] = q->outLen[k];
      if (q->changed[k] && q->outLen[k] > 0) {
        if (q->pinned) h_mask[b] = 1;
        else KZ_HIP(hipMemcpyAsync(bt.buf[0] + (int64_t)b * bt.stride, q->data(k), (size_t)q->outLen[k], hipMemcpyHostToDevice, st));
      }
    }
    KZ_HIP(hipMemcpyAsync(bt.d_len, bt.h_len.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    if (pre && pre->pinned) {                                                   // the stages' outputs sit in pinned slots: one gather kernel reads them in place
      KZ_HIP(hipMemcpyAsync(P.d_mask, h_mask.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
      KZ_LAUNCH(ctx, KID_COPY_BYTES, k_copy_bytes, dim3(64, B), dim3(256), pre->store, pre->slot, bt.buf[0], bt.stride, bt.d_len, (const int32_t*)nullptr, P.d_mask);
    }
    KZ_HIP(hipMemcpyAsync(bt.d_dtype, dts.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    KZ_HIP(kz_stream_sync(ctx, st));                                            // pageable sources; `mine` and `dts` are locals
    kz_stage_end(ctx, e0, KZ_STAGE_HOST_FWD, 0);
  }
  for (int i = hp; i < nb; i++) {
    for (int b = 0; b < B; b++) h_mask[b] = (!h_copy[b] && bt.h_len[b] > 0) ? 1 : 0;
    if (types[i] == KZ_T_NONE) { for (int b = 0; b < B; b++) if (h_mask[b]) h_skip[b] &= ~(1 << (7 - i)); continue; }
    hipEvent_t e0; kz_stage_begin(ctx, &e0);
    int64_t inBytes = 0; for (int b = 0; b < B; b++) if (h_mask[b]) inBytes += bt.h_len[b];
    const int type = types[i];
    rc = run_stage(ctx, P, h_mask, h_applied, [&](kz_batch& x) { return run_transform_stage(ctx, x, type, true, 0); });
    if (rc) return rc;
    kz_stage_end(ctx, e0, stage_id(type, true), inBytes);
    for (int b = 0; b < B; b++) {
      if (h_applied[b] > 0) h_skip[b] &= ~(1 << (7 - i));
      else if (h_applied[b] < 0) h_threw[b] = 1;                    // an exception in the reference: ERR_PROCESS_BLOCK for the block
    }
  }
  for (int b = 0; b < B; b++) if (h_copy[b]) h_skip[b] = 0x7F;                   // NullTransform applies (NONE_TYPE chain)
  KZ_HIP(hipMemcpyAsync(F.skipFlags, h_skip.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
  KZ_HIP(hipMemcpyAsync(F.isCopy, h_copy.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
  F.postLen = bt.d_len;
  KZ_LAUNCH(ctx, KID_FRAME_PREPARE, k_frame_prepare, dim3((B + 255) / 256), dim3(256), F, B);

  // ---- entropy (EntropyEncoder.encode + dispose) ----
  {
    hipEvent_t e0; kz_stage_begin(ctx, &e0);
    int64_t inBytes = 0; for (int b = 0; b < B; b++) inBytes += bt.h_len[b];
    // copy blocks and NONE entropy: raw bytes (NullEntropyEncoder.java:66-81)
    if (entropyType == KZ_E_ANS0 || entropyType == KZ_E_HUFFMAN || entropyType == KZ_E_FPAQ) {
      // small copy blocks use NONE: mask them out of the ANS stage by zero length, then copy raw
      for (int b = 0; b < B; b++) h_mask[b] = h_copy[b] ? 0 : 1;
      KZ_HIP(hipMemcpyAsync(P.d_mask, h_mask.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
      KZ_HIP(hipMemcpyAsync(P.d_lenSave, bt.d_len, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
      KZ_LAUNCH(ctx, KID_MASK_LEN, k_mask_len, dim3((B + 255) / 256), dim3(256), P.d_lenSave, P.d_mask, bt.d_len, B);
      std::vector<int32_t> saved = bt.h_len;
      for (int b = 0; b < B; b++) if (h_copy[b]) bt.h_len[b] = 0;
      rc = (entropyType == KZ_E_ANS0) ? kz_stage_ans0_encode(ctx, bt, d_out, outStride, F.hdrBytes, F.bits)
         : (entropyType == KZ_E_HUFFMAN) ? kz_stage_huffman_encode(ctx, bt, d_out, outStride, F.hdrBytes, F.bits)
                                         : kz_stage_fpaq_encode(ctx, bt, d_out, outStride, F.hdrBytes, F.bits);
      if (rc) return rc;
      bt.h_len = saved;
      KZ_HIP(hipMemcpyAsync(bt.d_len, P.d_lenSave, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    }
    // raw payload for copy blocks (and all blocks when entropy is NONE)
    {
      for (int b = 0; b < B; b++) h_mask[b] = (entropyType == KZ_E_NONE || h_copy[b]) ? 1 : 0;
      KZ_HIP(hipMemcpyAsync(P.d_mask, h_mask.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
      KZ_LAUNCH(ctx, KID_COPY_BYTES, k_copy_bytes, dim3(64, B), dim3(256), bt.buf[bt.cur], bt.stride, d_out, outStride, bt.d_len, F.hdrBytes, P.d_mask);
      // bits = 8*len for those blocks
      std::vector<int64_t> hb(B);
      if (entropyType == KZ_E_NONE) {
        for (int b = 0; b < B; b++) hb[b] = 8LL * bt.h_len[b];
        KZ_HIP(hipMemcpyAsync(F.bits, hb.data(), (size_t)B * 8, hipMemcpyHostToDevice, st));
        KZ_HIP(kz_stream_sync(ctx, st));
      } else {
        for (int b = 0; b < B; b++) if (h_copy[b]) { hb[b] = 8LL * bt.h_len[b]; KZ_HIP(hipMemcpyAsync(F.bits + b, &hb[b], 8, hipMemcpyHostToDevice, st)); }
        KZ_HIP(kz_stream_sync(ctx, st));                                          // hb is a local
      }
    }
    kz_stage_end(ctx, e0, KZ_STAGE_ENTROPY_ENC, inBytes);
  }
  // ---- block header, raw fallback (CompressedOutputStream.java:861-985) ----
  {
    hipEvent_t e0; kz_stage_begin(ctx, &e0);
    KZ_LAUNCH(ctx, KID_FRAME_DECIDE, k_frame_decide, dim3((B + 255) / 256), dim3(256), F, B);
    KZ_LAUNCH(ctx, KID_COPY_BYTES, k_copy_bytes, dim3(64, B), dim3(256), bt.buf[bt.cur], bt.stride, d_out, outStride, bt.d_len, F.hdrBytes, F.fallback);
    KZ_LAUNCH(ctx, KID_FRAME_HEADER, k_frame_header, dim3((B + 255) / 256), dim3(256), F, d_out, outStride, d_res, B);
    kz_stage_end(ctx, e0, KZ_STAGE_FRAME_ENC, 0);
  }
  KZ_HIP(hipMemcpyAsync(results, d_res, (size_t)B * sizeof(kz_block_result), hipMemcpyDeviceToHost, st));
  KZ_HIP(kz_stream_sync(ctx, st));
  for (int b = 0; b < B; b++) if (h_threw[b]) { results[b].status = -KZ_ERR_PROCESS_BLOCK; results[b].bits = 0; results[b].length = 0; }
  if (host) {
    for (int b = 0; b < B; b++) {
      const size_t nbytes = (size_t)((results[b].bits + 7) >> 3);
      if (nbytes) KZ_HIP(hipMemcpyAsync(out + (int64_t)b * outStride, d_out + (int64_t)b * outStride, nbytes, hipMemcpyDeviceToHost, st));
    }
    KZ_HIP(kz_stream_sync(ctx, st));
  }
  KZ_HIP(hipGetLastError());
  kz_ktimer_flush(ctx);
  return 0;
}

// TEXT sizes its hash map by the stream's block size and the decoder is told that size explicitly: an encoder that silently
// used the context's default would write blocks a decoder of another block size cannot read.  Chains with TEXT therefore need
// kz_ctx_set_block_size first (kz_compress and the Python / Java bindings do it).
static int32_t encode_block_size(kz_ctx* ctx, uint64_t transformType) {
  if (!ctx) return -KZ_ERR_INVALID_PARAM;
  int types[8];
  const int nb = split_types(transformType, types);
  for (int i = 0; i < nb; i++)
    if (types[i] == KZ_T_TEXT && !ctx->blockSizeSet) {
      snprintf(ctx->err, sizeof(ctx->err), "chains with TEXT need the stream's block size: call kz_ctx_set_block_size first");
      return -KZ_ERR_MISSING_PARAM;
    }
  return ctx->blockSize;
}
extern "C" int32_t kz_encode_blocks(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType,
                                    const uint8_t* in, int64_t inStride, const int32_t* lengths, int32_t nBlocks,
                                    uint8_t* out, int64_t outStride, kz_block_result* results, int32_t memKind) {
  const int32_t bsz = encode_block_size(ctx, transformType);
  if (bsz < 0) return bsz;
  return encode_blocks_bs(ctx, transformType, entropyType, bsz, in, inStride, lengths, nBlocks, out, outStride, results, memKind);
}

// =================================================================================================
// decode
struct FrameDec {
  int32_t* preLen; int32_t* skipFlags; int32_t* hdrBytes; int32_t* raw; int32_t* tcopy; int32_t* status;
  int64_t* bitOff; int64_t* bitEnd;
  u64* hashRef;          // [B] checksum stored in the block header
  int chk;
};
// CompressedInputStream.java:1025-1095 readBlockHeader
__global__ void k_frame_parse(const u8* __restrict__ in, int64_t inStride, const int64_t* __restrict__ bitLen, FrameDec F,
                              int nbFunctions, int maxTransformLength, int entropyNone, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const u8* p = in + (int64_t)b * inStride;
  const int64_t W = bitLen[b];
  int status = 0, preLen = 0, hdrBytes = 0, raw = 0, tcopy = 0;
  u32 skipFlags = 0;
  if (W == 0) { status = 0; }
  else if (W < 8) status = -KZ_ERR_BLOCK_SIZE;
  else {
    const u32 mode = p[0];
    bool hasSkip = false;
    if (mode & 0x80) {
      if (mode & 0x10) { tcopy = 1; if (nbFunctions > 4) hasSkip = true; else skipFlags = ((mode << 4) | 0x0F) & 0xFF; }
      else raw = 1;
    } else if (mode & 0x10) hasSkip = true;
    else skipFlags = ((mode << 4) | 0x0F) & 0xFF;
    const int dataSize = 1 + ((mode >> 5) & 3);
    hdrBytes = 1 + (hasSkip ? 1 : 0) + dataSize + 1;
    if (W < (int64_t)hdrBytes * 8) status = -KZ_ERR_BLOCK_SIZE;
    else {
      int idx = 1;
      if (hasSkip) skipFlags = p[idx++];
      for (int i = 0; i < dataSize; i++) preLen = (preLen << 8) | p[idx++];
      const u8 ck = p[idx++];
      const int cbytes = (F.chk == 1) ? 4 : (F.chk == 2 ? 8 : 0);
      if (cbytes && W >= (int64_t)(hdrBytes + cbytes) * 8) { u64 hv = 0; for (int k = 0; k < cbytes; k++) hv = (hv << 8) | p[idx++]; F.hashRef[b] = hv; }
      hdrBytes += cbytes;
      if (ck != kz_block_cksum(mode, skipFlags, (u32)preLen, (u64)W)) status = -KZ_ERR_CRC_CHECK;
      else if (W < (int64_t)hdrBytes * 8) status = -KZ_ERR_BLOCK_SIZE;
      else if (preLen < 0 || preLen > maxTransformLength) status = -KZ_ERR_READ_FILE;
      else if (((W + 7) >> 3) > (int64_t)preLen + hdrBytes) status = -KZ_ERR_BLOCK_SIZE;     // :1158-1164
      // raw bytes (copy block, "transformed copy", NONE entropy) are read with NullEntropyDecoder: a stream shorter than
      // the stated length makes the bit stream throw -> ERR_PROCESS_BLOCK (CompressedInputStream.java:1305-1316,1366-1371)
      else if (preLen > 0 && (raw || tcopy || entropyNone) && W < 8LL * ((int64_t)hdrBytes + preLen)) status = -KZ_ERR_PROCESS_BLOCK;
    }
  }
  if (status) { preLen = 0; }
  F.preLen[b] = preLen; F.skipFlags[b] = (int32_t)(raw ? 0xFF : skipFlags); F.hdrBytes[b] = hdrBytes;
  F.raw[b] = raw; F.tcopy[b] = tcopy; F.status[b] = status;
  F.bitOff[b] = 8LL * hdrBytes; F.bitEnd[b] = W;
}
__global__ void k_copy_payload(const u8* __restrict__ in, int64_t inStride, u8* __restrict__ dst, int64_t stride,
                               const int32_t* __restrict__ len, const int32_t* __restrict__ hdrBytes, const int32_t* __restrict__ cond) {
  const int b = blockIdx.y;
  if (!cond[b]) return;
  const int n = len[b];
  const u8* s = in + (int64_t)b * inStride + hdrBytes[b];
  u8* d = dst + (int64_t)b * stride;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d[i] = s[i];
}

// deferHost: leave the host stages (TEXT / UTF inverse) to the caller: the blocks come out as the GPU chain left them, results[b]
// carry their skip flags and lengths (kz_decode_blocks runs those stages on a helper thread under the next chunk's GPU work).
static int32_t decode_blocks_impl(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType, int32_t blockSize,
                                  const uint8_t* in, int64_t inStride, const int64_t* bitLengths, int32_t nBlocks,
                                  uint8_t* out, int64_t outStride, kz_block_result* results, int32_t memKind, bool deferHost = false) {
  if (!ctx) return -KZ_ERR_INVALID_PARAM;
  if (nBlocks <= 0) return 0;
  KZ_HIP(hipSetDevice(ctx->device));
  int types[8];
  const int nb = split_types(transformType, types);
  for (int i = 0; i < nb; i++) if (!transform_supported(types[i])) { snprintf(ctx->err, sizeof(ctx->err), "unsupported transform id %d", types[i]); return -KZ_ERR_INVALID_CODEC; }
  if (!entropy_supported((int)entropyType)) { snprintf(ctx->err, sizeof(ctx->err), "unsupported entropy id %u", entropyType); return -KZ_ERR_INVALID_CODEC; }
  const int B = nBlocks;
  const int hp = host_prefix(ctx, types, nb);
  if (hp < 0) return hp;
  const bool host = memKind == KZ_MEM_HOST;
  // decoder's working size: blockSize + max(512, blockSize/16) (CompressedInputStream.java:694-695)
  const int dataCap = blockSize + std::max(512, blockSize >> 4);
  const int maxTL = std::min(std::max(blockSize + blockSize / 2, 2048), 1 << 30);
  int64_t maxInBytes = 0;
  for (int b = 0; b < B; b++) maxInBytes = std::max(maxInBytes, (bitLengths[b] + 7) >> 3);
  // The reference accepts preTransformLength up to 1.5 x blockSize (CompressedInputStream.java:1139-1140);
  // no chain in scope expands a block by more than 1024 bytes, so buffers are sized dataCap+1024 and a
  // larger (corrupt) length is rejected with the same error code.
  const int maxLen = std::min(maxTL, dataCap + 1024);
  ChainSpec CS; CS.nb = nb; CS.entropy = (int)entropyType; for (int i = 0; i < nb; i++) CS.types[i] = types[i];
  const bool textGpu = hp > 0 && !deferHost && types[0] == KZ_T_TEXT && text_gpu_on(ctx, (uint32_t)entropyType, B);
  const int iu = (hp > 0 && types[hp - 1] == KZ_T_UTF) ? hp - 1 : -1;      // UTF as the host stage undone first
  const bool utfGpu = iu >= 0 && !deferHost && utf_gpu_on(ctx);
  {
    const size_t perBlock = pipeline_scratch(1, maxLen, true, CS) + (size_t)maxLen * 2 + (size_t)(host ? maxInBytes : 0) + (1 << 16) +
                            (textGpu ? kz_text_gpu_scratch_per_block(blockSize) : 0) + (utfGpu ? kz_utf_gpu_scratch_per_block(maxLen) : 0);
    int maxB = (int)std::min<size_t>(KZ_MAX_BATCH, std::max<size_t>(1, kz_arena_budget() / perBlock));   // grid.y carries the block index
    // the serial-per-block inverse stages like whole multiples of 8 blocks per CU (one wave per block, two per SIMD)
    const int unit = 8 * (ctx->numCUs > 0 ? ctx->numCUs : 256);
    if (B > maxB && maxB > unit) maxB = (maxB / unit) * unit;       // only when the batch has to be split anyway
    if (B > maxB) {
      for (int b0 = 0; b0 < B; b0 += maxB) {
        const int cnt = std::min(maxB, B - b0);
        int rc = decode_blocks_impl(ctx, transformType, entropyType, blockSize, in + (int64_t)b0 * inStride, inStride, bitLengths + b0, cnt,
                                    out + (int64_t)b0 * outStride, outStride, results + b0, memKind, deferHost);
        if (rc) return rc;
      }
      return 0;
    }
  }
  Pipe P;
  const int64_t inS = host ? (int64_t)kz_align((size_t)maxInBytes + 64, 256) : inStride;
  // (the expensive-blocks-first schedule runs the entropy and ZRLT stages once per group: a second set of their small scratch)
  const int64_t extra = (host ? inS * B : 0) + (int64_t)B * (sizeof(kz_block_result) + 128) + (int64_t)B * 32 + 8192 +
                        (int64_t)kz_zrlt_scratch(B, maxLen) + (int64_t)B * ((int64_t)(maxLen / 16384 + 4) * 8 + 64) + 65536 + (int64_t)B * 64 +
                        (textGpu ? (int64_t)B * (int64_t)kz_text_gpu_scratch_per_block(blockSize) + (int64_t)B * 16 + (1 << 16) : 0) +
                        (utfGpu ? (int64_t)B * (int64_t)kz_utf_gpu_scratch_per_block(maxLen) + (int64_t)B * 16 + (1 << 16) : 0);
  int rc = pipe_setup(ctx, P, B, maxLen, extra, true, CS);
  if (rc) return rc;
  kz_batch& bt = P.bt;
  hipStream_t st = ctx->stream;
  const u8* d_in = in;
  if (host) {
    u8* t = (u8*)kz_arena_alloc(ctx, (size_t)inS * B);
    if (!t) { snprintf(ctx->err, sizeof(ctx->err), "decode: arena overflow"); return -KZ_ERR_DEVICE; }
    for (int b = 0; b < B; b++) {
      const size_t nbytes = (size_t)((bitLengths[b] + 7) >> 3);
      if (nbytes) KZ_HIP(hipMemcpyAsync(t + (int64_t)b * inS, in + (int64_t)b * inStride, nbytes, hipMemcpyHostToDevice, st));
    }
    d_in = t;
  }
  FrameDec F;
  F.preLen = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  F.skipFlags = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  F.hdrBytes = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  F.raw = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  F.tcopy = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  F.status = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  F.bitOff = (int64_t*)kz_arena_alloc(ctx, (size_t)B * 8);
  F.bitEnd = (int64_t*)kz_arena_alloc(ctx, (size_t)B * 8);
  F.hashRef = (u64*)kz_arena_alloc(ctx, (size_t)B * 8);
  F.chk = ctx->checksum;
  u64* d_hashOut = (u64*)kz_arena_alloc(ctx, (size_t)B * 8);
  int64_t* d_bitLen = (int64_t*)kz_arena_alloc(ctx, (size_t)B * 8);
  if (!d_bitLen) { snprintf(ctx->err, sizeof(ctx->err), "decode: arena overflow"); return -KZ_ERR_DEVICE; }
  KZ_HIP(hipMemcpyAsync(d_bitLen, bitLengths, (size_t)B * 8, hipMemcpyHostToDevice, st));
  hipEvent_t e0; kz_stage_begin(ctx, &e0);
  KZ_LAUNCH(ctx, KID_FRAME_PARSE, k_frame_parse, dim3((B + 255) / 256), dim3(256), d_in, inS, d_bitLen, F, nb, maxLen, (entropyType == KZ_E_NONE) ? 1 : 0, B);
  // read back descriptors
  int32_t* hpb = ctx->hpin + 4 * B;
  KZ_HIP(hipMemcpyAsync(hpb + 0 * B, F.preLen, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  KZ_HIP(hipMemcpyAsync(hpb + 1 * B, F.skipFlags, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  KZ_HIP(hipMemcpyAsync(hpb + 2 * B, F.raw, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  KZ_HIP(hipMemcpyAsync(hpb + 3 * B, F.tcopy, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  KZ_HIP(hipMemcpyAsync(hpb + 4 * B, F.status, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  KZ_HIP(kz_stream_sync(ctx, st));
  kz_stage_end(ctx, e0, KZ_STAGE_FRAME_DEC, 0);
  std::vector<int32_t> h_pre(hpb, hpb + B), h_skip(hpb + B, hpb + 2 * B), h_raw(hpb + 2 * B, hpb + 3 * B), h_tc(hpb + 3 * B, hpb + 4 * B), h_status(hpb + 4 * B, hpb + 5 * B);
  for (int b = 0; b < B; b++) bt.h_len[b] = h_status[b] ? 0 : h_pre[b];
  KZ_HIP(hipMemcpyAsync(bt.d_len, bt.h_len.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));

  // ---- entropy decode into buf ----
  std::vector<int32_t> h_mask(B), h_applied;
  // one pass of the entropy stage over the blocks of `part` (all blocks when null)
  auto entropy_pass = [&](const std::vector<int32_t>* part, const int32_t* d_part) -> int {
    hipEvent_t e1; kz_stage_begin(ctx, &e1);
    int64_t outBytes = 0;
    std::vector<int32_t> h_rawp(B);
    for (int b = 0; b < B; b++) {
      const bool in = !part || (*part)[b];
      if (in) outBytes += bt.h_len[b];
      h_rawp[b] = (in && (entropyType == KZ_E_NONE || h_raw[b] || h_tc[b])) ? 1 : 0;
    }
    if (entropyType == KZ_E_ANS0 || entropyType == KZ_E_HUFFMAN || entropyType == KZ_E_FPAQ) {
      for (int b = 0; b < B; b++) h_mask[b] = ((!part || (*part)[b]) && !(h_raw[b] || h_tc[b])) ? 1 : 0;
      int r = run_stage(ctx, P, h_mask, h_applied, [&](kz_batch& x) {
        return (entropyType == KZ_E_ANS0) ? kz_stage_ans0_decode(ctx, x, d_in, inS, F.bitOff, F.bitEnd)
             : (entropyType == KZ_E_HUFFMAN) ? kz_stage_huffman_decode(ctx, x, d_in, inS, F.bitOff, F.bitEnd)
                                             : kz_stage_fpaq_decode(ctx, x, d_in, inS, F.bitOff, F.bitEnd); }, d_part);
      if (r) return r;
      for (int b = 0; b < B; b++) if (h_mask[b] && !h_applied[b] && !h_status[b]) h_status[b] = -KZ_ERR_PROCESS_BLOCK;
    } else {
      bt.cur ^= 1;    // raw path writes into the "next" buffer below
    }
    KZ_HIP(hipMemcpyAsync(P.d_mask, h_rawp.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    KZ_LAUNCH(ctx, KID_COPY_PAYLOAD, k_copy_payload, dim3(64, B), dim3(256), d_in, inS, bt.buf[bt.cur], bt.stride, bt.d_len, F.hdrBytes, P.d_mask);
    KZ_HIP(kz_stream_sync(ctx, st));                               // h_rawp is a local
    kz_stage_end(ctx, e1, KZ_STAGE_ENTROPY_DEC, outBytes);
    return 0;
  };
  // one inverse transform stage over the blocks of `part`
  auto transform_pass = [&](int i, const std::vector<int32_t>* part, const int32_t* d_part) -> int {
    const int type = types[i];
    bool any = false;
    for (int b = 0; b < B; b++) {
      h_mask[b] = ((!part || (*part)[b]) && !h_status[b] && !(h_skip[b] & (1 << (7 - i))) && bt.h_len[b] > 0) ? 1 : 0;
      any |= h_mask[b] != 0;
    }
    if (!any && !part) return 0;
    hipEvent_t e1; kz_stage_begin(ctx, &e1);
    int r = run_stage(ctx, P, h_mask, h_applied, [&](kz_batch& x) { return run_transform_stage(ctx, x, type, false, dataCap); }, d_part);
    if (r) return r;
    int64_t outBytes = 0; for (int b = 0; b < B; b++) if (h_mask[b]) outBytes += bt.h_len[b];
    kz_stage_end(ctx, e1, stage_id(type, false), outBytes);
    for (int b = 0; b < B; b++) if (h_mask[b] && !h_applied[b]) { h_status[b] = -KZ_ERR_PROCESS_BLOCK; bt.h_len[b] = 0; }
    return 0;
  };
  // ---- "expensive blocks first" for BWT+RANK+ZRLT (or MTFT) chains on large batches: the RANK inverse of a block is serial and
  //      its cost (the ZRLT-coded length = the decoded length, known from the block header) differs by an order of magnitude
  //      between blocks.  The expensive blocks go through entropy decoding and ZRLT inverse FIRST and start their RANK inverse on
  //      the side stream; the others follow on the main stream underneath it (overlap_* above). ----
  bool chainDone = false;
  if (nb - hp == 3 && types[hp] == KZ_T_BWT && (types[hp + 1] == KZ_T_RANK || types[hp + 1] == KZ_T_MTFT) && types[hp + 2] == KZ_T_ZRLT &&
      (entropyType == KZ_E_ANS0 || entropyType == KZ_E_HUFFMAN) && !ctx->sw.noSFirst) {
    bool all = true;
    const int need = (1 << (7 - hp)) | (1 << (7 - (hp + 1)));       // BWT and RANK applied to every block
    for (int b = 0; b < B && all; b++) all = !h_status[b] && bt.h_len[b] > 0 && !(h_skip[b] & need);   // (copy blocks carry all skip bits)
    Overlap O;
    std::vector<int32_t> ones(B, 1);
    if (ctx->sw.traceSched) {
      int nst = 0, ntc = 0, nz = 0, nsk = 0, nraw = 0;
      for (int b = 0; b < B; b++) { nst += h_status[b] != 0; ntc += h_tc[b] != 0; nz += bt.h_len[b] <= 0; nsk += (h_skip[b] & need) != 0; nraw += h_raw[b] != 0; }
      fprintf(stderr, "[sched] B=%d all=%d status=%d tcopy=%d empty=%d skipBwtRank=%d raw=%d\n", B, (int)all, nst, ntc, nz, nsk, nraw);
    }
    // cost of a block's RANK inverse: its COMPRESSED size.  The time per rank grows with the rank (positions shifted), and so
    // does the entropy coder's output; the ZRLT-coded length alone puts skewed and uniform data in one class (68 vs 134 ns / rank)
    std::vector<int32_t> cost(B);
    for (int b = 0; b < B; b++) cost[b] = (int32_t)std::min<int64_t>((bitLengths[b] + 7) >> 3, 0x7FFFFFFF);
    if (all && overlap_classify(ctx, B, ones, cost, O)) {
      const int mode = types[hp + 1] == KZ_T_RANK ? 2 : 1;
      rc = overlap_alloc(ctx, B, O);
      if (rc) return rc;
      bt.h_cost = cost;
      {
        std::vector<int32_t> rawp(B);
        for (int b = 0; b < B; b++) rawp[b] = (h_raw[b] || h_tc[b]) ? 1 : 0;
        overlap_plan(ctx, O, bt.h_len, rawp, blockSize, true);
      }
      for (size_t k = 0; k < O.launchOrder.size(); k++) {
        const int g = O.launchOrder[k];
        rc = entropy_pass(&O.groups[g].in, O.groups[g].d_in);
        if (!rc) rc = transform_pass(hp + 2, &O.groups[g].in, O.groups[g].d_in);
        if (!rc) rc = overlap_start_rank(ctx, bt, mode, O, g, (int)k);
        if (rc) return rc;
      }
      rc = overlap_finish(ctx, P, O, h_applied);
      if (rc) return rc;
      for (int b = 0; b < B; b++) if (!h_status[b] && !h_applied[b]) { h_status[b] = -KZ_ERR_PROCESS_BLOCK; bt.h_len[b] = 0; }
      chainDone = true;
    }
  }
  if (!chainDone) {
  rc = entropy_pass(nullptr, nullptr);
  if (rc) return rc;
  // ---- inverse chain (Sequence.inverse, K/transform/Sequence.java:137-207) ----
  // cost hint for the serial-per-block inverse stages: the length a block had at the input of the previous stage
  // (for RANK behind ZRLT that is the ZRLT-coded length ~ the number of non-zero ranks)
  std::vector<int32_t> prevIn(B);
  for (int b = 0; b < B; b++) prevIn[b] = (int32_t)std::min<int64_t>((bitLengths[b] + 7) >> 3, 0x7FFFFFFF);
  for (int i = nb - 1; i >= hp; i--) {
    if (types[i] == KZ_T_NONE) continue;
    bt.h_cost = prevIn;
    prevIn = bt.h_len;
    bool any = false;
    for (int b = 0; b < B; b++) { h_mask[b] = (!h_status[b] && !(h_skip[b] & (1 << (7 - i))) && bt.h_len[b] > 0) ? 1 : 0; any |= h_mask[b] != 0; }
    if (!any) continue;
    const int type = types[i];
    if ((type == KZ_T_RANK || type == KZ_T_MTFT) && i - 1 >= hp && types[i - 1] == KZ_T_BWT) {
      // RANK / MTFT inverse followed by the BWT inverse on the same blocks: overlapped schedule for large batches
      bool same = true;
      for (int b = 0; b < B && same; b++) same = ((h_skip[b] >> (7 - i)) & 1) == ((h_skip[b] >> (8 - i)) & 1);
      if (same) {
        const int fr = overlapped_rank_bwt_inverse(ctx, P, type == KZ_T_RANK ? 2 : 1, h_mask, bt.h_cost, h_applied);
        if (fr < 0) return fr;
        if (fr > 0) {
          for (int b = 0; b < B; b++) if (h_mask[b] && !h_applied[b]) { h_status[b] = -KZ_ERR_PROCESS_BLOCK; bt.h_len[b] = 0; }
          prevIn = bt.h_len;
          i--;                                                        // the BWT stage is done as well
          continue;
        }
      }
    }
    rc = transform_pass(i, nullptr, nullptr);
    if (rc) return rc;
  }
  }
  std::vector<int32_t> h_skipHost(h_skip);                         // the host stages' view: TEXT counts as skipped for the blocks the device took
  if (utfGpu) {
    // blocks that went through UTF (the host stage undone first): its inverse on the device
    std::vector<int32_t> take(B, 0), done;
    for (int b = 0; b < B; b++) take[b] = (!h_status[b] && bt.h_len[b] > 0 && !(h_skip[b] & (1 << (7 - iu)))) ? 1 : 0;
    hipEvent_t e0; kz_stage_begin(ctx, &e0);
    rc = kz_stage_utf_inverse_gpu(ctx, bt, dataCap, take, done);
    if (rc) return rc;
    kz_stage_end(ctx, e0, KZ_STAGE_HOST_INV, 0);
    for (int b = 0; b < B; b++) if (done[b]) h_skipHost[b] |= 1 << (7 - iu);
  }
  if (textGpu) {
    // blocks whose only host stage left is TEXT (every other host stage skipped for the block, or done above): its inverse on the device
    std::vector<int32_t> take(B, 0), done;
    for (int b = 0; b < B; b++) {
      bool t = !h_status[b] && bt.h_len[b] > 0 && !(h_skip[b] & 0x80);
      for (int i = 1; i < hp && t; i++) t = (h_skipHost[b] & (1 << (7 - i))) != 0;
      take[b] = t ? 1 : 0;
    }
    hipEvent_t e0; kz_stage_begin(ctx, &e0);
    rc = kz_stage_text_inverse_gpu(ctx, bt, blockSize, dataCap, entropyType == KZ_E_FPAQ, take, done, text_gpu_form(ctx, (uint32_t)entropyType, B));
    if (rc) return rc;
    kz_stage_end(ctx, e0, KZ_STAGE_HOST_INV, 0);
    for (int b = 0; b < B; b++) if (done[b]) h_skipHost[b] |= 0x80;
  }
  if (hp > 0 && !deferHost) {
    // host stages (UTF, TEXT inverse) behind the GPU stages: blocks that went through one come back to the host, are decoded
    // on host threads and return to their slot
    hipEvent_t e1; kz_stage_begin(ctx, &e1);
    KZ_HIP(kz_stream_sync(ctx, st));
    HostInv H;
    H.device = ctx->device; H.types = types; H.hp = hp; H.blockSize = blockSize; H.cap = dataCap;
    H.dbuf = bt.buf[bt.cur]; H.dstride = bt.stride; H.slotCap = bt.stride; H.hostMem = false;
    H.len = bt.h_len.data(); H.skip = h_skipHost.data(); H.status = h_status.data();
    kz_parallel_for(B, KZ_HOST_STAGE_THREADS, host_inverse_block, &H);
    if (H.fail) { snprintf(ctx->err, sizeof(ctx->err), "host stage: copy from / to the device failed"); return -KZ_ERR_DEVICE; }
    KZ_HIP(hipMemcpyAsync(bt.d_len, bt.h_len.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    KZ_HIP(kz_stream_sync(ctx, st));
    kz_stage_end(ctx, e1, KZ_STAGE_HOST_INV, 0);
  }
  // ---- checksum verification (CompressedInputStream.java:1349-1363) ----
  if (F.chk) {
    rc = kz_block_hashes(ctx, bt.buf[bt.cur], bt.stride, bt.d_len, B, F.chk, d_hashOut);
    if (rc) return rc;
    std::vector<u64> h1(B), h2(B);
    KZ_HIP(hipMemcpyAsync(h1.data(), F.hashRef, (size_t)B * 8, hipMemcpyDeviceToHost, st));
    KZ_HIP(hipMemcpyAsync(h2.data(), d_hashOut, (size_t)B * 8, hipMemcpyDeviceToHost, st));
    KZ_HIP(kz_stream_sync(ctx, st));
    for (int b = 0; b < B; b++) if (!h_status[b] && bt.h_len[b] > 0 && h1[b] != h2[b]) h_status[b] = -KZ_ERR_CRC_CHECK;
  }
  // ---- results ----
  for (int b = 0; b < B; b++) {
    if (!h_status[b] && (bt.h_len[b] > outStride || (!deferHost && bt.h_len[b] > blockSize))) h_status[b] = -KZ_ERR_PROCESS_BLOCK;
    results[b].bits = bitLengths[b]; results[b].length = h_status[b] ? 0 : bt.h_len[b]; results[b].status = h_status[b];
    results[b].skipFlags = (uint8_t)h_skip[b]; results[b].mode = 0;
    if (host && !h_status[b] && bt.h_len[b] > 0)
      KZ_HIP(hipMemcpyAsync(out + (int64_t)b * outStride, bt.buf[bt.cur] + (int64_t)b * bt.stride, (size_t)bt.h_len[b], hipMemcpyDeviceToHost, st));
  }
  if (!host) {                                                   // device output: one strided copy kernel (failed blocks: length 0)
    for (int b = 0; b < B; b++) ctx->hpin[b] = h_status[b] ? 0 : bt.h_len[b];
    KZ_HIP(hipMemcpyAsync(bt.d_len, ctx->hpin, (size_t)B * 4, hipMemcpyHostToDevice, st));
    KZ_LAUNCH(ctx, KID_COPY_BYTES, k_copy_bytes, dim3(64, B), dim3(256), bt.buf[bt.cur], bt.stride, out, outStride, bt.d_len, (const int32_t*)nullptr, (const int32_t*)nullptr);
  }
  KZ_HIP(kz_stream_sync(ctx, st));
  KZ_HIP(hipGetLastError());
  kz_ktimer_flush(ctx);
  return 0;
}

static void decode_error_sync(kz_ctx* ctx) {
  // an error exit may leave kernels of the overlapped schedule running on the side streams: the next call reuses (or frees) the
  // arena they work in
  for (int i = 0; i < 5; i++) if (ctx->side[i]) hipStreamSynchronize(ctx->side[i]);
  hipStreamSynchronize(ctx->stream);
}
extern "C" int32_t kz_decode_blocks(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType, int32_t blockSize,
                                    const uint8_t* in, int64_t inStride, const int64_t* bitLengths, int32_t nBlocks,
                                    uint8_t* out, int64_t outStride, kz_block_result* results, int32_t memKind) {
  if (!ctx) return -KZ_ERR_INVALID_PARAM;
  int types[8];
  const int nb = split_types(transformType, types);
  int hp = 0;
  while (hp < nb && kz_is_host_transform(types[hp])) hp++;
  const int CH = ctx->sw.hostChunkDec;
  // ---- chains led by TEXT / UTF on large batches: the host inverse stages of chunk k run on a helper thread (and the host pool)
  //      while the GPU decodes chunk k+1.  (Block checksums are verified on the device AFTER the host stages: that case takes the
  //      one-pass path.) ----
  if (hp > 0 && ctx->checksum == 0 && nBlocks >= 2 * CH && !(types[0] == KZ_T_TEXT && text_gpu_on(ctx, (uint32_t)entropyType, nBlocks))) {
    const int B = nBlocks, nch = (B + CH - 1) / CH;
    const int dataCap = blockSize + std::max(512, blockSize >> 4);
    std::vector<std::thread> finishers;
    std::vector<std::unique_ptr<HostInv>> jobs;
    std::vector<std::vector<int32_t>> lens(nch), skips(nch), stats(nch);
    int rc = 0;
    const bool trace = ctx->sw.tracePipe != 0;
    // KZ_HOST_INV_STAGED=1: the chunk's blocks go through pinned slots with one gather / scatter kernel per
    // sub-chunk instead of a copy pair per block.  Measured slower on the level-exact bench row (1 200 vs 1 066 ms per 2 048-block
    // decode), so it is opt-in.
    const bool staged = ctx->sw.hostInvStaged == 1 && memKind != KZ_MEM_HOST;
    const int64_t pslot = (int64_t)kz_align((size_t)std::max<int64_t>(outStride, dataCap) + 64, 64);
    if (staged) {                                                     // everything that can fail, before the first chunk is in flight
      for (int q = 0; q < 2 && !rc; q++) {
        rc = kz_stage_reserve(ctx, ctx->hsIn[q], (size_t)pslot * (size_t)CH + 64, true);
        if (!rc) rc = kz_stage_reserve(ctx, ctx->hiAux[q], (size_t)CH * 16 + 64, false);
        if (!rc && !ctx->hiStream[q] && hipStreamCreateWithFlags(&ctx->hiStream[q], hipStreamNonBlocking) != hipSuccess) rc = -KZ_ERR_DEVICE;
      }
      if (rc) return rc;
    }
    const auto tz = std::chrono::steady_clock::now();
    auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tz).count(); };
    for (int k = 0; k < nch && !rc; k++) {
      const int b0 = k * CH, cnt = std::min(CH, B - b0);
      const double tg = ms();
      rc = decode_blocks_impl(ctx, transformType, entropyType, blockSize, in + (int64_t)b0 * inStride, inStride, bitLengths + b0, cnt,
                              out + (int64_t)b0 * outStride, outStride, results + b0, memKind, true);
      if (trace) fprintf(stderr, "[pipe] decode chunk %d: gpu %.0f ms (at %.0f)\n", k, ms() - tg, ms());
      if (rc) break;
      lens[k].resize(cnt); skips[k].resize(cnt); stats[k].resize(cnt);
      for (int i = 0; i < cnt; i++) { lens[k][i] = results[b0 + i].length; skips[k][i] = results[b0 + i].skipFlags; stats[k][i] = results[b0 + i].status; }
      jobs.emplace_back(new HostInv());
      HostInv* H = jobs.back().get();
      H->device = ctx->device; H->types = types; H->hp = hp; H->blockSize = blockSize; H->cap = dataCap;
      H->dbuf = out + (int64_t)b0 * outStride; H->dstride = outStride; H->slotCap = outStride; H->hostMem = memKind == KZ_MEM_HOST;
      H->len = lens[k].data(); H->skip = skips[k].data(); H->status = stats[k].data();
      if (staged) {                                                   // ring of two pinned slot sets: chunk k - 2 must be done with its set
        if (k >= 2 && finishers[k - 2].joinable()) finishers[k - 2].join();
        H->pin = ctx->hsIn[k & 1].p; H->pinSlot = pslot;
      }
      kz_ctx* cx = ctx;
      finishers.emplace_back([H, cnt, k, trace, &ms, staged, cx]() {
        const double t0 = ms();
        if (staged) host_inverse_chunk_staged(cx, H, cnt, k & 1);
        else kz_parallel_for(cnt, KZ_HOST_STAGE_THREADS, host_inverse_block, H);
        if (trace) fprintf(stderr, "[pipe] decode chunk %d: host inverse stages %.0f .. %.0f ms\n", k, t0, ms());
      });
    }
    for (auto& t : finishers) if (t.joinable()) t.join();
    if (trace) fprintf(stderr, "[pipe] decode done at %.0f ms\n", ms());
    if (rc) { decode_error_sync(ctx); return rc; }
    for (int k = 0; k < nch; k++) {
      if (jobs[k]->fail) { snprintf(ctx->err, sizeof(ctx->err), "host stage: copy from / to the device failed"); return -KZ_ERR_DEVICE; }
      const int b0 = k * CH;
      for (size_t i = 0; i < lens[k].size(); i++) {
        kz_block_result& r = results[b0 + (int)i];
        if (!stats[k][i] && lens[k][i] > blockSize) stats[k][i] = -KZ_ERR_PROCESS_BLOCK;
        r.status = stats[k][i];
        r.length = stats[k][i] ? 0 : lens[k][i];
      }
    }
    return 0;
  }
  const int32_t rc = decode_blocks_impl(ctx, transformType, entropyType, blockSize, in, inStride, bitLengths, nBlocks, out, outStride, results, memKind);
  if (rc) decode_error_sync(ctx);
  return rc;
}

// =================================================================================================
// asynchronous batches: kz_submit_encode_blocks / kz_submit_decode_blocks queue the same call on the context's worker thread
// and return at once; kz_wait collects the call's return code.  A context runs its jobs one at a time in submission order
// (it owns one stream and one arena); overlap comes from a second context: one host thread can keep the H2D copies and the
// encode of batch k+1 on context A running under the decode of batch k on context B ("two streams"), which is how
// the reference's task pool overlaps blocks (K/io/CompressedOutputStream.java:541-566).
static void ctx_worker(kz_ctx* ctx) {
  for (;;) {
    std::pair<int64_t, std::function<int32_t()>> job;
    {
      std::unique_lock<std::mutex> lk(ctx->qmu);
      ctx->qcv.wait(lk, [&] { return ctx->stopWorker || !ctx->queue.empty(); });
      if (ctx->queue.empty()) return;                               // stop requested and nothing left
      job = std::move(ctx->queue.front());
      ctx->queue.pop_front();
    }
    const int32_t rc = job.second();
    { std::lock_guard<std::mutex> g(ctx->qmu); ctx->finished[job.first] = rc; }
    ctx->qcv.notify_all();
  }
}
static int64_t ctx_submit(kz_ctx* ctx, std::function<int32_t()> fn) {
  std::lock_guard<std::mutex> g(ctx->qmu);
  if (!ctx->worker.joinable()) ctx->worker = std::thread(ctx_worker, ctx);
  const int64_t id = ctx->nextJob++;
  ctx->queue.emplace_back(id, std::move(fn));
  ctx->qcv.notify_all();
  return id;
}
extern "C" int64_t kz_submit_encode_blocks(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType,
                                           const uint8_t* in, int64_t inStride, const int32_t* lengths, int32_t nBlocks,
                                           uint8_t* out, int64_t outStride, kz_block_result* results, int32_t memKind) {
  const int32_t bsz = encode_block_size(ctx, transformType);      // the context's "blockSize" entry NOW, not when the job runs
  if (bsz < 0) return bsz;
  return ctx_submit(ctx, [=]() { return encode_blocks_bs(ctx, transformType, entropyType, bsz, in, inStride, lengths, nBlocks, out, outStride, results, memKind); });
}
extern "C" int64_t kz_submit_decode_blocks(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType, int32_t blockSize,
                                           const uint8_t* in, int64_t inStride, const int64_t* bitLengths, int32_t nBlocks,
                                           uint8_t* out, int64_t outStride, kz_block_result* results, int32_t memKind) {
  if (!ctx) return -KZ_ERR_INVALID_PARAM;
  return ctx_submit(ctx, [=]() { return kz_decode_blocks(ctx, transformType, entropyType, blockSize, in, inStride, bitLengths, nBlocks, out, outStride, results, memKind); });
}
extern "C" int32_t kz_wait(kz_ctx* ctx, int64_t job) {
  if (!ctx || job <= 0) return -KZ_ERR_INVALID_PARAM;
  std::unique_lock<std::mutex> lk(ctx->qmu);
  if (job >= ctx->nextJob) return -KZ_ERR_INVALID_PARAM;
  if (job < ctx->collectedBelow || ctx->collected.count(job)) return -KZ_ERR_INVALID_PARAM;   // a job's result is handed out once
  ctx->qcv.wait(lk, [&] { return ctx->finished.count(job) != 0; });
  const int32_t rc = ctx->finished[job];
  ctx->finished.erase(job);
  ctx->collected.insert(job);
  while (ctx->collected.count(ctx->collectedBelow)) { ctx->collected.erase(ctx->collectedBelow); ctx->collectedBelow++; }
  return rc;
}
extern "C" int32_t kz_poll(kz_ctx* ctx, int64_t job) {            // 1 = finished (kz_wait will not block), 0 = queued or running
  if (!ctx || job <= 0) return -KZ_ERR_INVALID_PARAM;
  std::lock_guard<std::mutex> g(ctx->qmu);
  if (job >= ctx->nextJob) return -KZ_ERR_INVALID_PARAM;
  if (job < ctx->collectedBelow || ctx->collected.count(job)) return -KZ_ERR_INVALID_PARAM;
  return ctx->finished.count(job) ? 1 : 0;
}

// =================================================================================================
// single-block mirrors of ByteTransform / EntropyEncoder / EntropyDecoder (host buffers)
static int32_t transform_one(kz_ctx* ctx, uint32_t type, bool forward, const uint8_t* src, int32_t n, uint8_t* dst, int32_t dstCap, int32_t* produced) {
  if (!ctx || !src || !dst || !produced || n < 0) return -KZ_ERR_INVALID_PARAM;
  *produced = 0;
  if (n == 0) return 1;                                             // every reference codec: length 0 -> true
  if (!transform_supported((int)type) || type == KZ_T_NONE) { snprintf(ctx->err, sizeof(ctx->err), "transform %u has no HIP stage", type); return -KZ_ERR_INVALID_CODEC; }
  if (forward && dstCap < kz_transform_max_encoded_len(type, n)) return 0;   // e.g. ZRLT.java:68, BWTBlockCodec.java:84-86
  if (kz_is_host_transform((int)type)) {                                     // TEXT, UTF: host stages, no GPU involved
    int dt = ctx->dataType;
    const int r = forward ? kz_host_transform_forward((int)type, ctx->entropy, ctx->blockSize, &dt, src, n, dst, dstCap, produced)
                          : kz_host_transform_inverse((int)type, ctx->blockSize, src, n, dst, dstCap, produced);
    if (forward) ctx->dataType = dt;
    if (!r) *produced = 0;
    return r ? 1 : 0;
  }
  KZ_HIP(hipSetDevice(ctx->device));
  Pipe P;
  const int maxLen = std::max(kz_transform_max_encoded_len(type, n), std::max(dstCap, n));
  ChainSpec CS; CS.nb = 1; CS.types[0] = (int)type; CS.entropy = KZ_E_NONE;
  int rc = pipe_setup(ctx, P, 1, maxLen, 0, !forward, CS);
  if (rc) return rc;
  kz_batch& bt = P.bt;
  hipStream_t st = ctx->stream;
  KZ_HIP(hipMemcpyAsync(bt.buf[0], src, (size_t)n, hipMemcpyHostToDevice, st));
  bt.h_len[0] = n;
  KZ_HIP(hipMemcpyAsync(bt.d_len, &n, 4, hipMemcpyHostToDevice, st));
  // the instance's context entry "dataType" (kz_ctx_set_data_type): read by MM and LZ/LZX forward, rewritten by MM
  rc = kz_block_data_types(ctx, bt, ctx->dataType, false);
  if (rc) return rc;
  std::vector<int32_t> mask(1, 1), applied;
  rc = run_stage(ctx, P, mask, applied, [&](kz_batch& x) { return run_transform_stage(ctx, x, (int)type, forward, dstCap); });
  if (rc) return rc;
  if (forward) {
    int32_t dt = 0;
    KZ_HIP(hipMemcpyAsync(&dt, bt.d_dtype, 4, hipMemcpyDeviceToHost, st));
    KZ_HIP(kz_stream_sync(ctx, st));
    ctx->dataType = dt;
  }
  if (applied[0] < 0) { snprintf(ctx->err, sizeof(ctx->err), "transform %u: the reference codec throws on this block (LZ token buffer)", type); return -KZ_ERR_PROCESS_BLOCK; }
  if (!applied[0]) return 0;
  if (bt.h_len[0] > dstCap) return 0;
  *produced = bt.h_len[0];
  if (bt.h_len[0] > 0) KZ_HIP(hipMemcpyAsync(dst, bt.buf[bt.cur], (size_t)bt.h_len[0], hipMemcpyDeviceToHost, st));
  KZ_HIP(kz_stream_sync(ctx, st));
  return 1;
}
extern "C" int32_t kz_transform_forward(kz_ctx* ctx, uint32_t type, const uint8_t* src, int32_t n, uint8_t* dst, int32_t dstCap, int32_t* produced) {
  return transform_one(ctx, type, true, src, n, dst, dstCap, produced);
}
extern "C" int32_t kz_transform_inverse(kz_ctx* ctx, uint32_t type, const uint8_t* src, int32_t n, uint8_t* dst, int32_t dstCap, int32_t* produced) {
  return transform_one(ctx, type, false, src, n, dst, dstCap, produced);
}

extern "C" int64_t kz_entropy_encode(kz_ctx* ctx, uint32_t type, const uint8_t* src, int32_t n, uint8_t* out, int64_t outCapBytes) {
  if (!ctx || !src || !out || n < 0) return -KZ_ERR_INVALID_PARAM;
  if (!entropy_supported((int)type)) return -KZ_ERR_INVALID_CODEC;
  if (n == 0) {
    // encode() of nothing writes nothing; FPAQ's dispose() still flushes its 56-bit low register, all zero but the
    // 24 padding ones (FPAQEncoder.java:232-238): the call stands for encode + dispose
    if (type != KZ_E_FPAQ) return 0;
    if (outCapBytes < 7) return -KZ_ERR_INVALID_PARAM;
    const uint8_t flush[7] = {0, 0, 0, 0, 0xFF, 0xFF, 0xFF};
    memcpy(out, flush, 7);
    return 56;
  }
  KZ_HIP(hipSetDevice(ctx->device));
  Pipe P;
  const int64_t oS = kz_max_block_stream_bytes(n);
  ChainSpec CS; CS.nb = 0; CS.entropy = (int)type;
  int rc = pipe_setup(ctx, P, 1, n, oS + 256, false, CS);
  if (rc) return rc;
  kz_batch& bt = P.bt;
  hipStream_t st = ctx->stream;
  u8* d_out = (u8*)kz_arena_alloc(ctx, (size_t)oS);
  int32_t* d_hdr = (int32_t*)kz_arena_alloc(ctx, 64);
  int64_t* d_bits = (int64_t*)kz_arena_alloc(ctx, 64);
  if (!d_bits) return -KZ_ERR_DEVICE;
  KZ_HIP(hipMemcpyAsync(bt.buf[0], src, (size_t)n, hipMemcpyHostToDevice, st));
  bt.h_len[0] = n;
  KZ_HIP(hipMemcpyAsync(bt.d_len, &n, 4, hipMemcpyHostToDevice, st));
  KZ_HIP(hipMemsetAsync(d_out, 0, (size_t)oS, st));
  KZ_HIP(hipMemsetAsync(d_hdr, 0, 64, st));
  int64_t bits = 0;
  if (type == KZ_E_ANS0 || type == KZ_E_HUFFMAN || type == KZ_E_FPAQ) {
    rc = (type == KZ_E_ANS0) ? kz_stage_ans0_encode(ctx, bt, d_out, oS, d_hdr, d_bits) : (type == KZ_E_HUFFMAN) ? kz_stage_huffman_encode(ctx, bt, d_out, oS, d_hdr, d_bits) : kz_stage_fpaq_encode(ctx, bt, d_out, oS, d_hdr, d_bits);
    if (rc) return rc;
    KZ_HIP(hipMemcpyAsync(&bits, d_bits, 8, hipMemcpyDeviceToHost, st));
    KZ_HIP(kz_stream_sync(ctx, st));
  } else {
    KZ_HIP(hipMemcpyAsync(d_out, bt.buf[0], (size_t)n, hipMemcpyDeviceToDevice, st));
    bits = 8LL * n;
  }
  const int64_t nbytes = (bits + 7) >> 3;
  if (nbytes > outCapBytes) return -KZ_ERR_WRITE_FILE;
  KZ_HIP(hipMemcpyAsync(out, d_out, (size_t)nbytes, hipMemcpyDeviceToHost, st));
  KZ_HIP(kz_stream_sync(ctx, st));
  return bits;
}

extern "C" int32_t kz_entropy_decode(kz_ctx* ctx, uint32_t type, const uint8_t* in, int64_t inBits, uint8_t* dst, int32_t count, int64_t* bitsConsumed) {
  if (!ctx || !in || !dst || count < 0) return -KZ_ERR_INVALID_PARAM;
  if (!entropy_supported((int)type)) return -KZ_ERR_INVALID_CODEC;
  if (bitsConsumed) *bitsConsumed = 0;
  if (count == 0) return 0;
  KZ_HIP(hipSetDevice(ctx->device));
  Pipe P;
  const int64_t inBytes = (inBits + 7) >> 3;
  const int64_t inS = (int64_t)kz_align((size_t)inBytes + 64, 256);
  ChainSpec CS; CS.nb = 0; CS.entropy = (int)type;
  int rc = pipe_setup(ctx, P, 1, count, inS + 256, true, CS);
  if (rc) return rc;
  kz_batch& bt = P.bt;
  hipStream_t st = ctx->stream;
  u8* d_in = (u8*)kz_arena_alloc(ctx, (size_t)inS);
  int64_t* d_off = (int64_t*)kz_arena_alloc(ctx, 64);
  if (!d_off) return -KZ_ERR_DEVICE;
  KZ_HIP(hipMemsetAsync(d_in, 0, (size_t)inS, st));
  KZ_HIP(hipMemcpyAsync(d_in, in, (size_t)inBytes, hipMemcpyHostToDevice, st));
  int64_t h[2] = {0, inBits};
  KZ_HIP(hipMemcpyAsync(d_off, h, 16, hipMemcpyHostToDevice, st));
  bt.h_len[0] = count;
  KZ_HIP(hipMemcpyAsync(bt.d_len, &count, 4, hipMemcpyHostToDevice, st));
  if (type == KZ_E_ANS0 || type == KZ_E_HUFFMAN || type == KZ_E_FPAQ) {
    ctx->d_endBits = (long long*)(d_off + 2);
    rc = (type == KZ_E_ANS0) ? kz_stage_ans0_decode(ctx, bt, d_in, inS, d_off, d_off + 1) : (type == KZ_E_HUFFMAN) ? kz_stage_huffman_decode(ctx, bt, d_in, inS, d_off, d_off + 1) : kz_stage_fpaq_decode(ctx, bt, d_in, inS, d_off, d_off + 1);
    ctx->d_endBits = nullptr;
    if (rc) return rc;
    int32_t flag = 0;
    long long used = 0;
    KZ_HIP(hipMemcpyAsync(&flag, bt.d_flag, 4, hipMemcpyDeviceToHost, st));
    KZ_HIP(hipMemcpyAsync(&used, d_off + 2, 8, hipMemcpyDeviceToHost, st));
    KZ_HIP(kz_stream_sync(ctx, st));
    if (!flag) return -KZ_ERR_PROCESS_BLOCK;
    if (bitsConsumed) *bitsConsumed = (int64_t)used;
    KZ_HIP(hipMemcpyAsync(dst, bt.buf[bt.cur], (size_t)count, hipMemcpyDeviceToHost, st));
  } else {
    if (inBits < 8LL * count) return -KZ_ERR_PROCESS_BLOCK;
    KZ_HIP(hipMemcpyAsync(dst, d_in, (size_t)count, hipMemcpyDeviceToHost, st));
    if (bitsConsumed) *bitsConsumed = 8LL * count;
  }
  KZ_HIP(kz_stream_sync(ctx, st));
  return count;
}
