// kz_stream.hip -- host side of the .knz container around the HIP block pipeline: the part of
// K/io/CompressedOutputStream.java (writeHeader :236-313, ordered block emission :1024-1035, end
// marker :483-493) and K/io/CompressedInputStream.java (readHeader :359-515, block length prefix
// :1127-1129) that stays on the CPU.  Blocks go to the GPU in batches (kz_encode_blocks /
// kz_decode_blocks); only the bit-granular concatenation happens here.
#include "kz_internal.h"
#include <stdlib.h>
#include <algorithm>
#include <memory>
#include <thread>
#include <atomic>
#include <mutex>
#include <condition_variable>

namespace {
struct HostBits {            // MSB-first writer (DefaultOutputBitStream.java:103-205)
  uint8_t* p; int64_t cap; uint64_t pos; bool overflow;
  void put(uint64_t v, int count) {
    if (count <= 0) return;
    if ((int64_t)((pos + count + 7) >> 3) > cap) { overflow = true; return; }
    if (count < 64) v &= ((1ULL << count) - 1);
    while (count > 0) {
      const int bitoff = (int)(pos & 7), room = 8 - bitoff;
      const int take = count < room ? count : room;
      const uint8_t bits = (uint8_t)((v >> (count - take)) & ((1u << take) - 1));
      if (bitoff == 0) p[pos >> 3] = 0;
      p[pos >> 3] |= (uint8_t)(bits << (room - take));
      pos += take; count -= take;
    }
  }
  void putBytes(const uint8_t* s, uint64_t nbits) {
    if (!nbits) return;
    if ((int64_t)((pos + nbits + 7) >> 3) > cap) { overflow = true; return; }
    const int sh = (int)(pos & 7);
    const uint64_t full = nbits >> 3;
    if (sh == 0) { memcpy(p + (pos >> 3), s, (size_t)full); pos += full << 3; }
    else {
      uint8_t* d = p + (pos >> 3);
      const int up = 8 - sh;
      if (full) {
        d[0] = (uint8_t)((d[0] & (0xFF << up)) | (s[0] >> sh));         // keep the bits already in the partial byte
        for (uint64_t i = 1; i < full; i++) d[i] = (uint8_t)((s[i - 1] << up) | (s[i] >> sh));   // no carried dependency: vectorises
        d[full] = (uint8_t)(s[full - 1] << up);
      }
      pos += full << 3;
    }
    const int r = (int)(nbits & 7);
    if (r) put((uint64_t)(s[full] >> (8 - r)), r);
  }
};
struct HostBitsIn {
  const uint8_t* p; uint64_t nbits; uint64_t pos; bool error;
  uint64_t get(int count) {
    if (count <= 0) return 0;
    if (pos + (uint64_t)count > nbits) { error = true; pos = nbits; return 0; }
    uint64_t v = 0;
    while (count > 0) {
      const int bitoff = (int)(pos & 7), room = 8 - bitoff;
      const int take = count < room ? count : room;
      v = (v << take) | (uint64_t)((p[pos >> 3] >> (room - take)) & ((1u << take) - 1));
      pos += take; count -= take;
    }
    return v;
  }
  void getBytes(uint8_t* d, uint64_t nb) {
    if (pos + nb > nbits) { error = true; pos = nbits; return; }
    const int sh = (int)(pos & 7);
    const uint64_t full = nb >> 3;
    const uint8_t* s = p + (pos >> 3);
    if (sh == 0) memcpy(d, s, (size_t)full);
    else for (uint64_t i = 0; i < full; i++) d[i] = (uint8_t)((s[i] << sh) | (s[i + 1] >> (8 - sh)));
    pos += full << 3;
    const int r = (int)(nb & 7);
    if (r) d[full] = (uint8_t)(get(r) << (8 - r));
  }
};
inline uint32_t mix32(uint32_t c, uint32_t h, uint32_t v) { c ^= h * ~v; c = (c << 13) | (c >> 19); return c * 5u + 0x52DCE729u; }
inline int ilog2(uint32_t x) { return 31 - __builtin_clz(x); }

uint32_t header_cksum(int chkKind, int entropyType, uint64_t tt, int blockSize, int szMask, int64_t inputSize) {
  const uint32_t HASH = 0x1E35A7BDu;                          // CompressedOutputStream.java:293-309
  uint32_t c = HASH * (0x01030507u * 7u);
  c = mix32(c, HASH, (uint32_t)chkKind);
  c = mix32(c, HASH, (uint32_t)entropyType);
  c = mix32(c, HASH, (uint32_t)(tt >> 32));
  c = mix32(c, HASH, (uint32_t)tt);
  c = mix32(c, HASH, (uint32_t)blockSize);
  if (szMask > 0) { c = mix32(c, HASH, (uint32_t)((uint64_t)inputSize >> 32)); c = mix32(c, HASH, (uint32_t)inputSize); }
  return ((c >> 23) ^ (c >> 3)) & 0xFFFFFF;
}
int batch_blocks(int blockSize) {
  // blocks per kz_encode_blocks / kz_decode_blocks call of the host-buffer entry points: 8 GiB of input per call
  // (device scratch is bounded separately by the arena budget, which splits a call into sub-batches); the
  // serial kernels want a thousand blocks or more in flight.  The strided host staging buffers are allocated
  // uninitialised, so only the bytes actually produced/consumed are ever touched.
  return (int)std::max<int64_t>(1, std::min<int64_t>(2048, (8LL << 30) / blockSize));
}
}  // namespace

// ---- host-only container helpers (no GPU needed): used by kz_compress and by multi-GPU gathers ----
static int write_stream_header(HostBits& bs, uint64_t transformType, uint32_t entropyType, int32_t blockSize, int64_t n, int chkKind = 0) {
  bs.put(0x4B414E5A, 32); bs.put(7, 4); bs.put((uint64_t)chkKind, 2);
  bs.put(entropyType, 5); bs.put(transformType, 48); bs.put((uint32_t)blockSize >> 4, 28);
  int szMask = 0;
  if (n != 0 && n < (1LL << 48)) {
    if (n >= (1LL << 32)) szMask = 3;
    else { int64_t isz = n; if (isz > (1LL << 30)) { isz >>= 4; szMask++; } szMask += (ilog2((uint32_t)isz) >> 4) + 1; }
  }
  bs.put((uint64_t)szMask, 2);
  if (szMask > 0) bs.put((uint64_t)n, 16 * szMask);
  bs.put(0, 15);
  bs.put(header_cksum(chkKind, (int)entropyType, transformType, blockSize, szMask, n), 24);
  return szMask;
}
static void write_block(HostBits& bs, const uint8_t* stream, uint64_t written) {   // :1024-1035
  const int lw = (written < 8) ? 3 : ilog2((uint32_t)(written >> 3)) + 4;
  bs.put((uint64_t)(lw - 3), 5);
  bs.put(written, lw);
  bs.putBytes(stream, written);
}

// Assemble a complete .knz stream from per-block private streams (e.g. gathered from several GPUs
// in block-id order).  Pure host code.
extern "C" int64_t kz_knz_assemble(uint64_t transformType, uint32_t entropyType, int32_t blockSize, int64_t inputSize,
                                   int32_t checksumBits, const uint8_t* streams, int64_t stride, const int64_t* bits, int32_t nBlocks,
                                   uint8_t* dst, int64_t dstCap) {
  if (!dst || nBlocks < 0 || (nBlocks > 0 && (!streams || !bits))) return -KZ_ERR_INVALID_PARAM;
  if (checksumBits != 0 && checksumBits != 32 && checksumBits != 64) return -KZ_ERR_INVALID_PARAM;
  HostBits bs{dst, dstCap, 0, false};
  // the block streams carry 4 / 8 hash bytes in their headers when they were coded with kz_ctx_set_checksum(32 / 64): the
  // stream header must say so (CompressedOutputStream.java:244-250) or no reader can parse them
  write_stream_header(bs, transformType, entropyType, blockSize, inputSize, checksumBits == 32 ? 1 : (checksumBits == 64 ? 2 : 0));
  for (int i = 0; i < nBlocks; i++) if (bits[i] > 0) write_block(bs, streams + (int64_t)i * stride, (uint64_t)bits[i]);
  bs.put(0, 5); bs.put(0, 3);
  if (bs.overflow) return -KZ_ERR_WRITE_FILE;
  return (int64_t)((bs.pos + 7) >> 3);
}

// The same, block by block: a gatherer (rank 0 of a multi-GPU job) appends block streams in block-id order as they arrive instead
// of holding all of them first; memory is one block plus the destination.
struct kz_knz_writer { HostBits bs; };
extern "C" kz_knz_writer* kz_knz_writer_open(uint64_t transformType, uint32_t entropyType, int32_t blockSize, int64_t inputSize,
                                             int32_t checksumBits, uint8_t* dst, int64_t dstCap) {
  if (!dst || dstCap < 32 || (checksumBits != 0 && checksumBits != 32 && checksumBits != 64)) return nullptr;
  kz_knz_writer* w = new kz_knz_writer{HostBits{dst, dstCap, 0, false}};
  write_stream_header(w->bs, transformType, entropyType, blockSize, inputSize, checksumBits == 32 ? 1 : (checksumBits == 64 ? 2 : 0));
  return w;
}
extern "C" int32_t kz_knz_writer_add(kz_knz_writer* w, const uint8_t* stream, int64_t bits) {
  if (!w || bits < 0 || (bits > 0 && !stream)) return -KZ_ERR_INVALID_PARAM;
  if (bits > 0) write_block(w->bs, stream, (uint64_t)bits);
  return w->bs.overflow ? -KZ_ERR_WRITE_FILE : 0;
}
extern "C" int64_t kz_knz_writer_close(kz_knz_writer* w) {          // end marker (:491-492); returns the stream's size in bytes
  if (!w) return -KZ_ERR_INVALID_PARAM;
  w->bs.put(0, 5); w->bs.put(0, 3);
  const int64_t r = w->bs.overflow ? -KZ_ERR_WRITE_FILE : (int64_t)((w->bs.pos + 7) >> 3);
  delete w;
  return r;
}

// Parse the stream header and walk the block length prefixes: blockBitOff[i] = bit offset of block
// i's private stream inside src, blockBits[i] = its length W.  Returns the number of blocks or <0.
// Stream header, checked in the reference's order and with its codes (CompressedInputStream.java:359-478).
struct KnzHeader { int chkKind, entropyType, blockSize, szMask; uint64_t transformType; int64_t inputSize; };
static int read_stream_header(HostBitsIn& bs, KnzHeader& h, char* err, size_t errCap) {
  if (bs.get(32) != 0x4B414E5A) return -KZ_ERR_INVALID_FILE;                                    // :367-368
  const int bsVersion = (int)bs.get(4);
  if (bsVersion != 7) {                                                                            // :374-377; older layouts are not built here
    if (err) snprintf(err, errCap, "cannot read this version of the stream: %d (only 7 is built)", bsVersion);
    return -KZ_ERR_STREAM_VERSION;
  }
  h.chkKind = (int)bs.get(2);
  if (h.chkKind > 2) return -KZ_ERR_INVALID_FILE;                                               // :390-392
  h.entropyType = (int)bs.get(5);
  if (h.entropyType == 3 || h.entropyType > 9) return -KZ_ERR_INVALID_CODEC;                     // EntropyCodecFactory.getName :212-243
  h.transformType = bs.get(48);
  for (int i = 0; i < 8; i++) {                                                                  // TransformFactory.getName :368-449
    const int t = (int)((h.transformType >> (42 - 6 * i)) & 0x3F);
    if (t == 4 || t > 19) return -KZ_ERR_INVALID_CODEC;
  }
  h.blockSize = (int)(bs.get(28) << 4);
  if (h.blockSize < 1024 || h.blockSize > (1 << 30)) return -KZ_ERR_BLOCK_SIZE;                  // :419-422
  h.szMask = (int)bs.get(2);
  h.inputSize = h.szMask ? (int64_t)bs.get(16 * h.szMask) : 0;
  bs.get(15);
  const uint32_t ck = (uint32_t)bs.get(24);
  if (bs.error || ck != header_cksum(h.chkKind, h.entropyType, h.transformType, h.blockSize, h.szMask, h.inputSize))
    return -KZ_ERR_CRC_CHECK;                                                                    // :477-478
  return 0;
}

extern "C" int32_t kz_knz_index(const uint8_t* src, int64_t n, uint64_t* transformType, uint32_t* entropyType,
                                int32_t* blockSize, int64_t* inputSize, int32_t* checksumBits, int64_t* blockBitOff, int64_t* blockBits, int32_t cap) {
  if (!src || n < 20) return -KZ_ERR_INVALID_FILE;
  HostBitsIn bs{src, (uint64_t)n * 8, 0, false};
  KnzHeader h;
  { const int hrc = read_stream_header(bs, h, nullptr, 0); if (hrc) return hrc; }
  const uint64_t tt = h.transformType; const int et = h.entropyType, bsz = h.blockSize; const int64_t isz = h.inputSize;
  if (transformType) *transformType = tt;
  if (entropyType) *entropyType = (uint32_t)et;
  if (blockSize) *blockSize = bsz;
  if (inputSize) *inputSize = isz;
  if (checksumBits) *checksumBits = h.chkKind == 1 ? 32 : (h.chkKind == 2 ? 64 : 0);
  int nb = 0;
  for (;;) {
    const int lr = (int)bs.get(5) + 3;
    const uint64_t rd = bs.get(lr);
    if (bs.error) return -KZ_ERR_READ_FILE;
    if (rd == 0) break;
    if (nb < cap) { if (blockBitOff) blockBitOff[nb] = (int64_t)bs.pos; if (blockBits) blockBits[nb] = (int64_t)rd; }
    nb++;
    if (bs.pos + rd > bs.nbits) return -KZ_ERR_READ_FILE;
    bs.pos += rd;
  }
  return nb;
}


// run fn(i) for i in [0, n) on the host pool (blocks are independent)
template <typename F>
static void parallel_blocks(int n, F fn, int maxThreads = 32) {
  struct Thunk { static void run(int i, void* a) { (*(F*)a)(i); } };
  kz_parallel_for(n, maxThreads, &Thunk::run, (void*)&fn);
}

// Ordered emission of a batch (CompressedOutputStream.java:1024-1035) at bit granularity.  The serial pass writes,
// per block, the 5-bit / lw-bit length prefix and the (at most 7 + 7) payload bits that share a byte with a
// neighbour, skipping the bytes that belong to one payload only; those interiors are then copied (shifted) by a
// few threads.
static void emit_blocks(HostBits& bs, const uint8_t* streams, int64_t stride, const kz_block_result* res, int cnt) {
  struct Job { const uint8_t* s; uint8_t* d; uint64_t nbytes; int e; };
  std::vector<Job> jobs;
  jobs.reserve(cnt);
  for (int i = 0; i < cnt; i++) {
    const uint64_t W = (uint64_t)res[i].bits;
    if (W == 0) continue;
    const uint8_t* s = streams + (size_t)i * stride;
    const int lw = (W < 8) ? 3 : ilog2((uint32_t)(W >> 3)) + 4;
    bs.put((uint64_t)(lw - 3), 5);
    bs.put(W, lw);
    if (bs.overflow || (int64_t)((bs.pos + W + 7) >> 3) > bs.cap) { bs.overflow = true; return; }
    const int e = (int)((8 - (bs.pos & 7)) & 7);                  // payload bits that complete the current byte
    if (W < (uint64_t)e + 16) { bs.putBytes(s, W); continue; }    // tiny block: all serial
    if (e) bs.put((uint64_t)(s[0] >> (8 - e)), e);
    const uint64_t nbytes = (W - (uint64_t)e) >> 3;               // whole bytes owned by this payload alone
    jobs.push_back({s, bs.p + (bs.pos >> 3), nbytes, e});
    bs.pos += nbytes << 3;
    const int r = (int)((W - (uint64_t)e) & 7);                   // trailing bits
    if (r) {
      const uint64_t bitoff = (uint64_t)e + (nbytes << 3);
      const uint32_t two = ((uint32_t)s[bitoff >> 3] << 8) | (uint32_t)s[(bitoff >> 3) + 1];
      bs.put((uint64_t)((two >> (16 - (int)(bitoff & 7) - r)) & ((1u << r) - 1)), r);
    }
  }
  parallel_blocks((int)jobs.size(), [&](int k) {
    const Job& j = jobs[k];
    if (j.e == 0) { memcpy(j.d, j.s, (size_t)j.nbytes); return; }
    const int e = j.e, up = 8 - e;
    for (uint64_t i = 0; i < j.nbytes; i++) j.d[i] = (uint8_t)((j.s[i] << e) | (j.s[i + 1] >> up));
  });
}

extern "C" int64_t kz_compress_bound(int64_t n, int32_t blockSize) {
  if (n < 0 || blockSize <= 0) return -KZ_ERR_INVALID_PARAM;
  const int64_t nBlocks = (n + blockSize - 1) / blockSize;
  // per block: its stream (kz_max_block_stream_bytes) + length prefix and checksum; stream header, end marker, padding
  return 64 + nBlocks * (kz_max_block_stream_bytes((int32_t)std::min<int64_t>(n, blockSize)) + 24);
}

// ---- staging for the host-buffer stream calls: pinned host memory and HBM outside the arena, grow-only, kept by the context ----
static int stage_reserve(kz_ctx* ctx, kz_ctx::Stage& s, size_t need, bool pinned) { return kz_stage_reserve(ctx, s, need, pinned); }
static int stage_streams(kz_ctx* ctx) {
  if (!ctx->copyUp) KZ_HIP(hipStreamCreateWithFlags(&ctx->copyUp, hipStreamNonBlocking));
  if (!ctx->copyDown) KZ_HIP(hipStreamCreateWithFlags(&ctx->copyDown, hipStreamNonBlocking));
  return 0;
}
static int stream_chunk_blocks(const kz_ctx* ctx, int blockSize) {   // blocks per pipeline step of kz_compress: about 1 GiB of input
  if (ctx->sw.streamChunk > 0) return ctx->sw.streamChunk;
  return (int)std::max<int64_t>(8, std::min<int64_t>(2048, (1LL << 30) / blockSize));
}
struct BlockSizeScope {                                            // the context's "blockSize" entry = this stream's (TEXT reads it)
  kz_ctx* c; int saved; bool savedSet;
  BlockSizeScope(kz_ctx* c_, int v) : c(c_), saved(c_->blockSize), savedSet(c_->blockSizeSet) { c->blockSize = v; c->blockSizeSet = true; }
  ~BlockSizeScope() { c->blockSize = saved; c->blockSizeSet = savedSet; }
};

// Inputs of several chunks: a three-stage pipeline like the reference's task pool (K/io/CompressedOutputStream.java:541-566 fills
// `jobs` buffers, codes them side by side and emits in order).  Thread U stages chunk k+1 (host TEXT / UTF stages, copy into pinned
// memory, H2D on its own stream), the calling thread codes chunk k on the GPU (HBM in, HBM out), thread D brings chunk k-1 back
// (D2H of the bytes produced, ordered bit-granular emission into dst).  Blocks are independent: the stream is the same as the
// one-batch-at-a-time form's.
static int64_t compress_pipelined(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType, int32_t blockSize,
                                  const uint8_t* src, int64_t n, HostBits& bs, int64_t nblocks, int CH) {
  const int nch = (int)((nblocks + CH - 1) / CH);
  const int64_t oS = kz_max_block_stream_bytes(blockSize);
  { int rc = stage_streams(ctx); if (rc) return rc; }
  for (int i = 0; i < 2; i++) {
    int rc = stage_reserve(ctx, ctx->pinIn[i], (size_t)CH * blockSize, true);
    if (!rc) rc = stage_reserve(ctx, ctx->devIn[i], (size_t)CH * blockSize, false);
    if (!rc) rc = stage_reserve(ctx, ctx->devOut[i], (size_t)CH * oS, false);
    if (!rc) rc = stage_reserve(ctx, ctx->pinOut[i], (size_t)CH * oS, true);
    if (rc) return rc;
  }
  struct Chunk { std::vector<int32_t> lens; std::vector<kz_block_result> res; HostPre* pre = nullptr; };
  std::vector<Chunk> ck(nch);
  std::mutex mu;
  std::condition_variable cv;
  int uploaded = 0, encoded = 0, emitted = 0, failed = 0;          // chunks through each stage (under mu)
  auto fail = [&](int code) { std::lock_guard<std::mutex> g(mu); if (!failed) failed = code; cv.notify_all(); };
  const int dev = ctx->device;
  std::thread U([&]() {
    if (hipSetDevice(dev) != hipSuccess) { fail(-KZ_ERR_DEVICE); return; }
    for (int k = 0; k < nch; k++) {
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return failed || encoded >= k - 1; }); if (failed) return; }   // slot k & 1 is free
      const int64_t b0 = (int64_t)k * CH;
      const int cnt = (int)std::min<int64_t>(CH, nblocks - b0);
      ck[k].lens.resize(cnt);
      for (int i = 0; i < cnt; i++) ck[k].lens[i] = (int32_t)std::min<int64_t>(blockSize, n - (b0 + i) * blockSize);
      const uint8_t* cs = src + b0 * blockSize;
      // (a chunk whose TEXT stage runs on the device is not pre-staged: kz_encode_blocks_pre does it from the uploaded blocks)
      ck[k].pre = kz_text_fwd_gpu_applies(ctx, transformType, entropyType, cnt) ? nullptr
                : kz_host_prestage(ctx, transformType, entropyType, blockSize, cs, blockSize, ck[k].lens.data(), cnt, k & 1);
      uint8_t* pin = ctx->pinIn[k & 1].p;
      const int64_t bytes = std::min<int64_t>((int64_t)cnt * blockSize, n - b0 * blockSize);
      const int pieces = (int)((bytes + (8 << 20) - 1) / (8 << 20));
      parallel_blocks(pieces, [&](int q) { const int64_t o = (int64_t)q << 23; memcpy(pin + o, cs + o, (size_t)std::min<int64_t>(8 << 20, bytes - o)); });
      if (hipMemcpyAsync(ctx->devIn[k & 1].p, pin, (size_t)bytes, hipMemcpyHostToDevice, ctx->copyUp) != hipSuccess ||
          hipStreamSynchronize(ctx->copyUp) != hipSuccess) { fail(-KZ_ERR_DEVICE); return; }
      { std::lock_guard<std::mutex> g(mu); uploaded = k + 1; }
      cv.notify_all();
    }
  });
  std::thread D([&]() {
    if (hipSetDevice(dev) != hipSuccess) { fail(-KZ_ERR_DEVICE); return; }
    for (int k = 0; k < nch; k++) {
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return failed || encoded >= k + 1; }); if (failed) return; }
      const int cnt = (int)ck[k].lens.size();
      uint8_t* pin = ctx->pinOut[k & 1].p;
      const uint8_t* d = ctx->devOut[k & 1].p;
      bool ok = true;
      for (int i = 0; i < cnt && ok; i++) {
        if (ck[k].res[i].status) { fail(ck[k].res[i].status); return; }
        const size_t nby = (size_t)((ck[k].res[i].bits + 7) >> 3);
        if (nby) ok = hipMemcpyAsync(pin + (size_t)i * oS, d + (size_t)i * oS, nby + 1 <= (size_t)oS ? nby + 1 : nby, hipMemcpyDeviceToHost, ctx->copyDown) == hipSuccess;
      }
      if (!ok || hipStreamSynchronize(ctx->copyDown) != hipSuccess) { fail(-KZ_ERR_DEVICE); return; }
      emit_blocks(bs, pin, oS, ck[k].res.data(), cnt);              // ordered emission (:1024-1035)
      if (bs.overflow) { fail(-KZ_ERR_WRITE_FILE); return; }
      { std::lock_guard<std::mutex> g(mu); emitted = k + 1; }
      cv.notify_all();
    }
  });
  for (int k = 0; k < nch; k++) {
    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return failed || (uploaded >= k + 1 && emitted >= k - 1); }); if (failed) break; }
    const int cnt = (int)ck[k].lens.size();
    ck[k].res.resize(cnt);
    const int rc = kz_encode_blocks_pre(ctx, transformType, entropyType, blockSize, ctx->devIn[k & 1].p, blockSize, ck[k].lens.data(), cnt,
                                        ctx->devOut[k & 1].p, oS, ck[k].res.data(), KZ_MEM_DEVICE, ck[k].pre);
    kz_host_pre_free(ck[k].pre); ck[k].pre = nullptr;
    if (rc) { fail(rc); break; }
    { std::lock_guard<std::mutex> g(mu); encoded = k + 1; }
    cv.notify_all();
  }
  U.join(); D.join();
  for (auto& c : ck) kz_host_pre_free(c.pre);
  return failed;
}

extern "C" int64_t kz_compress(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType, int32_t blockSize,
                               const uint8_t* src, int64_t n, uint8_t* dst, int64_t dstCap) {
  if (!ctx || !src || !dst || n < 0) return -KZ_ERR_INVALID_PARAM;
  if (blockSize < 1024 || blockSize > (1 << 30) || (blockSize & 15)) return -KZ_ERR_BLOCK_SIZE;   // :165-174
  KZ_HIP(hipSetDevice(ctx->device));
  HostBits bs{dst, dstCap, 0, false};
  // ---- stream header (CompressedOutputStream.java:236-313) ----
  write_stream_header(bs, transformType, entropyType, blockSize, n, ctx->checksum);
  const int64_t nblocks = (n + blockSize - 1) / blockSize;
  BlockSizeScope scope(ctx, blockSize);
  const int CH = stream_chunk_blocks(ctx, blockSize);
  if (nblocks >= 2LL * CH && !ctx->sw.streamSerial) {
    const int64_t rc = compress_pipelined(ctx, transformType, entropyType, blockSize, src, n, bs, nblocks, CH);
    if (rc == -KZ_ERR_WRITE_FILE) snprintf(ctx->err, sizeof(ctx->err), "kz_compress: destination too small");
    if (rc) return rc;
  } else {
    // ---- blocks, one batch after the other ----
    const int NB = batch_blocks(blockSize);
    const int64_t oS = kz_max_block_stream_bytes(blockSize);
    std::unique_ptr<uint8_t[]> outbuf(new uint8_t[(size_t)oS * (size_t)std::min<int64_t>(NB, std::max<int64_t>(nblocks, 1))]);
    std::vector<int32_t> lens(NB);
    std::vector<kz_block_result> res(NB);
    for (int64_t b0 = 0; b0 < nblocks; b0 += NB) {
      const int cnt = (int)std::min<int64_t>(NB, nblocks - b0);
      for (int i = 0; i < cnt; i++) lens[i] = (int32_t)std::min<int64_t>(blockSize, n - (b0 + i) * blockSize);
      int rc = kz_encode_blocks(ctx, transformType, entropyType, src + b0 * blockSize, blockSize, lens.data(), cnt,
                                outbuf.get(), oS, res.data(), KZ_MEM_HOST);
      if (rc) return rc;
      for (int i = 0; i < cnt; i++) if (res[i].status) return res[i].status;
      emit_blocks(bs, outbuf.get(), oS, res.data(), cnt);           // ordered emission (:1024-1035)
      if (bs.overflow) break;
    }
  }
  bs.put(0, 5); bs.put(0, 3);                                   // end marker (:491-492)
  if (bs.overflow) { snprintf(ctx->err, sizeof(ctx->err), "kz_compress: destination too small"); return -KZ_ERR_WRITE_FILE; }
  return (int64_t)((bs.pos + 7) >> 3);
}


// Host restatement of DecodingTask.readBlockHeader + the two checks that follow it
// (CompressedInputStream.java:1025-1095, :1145-1164): the reference validates a block's header on the shared
// bit stream BEFORE it reads the payload, so a truncated or tampered stream is reported with the header's error
// (ERR_CRC_CHECK / ERR_BLOCK_SIZE / ERR_READ_FILE), not as a short read.  Returns 0 or -(error code).
static int precheck_block_header(HostBitsIn bs /* by value: peek */, uint64_t W, int nbFunctions, int blockSize, int chkKind) {
  if (W < 8) return -KZ_ERR_BLOCK_SIZE;
  const uint32_t mode = (uint32_t)bs.get(8);
  uint32_t skipFlags = 0;
  bool hasSkip = false;
  if (mode & 0x80) {
    if (mode & 0x10) { if (nbFunctions > 4) hasSkip = true; else skipFlags = ((mode << 4) | 0x0F) & 0xFF; }
  } else if (mode & 0x10) hasSkip = true;
  else skipFlags = ((mode << 4) | 0x0F) & 0xFF;
  const int dataSize = 1 + (int)((mode >> 5) & 3);
  const int headerSize = 1 + (hasSkip ? 1 : 0) + dataSize + 1;
  if (W < (uint64_t)headerSize * 8) return -KZ_ERR_BLOCK_SIZE;
  if (hasSkip) skipFlags = (uint32_t)bs.get(8);
  uint32_t preLen = 0;
  for (int i = 0; i < dataSize; i++) preLen = (preLen << 8) | (uint32_t)bs.get(8);
  const uint32_t ck = (uint32_t)bs.get(8);
  if (bs.error) return -KZ_ERR_READ_FILE;
  const uint32_t HASH = 0x1E35A7BDu;
  uint32_t c = HASH * 0x01030507u;
  c = mix32(c, HASH, mode & 0xFF);
  c = mix32(c, HASH, skipFlags & 0xFF);
  c = mix32(c, HASH, preLen);
  c = mix32(c, HASH, (uint32_t)(W >> 32));
  c = mix32(c, HASH, (uint32_t)W);
  c = (c >> 23) ^ (c >> 3);
  if (ck != (c & 0xFF)) return -KZ_ERR_CRC_CHECK;
  const int64_t maxTL = std::min<int64_t>(std::max<int64_t>((int64_t)blockSize + blockSize / 2, 2048), 1LL << 30);
  if ((int32_t)preLen < 0 || (int64_t)preLen > maxTL) return -KZ_ERR_READ_FILE;
  const int checksumSize = (chkKind == 2) ? 8 : (chkKind == 1 ? 4 : 0);
  if ((int64_t)((W + 7) >> 3) > (int64_t)preLen + headerSize + checksumSize) return -KZ_ERR_BLOCK_SIZE;
  return 0;
}

extern "C" int64_t kz_decompress(kz_ctx* ctx, const uint8_t* src, int64_t n, uint8_t* dst, int64_t dstCap) {
  if (!ctx || !src || !dst || n < 0) return -KZ_ERR_INVALID_PARAM;
  KZ_HIP(hipSetDevice(ctx->device));
  HostBitsIn bs{src, (uint64_t)n * 8, 0, false};
  KnzHeader h;
  { const int hrc = read_stream_header(bs, h, ctx->err, sizeof(ctx->err)); if (hrc) return hrc; }
  const int chkKind = h.chkKind, entropyType = h.entropyType, blockSize = h.blockSize, szMask = h.szMask;
  const uint64_t tt = h.transformType; const int64_t inputSize = h.inputSize;
  const int savedChk = ctx->checksum;
  ctx->checksum = chkKind;
  struct Restore { kz_ctx* c; int v; ~Restore() { c->checksum = v; } } restore_{ctx, savedChk};
  int nbFunctions = 0;
  for (int i = 0; i < 8; i++) if (((tt >> (42 - 6 * i)) & 0x3F) != 0) nbFunctions++;         // Sequence length (TransformFactory.java:240-266)
  if (nbFunctions == 0) nbFunctions = 1;
  int NB = batch_blocks(blockSize);
  // no more staging than the stream can need: declared size if present, else >= 8 bytes of stream per block
  if (szMask && inputSize > 0) NB = (int)std::min<int64_t>(NB, (inputSize + blockSize - 1) / blockSize);
  else NB = (int)std::min<int64_t>(NB, n / 8 + 1);
  const int64_t iS = (int64_t)kz_align((size_t)blockSize + (size_t)(blockSize >> 3) + 1024 + 64, 256);
  std::unique_ptr<uint8_t[]> inbuf;                                // (the unstaged path's strided copy of the payloads)
  std::vector<int64_t> bits(NB);
  std::vector<uint64_t> starts(NB);
  std::vector<kz_block_result> res(NB);
  int64_t produced = 0;
  bool done = false;
  int pending = 0;
  while (!done) {
    int cnt = 0;
    // A fault met while walking the length prefixes is held back until the blocks before it have been decoded:
    // the reference reports the FIRST failing block in stream order (processedBlockId ordering, :1127-1170).
    while (cnt < NB) {                                          // serial walk of block length prefixes (:1127-1129)
      const int lr = (int)bs.get(5) + 3;
      const uint64_t rd = bs.get(lr);
      if (bs.error) { pending = -KZ_ERR_READ_FILE; break; }
      if (rd == 0) { done = true; break; }
      pending = precheck_block_header(bs, rd, nbFunctions, blockSize, chkKind);
      if (!pending && (int64_t)((rd + 7) >> 3) > iS - 64) pending = -KZ_ERR_BLOCK_SIZE;
      if (!pending && bs.pos + rd > bs.nbits) pending = -KZ_ERR_READ_FILE;
      if (pending) break;
      starts[cnt] = bs.pos;                                     // the payload is extracted below, by several threads
      bs.pos += rd;
      bits[cnt++] = (int64_t)rd;
    }
    if (pending) done = true;
    if (cnt == 0) break;
    {
      // Blocks that start past the end of the destination cannot be stored: that is this interface's fault to report
      // (ERR_WRITE_FILE), but only if every block before them decodes -- the reader reports the first failing block.
      const int64_t room = dstCap - produced;
      const int64_t fit = (room <= 0) ? 0 : (room + blockSize - 1) / blockSize;
      if (fit < cnt) { cnt = (int)fit; pending = -KZ_ERR_WRITE_FILE; done = true; }
      if (cnt == 0) break;
    }
    if (cnt >= 16 && !ctx->sw.streamSerial) {
      // ---- staged batch: payloads are extracted (bit-shifted) straight into pinned memory, go to HBM with asynchronous copies,
      //      the batch is decoded HBM -> HBM, and the blocks come back through a pinned ring while host threads move the previous
      //      piece to its place in dst ----
      { int rc = stage_streams(ctx); if (rc) return rc; }
      std::vector<int64_t> off(cnt + 1, 0);
      for (int i = 0; i < cnt; i++) off[i + 1] = off[i] + (int64_t)kz_align((size_t)((bits[i] + 7) >> 3) + 64, 256);
      const int P = (int)std::max<int64_t>(1, std::min<int64_t>(cnt, (256LL << 20) / blockSize));   // blocks per piece of the way back
      int rc = stage_reserve(ctx, ctx->pinIn[0], (size_t)off[cnt], true);
      if (!rc) rc = stage_reserve(ctx, ctx->devIn[0], (size_t)cnt * iS, false);
      if (!rc) rc = stage_reserve(ctx, ctx->devOut[0], (size_t)cnt * blockSize, false);
      for (int q = 0; q < 2 && !rc; q++) rc = stage_reserve(ctx, ctx->pinOut[q], (size_t)P * blockSize, true);
      if (rc) return rc;
      uint8_t* pin = ctx->pinIn[0].p;
      parallel_blocks(cnt, [&](int i) { HostBitsIn t = bs; t.pos = starts[i]; t.error = false; t.getBytes(pin + off[i], (uint64_t)bits[i]); });
      for (int i = 0; i < cnt; i++)
        KZ_HIP(hipMemcpyAsync(ctx->devIn[0].p + (size_t)i * iS, pin + off[i], (size_t)((bits[i] + 7) >> 3), hipMemcpyHostToDevice, ctx->copyUp));
      KZ_HIP(hipStreamSynchronize(ctx->copyUp));
      rc = kz_decode_blocks(ctx, tt, (uint32_t)entropyType, blockSize, ctx->devIn[0].p, iS, bits.data(), cnt,
                            ctx->devOut[0].p, blockSize, res.data(), KZ_MEM_DEVICE);
      if (rc) return rc;
      // where every block goes: the reader appends whatever a block produced (:783-785); the first failing block ends the stream
      std::vector<int64_t> at(cnt + 1, produced);
      int good = 0, code = 0;
      for (; good < cnt; good++) {
        if (res[good].status) { code = res[good].status; break; }
        if (res[good].length > blockSize) { code = -KZ_ERR_PROCESS_BLOCK; break; }              // "incorrectly decompressed" (:756-759)
        if (at[good] + res[good].length > dstCap) { code = -KZ_ERR_WRITE_FILE; break; }
        at[good + 1] = at[good] + res[good].length;
      }
      // a failing block ends the stream, but the blocks in front of it are delivered first, as the unstaged path and the
      // reference's reader do: dst holds the same bytes on error whatever the batch size
      const int ncopy = code ? good : cnt;
      const int pieces = (ncopy + P - 1) / P;
      auto fetch = [&](int p) -> hipError_t {
        const int i0 = p * P, c = std::min(P, ncopy - i0);
        return hipMemcpyAsync(ctx->pinOut[p & 1].p, ctx->devOut[0].p + (size_t)i0 * blockSize, (size_t)c * blockSize, hipMemcpyDeviceToHost, ctx->copyDown);
      };
      if (pieces > 0) KZ_HIP(fetch(0));
      for (int p = 0; p < pieces; p++) {
        KZ_HIP(hipStreamSynchronize(ctx->copyDown));
        if (p + 1 < pieces) KZ_HIP(fetch(p + 1));                       // the next piece travels while this one is put in place
        const int i0 = p * P, c = std::min(P, ncopy - i0);
        const uint8_t* ring = ctx->pinOut[p & 1].p;
        parallel_blocks(c * 4, [&](int q) {                             // four copies per block
          const int i = i0 + (q >> 2);
          const int64_t len = res[i].length, a = (len * (q & 3)) >> 2, b = (len * ((q & 3) + 1)) >> 2;
          if (b > a) memcpy(dst + at[i] + a, ring + (size_t)(i - i0) * blockSize + a, (size_t)(b - a));
        });
      }
      if (code) return code;
      produced = at[cnt];
      continue;
    }
    if (!inbuf) inbuf.reset(new uint8_t[(size_t)iS * NB]);
    parallel_blocks(cnt, [&](int i) { HostBitsIn t = bs; t.pos = starts[i]; t.error = false; t.getBytes(inbuf.get() + (size_t)i * iS, (uint64_t)bits[i]); });
    // decode into a temporary when the tail would overflow dst
    const int64_t room = dstCap - produced;
    if (room >= (int64_t)cnt * blockSize) {
      int rc = kz_decode_blocks(ctx, tt, (uint32_t)entropyType, blockSize, inbuf.get(), iS, bits.data(), cnt,
                                dst + produced, blockSize, res.data(), KZ_MEM_HOST);
      if (rc) return rc;
      const int64_t base = produced;
      for (int i = 0; i < cnt; i++) {
        if (res[i].status) return res[i].status;
        if (res[i].length > blockSize) return -KZ_ERR_PROCESS_BLOCK;                  // "incorrectly decompressed" (:756-759)
        // the reader appends whatever a block produced (:783-785): close up behind a short block (corrupted streams only)
        if (produced != base + (int64_t)i * blockSize && res[i].length > 0)
          memmove(dst + produced, dst + base + (int64_t)i * blockSize, (size_t)res[i].length);
        produced += res[i].length;
      }
    } else {
      std::vector<uint8_t> tmp((size_t)cnt * blockSize);
      int rc = kz_decode_blocks(ctx, tt, (uint32_t)entropyType, blockSize, inbuf.get(), iS, bits.data(), cnt,
                                tmp.data(), blockSize, res.data(), KZ_MEM_HOST);
      if (rc) return rc;
      for (int i = 0; i < cnt; i++) {
        if (res[i].status) return res[i].status;
        if (res[i].length > blockSize) return -KZ_ERR_PROCESS_BLOCK;
        if (produced + res[i].length > dstCap) return -KZ_ERR_WRITE_FILE;
        memcpy(dst + produced, tmp.data() + (size_t)i * blockSize, (size_t)res[i].length);
        produced += res[i].length;
      }
    }
  }
  return pending ? pending : produced;
}
