// kz_stream.hip -- host side of the .knz container around the HIP block pipeline: the part of
// K/io/CompressedOutputStream.java (writeHeader :236-313, ordered block emission :1024-1035, end
// marker :483-493) and K/io/CompressedInputStream.java (readHeader :359-515, block length prefix
// :1127-1129) that stays on the CPU.  Blocks go to the GPU in batches (kz_encode_blocks /
// kz_decode_blocks); only the bit-granular concatenation happens here.
#include "kz_internal.h"
#include <stdlib.h>
#include <algorithm>

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
      uint32_t acc = d[0] >> (8 - sh);                      // bits already in the partial byte
      for (uint64_t i = 0; i < full; i++) { acc = (acc << 8) | s[i]; d[i] = (uint8_t)(acc >> sh); }
      d[full] = (uint8_t)(acc << (8 - sh));
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
  // bound device memory per batch (~140 B of scratch per input byte for the BWT stage)
  int64_t budget = 48LL << 30;
  int64_t per = (int64_t)blockSize * 160 + (64 << 20);
  int nb = (int)std::max<int64_t>(1, std::min<int64_t>(256, budget / per));
  return nb;
}
}  // namespace

extern "C" int64_t kz_compress(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType, int32_t blockSize,
                               const uint8_t* src, int64_t n, uint8_t* dst, int64_t dstCap) {
  if (!ctx || !src || !dst || n < 0) return -KZ_ERR_INVALID_PARAM;
  if (blockSize < 1024 || blockSize > (1 << 30) || (blockSize & 15)) return -KZ_ERR_BLOCK_SIZE;   // :165-174
  HostBits bs{dst, dstCap, 0, false};
  // ---- stream header (CompressedOutputStream.java:236-313) ----
  bs.put(0x4B414E5A, 32); bs.put(7, 4); bs.put(0, 2);
  bs.put(entropyType, 5); bs.put(transformType, 48); bs.put((uint32_t)blockSize >> 4, 28);
  int szMask = 0;
  if (n != 0 && n < (1LL << 48)) {
    if (n >= (1LL << 32)) szMask = 3;
    else { int64_t isz = n; if (isz > (1LL << 30)) { isz >>= 4; szMask++; } szMask += (ilog2((uint32_t)isz) >> 4) + 1; }
  }
  bs.put((uint64_t)szMask, 2);
  if (szMask > 0) bs.put((uint64_t)n, 16 * szMask);
  bs.put(0, 15);
  bs.put(header_cksum(0, (int)entropyType, transformType, blockSize, szMask, n), 24);
  // ---- blocks, in batches ----
  const int64_t nblocks = (n + blockSize - 1) / blockSize;
  const int NB = batch_blocks(blockSize);
  const int64_t oS = kz_max_block_stream_bytes(blockSize);
  std::vector<uint8_t> outbuf((size_t)oS * (size_t)std::min<int64_t>(NB, std::max<int64_t>(nblocks, 1)));
  std::vector<int32_t> lens(NB);
  std::vector<kz_block_result> res(NB);
  for (int64_t b0 = 0; b0 < nblocks; b0 += NB) {
    const int cnt = (int)std::min<int64_t>(NB, nblocks - b0);
    for (int i = 0; i < cnt; i++) lens[i] = (int32_t)std::min<int64_t>(blockSize, n - (b0 + i) * blockSize);
    int rc = kz_encode_blocks(ctx, transformType, entropyType, src + b0 * blockSize, blockSize, lens.data(), cnt,
                              outbuf.data(), oS, res.data(), KZ_MEM_HOST);
    if (rc) return rc;
    for (int i = 0; i < cnt; i++) {                             // ordered emission (:1024-1035)
      if (res[i].status) return res[i].status;
      const uint64_t written = (uint64_t)res[i].bits;
      const int lw = (written < 8) ? 3 : ilog2((uint32_t)(written >> 3)) + 4;
      bs.put((uint64_t)(lw - 3), 5);
      bs.put(written, lw);
      bs.putBytes(outbuf.data() + (size_t)i * oS, written);
    }
  }
  bs.put(0, 5); bs.put(0, 3);                                   // end marker (:491-492)
  if (bs.overflow) { snprintf(ctx->err, sizeof(ctx->err), "kz_compress: destination too small"); return -KZ_ERR_WRITE_FILE; }
  return (int64_t)((bs.pos + 7) >> 3);
}

extern "C" int64_t kz_decompress(kz_ctx* ctx, const uint8_t* src, int64_t n, uint8_t* dst, int64_t dstCap) {
  if (!ctx || !src || !dst || n < 0) return -KZ_ERR_INVALID_PARAM;
  HostBitsIn bs{src, (uint64_t)n * 8, 0, false};
  if (bs.get(32) != 0x4B414E5A) return -KZ_ERR_INVALID_FILE;    // CompressedInputStream.java:359-515
  if (bs.get(4) != 7) { snprintf(ctx->err, sizeof(ctx->err), "only bitstream version 7 is supported"); return -KZ_ERR_INVALID_FILE; }
  const int chkKind = (int)bs.get(2);
  const int entropyType = (int)bs.get(5);
  const uint64_t tt = bs.get(48);
  const int blockSize = (int)(bs.get(28) << 4);
  const int szMask = (int)bs.get(2);
  int64_t inputSize = 0;
  if (szMask) inputSize = (int64_t)bs.get(16 * szMask);
  bs.get(15);
  const uint32_t ck = (uint32_t)bs.get(24);
  if (bs.error || ck != header_cksum(chkKind, entropyType, tt, blockSize, szMask, inputSize)) return -KZ_ERR_CRC_CHECK;
  if (chkKind != 0) { snprintf(ctx->err, sizeof(ctx->err), "block checksums (-x) are not supported by the HIP path"); return -KZ_ERR_INVALID_CODEC; }
  if (blockSize < 1024 || blockSize > (1 << 30)) return -KZ_ERR_BLOCK_SIZE;
  const int NB = batch_blocks(blockSize);
  const int64_t iS = (int64_t)kz_align((size_t)blockSize + (size_t)(blockSize >> 3) + 1024 + 64, 256);
  std::vector<uint8_t> inbuf((size_t)iS * NB);
  std::vector<int64_t> bits(NB);
  std::vector<kz_block_result> res(NB);
  int64_t produced = 0;
  bool done = false;
  while (!done) {
    int cnt = 0;
    while (cnt < NB) {                                          // serial walk of block length prefixes (:1127-1129)
      const int lr = (int)bs.get(5) + 3;
      const uint64_t rd = bs.get(lr);
      if (bs.error) return -KZ_ERR_READ_FILE;
      if (rd == 0) { done = true; break; }
      if ((int64_t)((rd + 7) >> 3) > iS - 64) return -KZ_ERR_BLOCK_SIZE;
      bs.getBytes(inbuf.data() + (size_t)cnt * iS, rd);
      if (bs.error) return -KZ_ERR_READ_FILE;
      bits[cnt++] = (int64_t)rd;
    }
    if (cnt == 0) break;
    if (produced + (int64_t)cnt * blockSize > dstCap + blockSize) return -KZ_ERR_WRITE_FILE;
    // decode into a temporary when the tail would overflow dst
    const int64_t room = dstCap - produced;
    if (room >= (int64_t)cnt * blockSize) {
      int rc = kz_decode_blocks(ctx, tt, (uint32_t)entropyType, blockSize, inbuf.data(), iS, bits.data(), cnt,
                                dst + produced, blockSize, res.data(), KZ_MEM_HOST);
      if (rc) return rc;
      for (int i = 0; i < cnt; i++) {
        if (res[i].status) return res[i].status;
        if (i < cnt - 1 && res[i].length != blockSize) return -KZ_ERR_PROCESS_BLOCK;
        produced += res[i].length;
      }
    } else {
      std::vector<uint8_t> tmp((size_t)cnt * blockSize);
      int rc = kz_decode_blocks(ctx, tt, (uint32_t)entropyType, blockSize, inbuf.data(), iS, bits.data(), cnt,
                                tmp.data(), blockSize, res.data(), KZ_MEM_HOST);
      if (rc) return rc;
      for (int i = 0; i < cnt; i++) {
        if (res[i].status) return res[i].status;
        if (produced + res[i].length > dstCap) return -KZ_ERR_WRITE_FILE;
        memcpy(dst + produced, tmp.data() + (size_t)i * blockSize, (size_t)res[i].length);
        produced += res[i].length;
      }
    }
  }
  return produced;
}
