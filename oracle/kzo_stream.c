/* ORACLE (test infrastructure only; parity unpinned -- see kzo.h).
 * Transform chain, block framing and .knz stream container restated from
 *   K/transform/Sequence.java:56-127 (forward), :137-207 (inverse)
 *   K/transform/TransformFactory.java:240-266 (newFunction), :273-351 (token -> codec)
 *   K/io/CompressedOutputStream.java:236-313 (stream header), :733-1054 (encodeBlock), :483-493 (end marker)
 *   K/io/CompressedInputStream.java:359-515 (readHeader), :1025-1100 (readBlockHeader), :1106-1378 (decodeBlock)
 */
#include "kzo.h"
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

static int ilog2(uint32_t x) { return 31 - __builtin_clz(x); }

/* ---------------- dispatch ---------------- */
/* The two entries of the context map that TEXT reads besides "dataType": "entropy" selects TextCodec1 / TextCodec2
 * (TransformFactory.java:275-286) and "blockSize" sizes its hash map (TextCodec.java:561-575,1068-1081).  Per thread: set by the
 * block coder before it runs a Sequence, or by kzo_set_transform_ctx for single-transform calls. */
static __thread int tls_entropy = KZO_E_NONE;
static __thread int tls_block_size = 4 * 1024 * 1024;
void kzo_set_transform_ctx(int entropyType, int blockSize) { tls_entropy = entropyType; tls_block_size = blockSize; }
static int text_codec_type(void) {
  return (tls_entropy == KZO_E_NONE || tls_entropy == KZO_E_ANS0 || tls_entropy == KZO_E_HUFFMAN || tls_entropy == KZO_E_RANGE) ? 2 : 1;
}

int kzo_transform_max_encoded_len(int type, int n) {
  switch (type) {
    case KZO_T_UTF: return n + 8192;                             /* UTFCodec.java:308-310 */
    case KZO_T_BWT: return n + 33;                               /* BWTBlockCodec.java:40,222 */
    case KZO_T_SRT: return n + 1024;                             /* SRT.java:30,365 */
    case KZO_T_LZ: case KZO_T_LZX:
      return ((n <= 1024) ? n + 16 : n + (n / 64)) + 2;          /* LZCodec.java:961-964 */
    case KZO_T_MM: return kzo_fsd_max_encoded_len(n);            /* FSDCodec.java:320-323 */
    case KZO_T_PACK: case KZO_T_DNA: return kzo_alias_max_encoded_len(n);   /* AliasCodec.java:472-475 */
    default: return n;                                           /* ZRLT.java:243, SBRT.java:224, NullTransform */
  }
}

int kzo_transform_forward(int type, int* dataType, const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  switch (type) {
    case KZO_T_NONE: if (dstCap < n) return 0; memcpy(dst, src, (size_t)n); *produced = n; return 1;
    case KZO_T_BWT:  return kzo_bwt_forward(src, n, dst, dstCap, produced);
    case KZO_T_RANK: if (dstCap < n) return 0; *produced = n; return kzo_sbrt_forward(2, src, n, dst);
    case KZO_T_MTFT: if (dstCap < n) return 0; *produced = n; return kzo_sbrt_forward(1, src, n, dst);
    case KZO_T_ZRLT: return kzo_zrlt_forward(src, n, dst, dstCap, produced);
    case KZO_T_SRT:  return kzo_srt_forward(src, n, dst, dstCap, produced);
    case KZO_T_LZ:   return kzo_lz_forward(0, dataType ? *dataType : KZO_DT_UNDEFINED, src, n, dst, dstCap, produced);
    case KZO_T_LZX:  return kzo_lz_forward(1, dataType ? *dataType : KZO_DT_UNDEFINED, src, n, dst, dstCap, produced);
    case KZO_T_MM:   return kzo_fsd_forward(dataType, src, n, dst, dstCap, produced);
    case KZO_T_PACK: return kzo_alias_forward(0, dataType, src, n, dst, dstCap, produced);
    case KZO_T_DNA:  return kzo_alias_forward(1, dataType, src, n, dst, dstCap, produced);   /* TransformFactory.java:341-343 */
    case KZO_T_TEXT: return kzo_text_forward(text_codec_type(), tls_block_size, dataType, src, n, dst, dstCap, produced);
    case KZO_T_UTF:  return kzo_utf_forward(dataType, src, n, dst, dstCap, produced);
    default: return 0;
  }
}

int kzo_transform_inverse(int type, const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced) {
  *produced = 0;
  switch (type) {
    case KZO_T_NONE: if (dstCap < n) return 0; memcpy(dst, src, (size_t)n); *produced = n; return 1;
    case KZO_T_BWT:  return kzo_bwt_inverse(src, n, dst, dstCap, produced);
    case KZO_T_RANK: if (dstCap < n) return 0; *produced = n; return kzo_sbrt_inverse(2, src, n, dst);
    case KZO_T_MTFT: if (dstCap < n) return 0; *produced = n; return kzo_sbrt_inverse(1, src, n, dst);
    case KZO_T_ZRLT: return kzo_zrlt_inverse(src, n, dst, dstCap, produced);
    case KZO_T_SRT:  return kzo_srt_inverse(src, n, dst, dstCap, produced);
    case KZO_T_LZ:   return kzo_lz_inverse(0, src, n, dst, dstCap, produced);
    case KZO_T_LZX:  return kzo_lz_inverse(1, src, n, dst, dstCap, produced);
    case KZO_T_MM:   return kzo_fsd_inverse(src, n, dst, dstCap, produced);
    case KZO_T_PACK: case KZO_T_DNA: return kzo_alias_inverse(src, n, dst, dstCap, produced);
    case KZO_T_TEXT: return kzo_text_inverse(tls_block_size, src, n, dst, dstCap, produced);
    case KZO_T_UTF:  return kzo_utf_inverse(src, n, dst, dstCap, produced);
    default: return 0;
  }
}

int kzo_entropy_encode(int type, kzo_obs* s, const uint8_t* block, int count) {
  switch (type) {
    case KZO_E_NONE: return kzo_null_encode(s, block, count);
    case KZO_E_ANS0: return kzo_ans0_encode(s, block, count);
    case KZO_E_HUFFMAN: return kzo_huffman_encode(s, block, count);
    case KZO_E_FPAQ: return kzo_fpaq_encode(s, block, count);
    default: return -1;
  }
}
int kzo_entropy_decode(int type, kzo_ibs* s, uint8_t* block, int count) {
  switch (type) {
    case KZO_E_NONE: return kzo_null_decode(s, block, count);
    case KZO_E_ANS0: return kzo_ans0_decode(s, block, count);
    case KZO_E_HUFFMAN: return kzo_huffman_decode(s, block, count);
    case KZO_E_FPAQ: return kzo_fpaq_decode(s, block, count);
    default: return -1;
  }
}

/* ---------------- Sequence ---------------- */
uint64_t kzo_transform_type(const int* types, int nb) {        /* TransformFactory.java:132-158 */
  uint64_t t = 0;
  for (int i = 0; i < 8; i++) t = (t << 6) | (uint64_t)((i < nb) ? (types[i] & 0x3F) : 0);
  return t;
}

static int split_types(uint64_t transformType, int* types) {   /* TransformFactory.java:240-266 */
  int nbtr = 0;
  for (int i = 0; i < 8; i++)
    if (((transformType >> (42 - 6 * i)) & 0x3F) != KZO_T_NONE) nbtr++;
  if (nbtr == 0) nbtr = 1;
  int k = 0;
  for (int i = 0; i < nbtr; i++) {             /* note: the reference loops i < transforms.length */
    int t = (int)((transformType >> (42 - 6 * i)) & 0x3F);
    if ((t != KZO_T_NONE) || (i == 0)) types[k++] = t;
  }
  /* slots the reference leaves null are never reached for well-formed chains (no NONE gaps) */
  return k;
}

static int seq_max_encoded_len(const int* types, int nb, int n) {   /* Sequence.java:215-226 */
  int req = n;
  for (int i = 0; i < nb; i++) { int m = kzo_transform_max_encoded_len(types[i], req); if (m > req) req = m; }
  return req;
}

int kzo_sequence_forward(const int* types, int nb, int* dataType, const uint8_t* src, int n,
                         uint8_t* dst, int dstCap, uint8_t* skipFlagsOut) {
  uint8_t skipFlags = 0xFF;
  *skipFlagsOut = skipFlags;
  if (n == 0) return 0;
  int required = seq_max_encoded_len(types, nb, n);
  uint8_t* bufA = (uint8_t*)malloc((size_t)required + 64);
  uint8_t* bufB = (uint8_t*)malloc((size_t)required + 64);
  memcpy(bufA, src, (size_t)n);
  uint8_t *in = bufA, *out = bufB;
  int count = n;
  for (int i = 0; i < nb; i++) {
    int produced = 0;
    /* a transform that returns false leaves the data untouched (Sequence.java:95-105); one that throws (LZ's token buffer) ends
       the block: nothing catches it before EncodingTask.call (-13 = ERR_PROCESS_BLOCK) */
    const int tr = kzo_transform_forward(types[i], dataType, in, count, out, required, &produced);
    if (tr < 0) { free(bufA); free(bufB); return -13; }
    if (!tr) continue;
    skipFlags &= (uint8_t)~(1 << (7 - i));
    count = produced;
    uint8_t* t = in; in = out; out = t;
  }
  int ret = count;
  if (count > dstCap) { skipFlags = 0xFF; ret = -1; }
  else memcpy(dst, in, (size_t)count);
  free(bufA); free(bufB);
  *skipFlagsOut = skipFlags;
  return ret;
}

int kzo_sequence_inverse(const int* types, int nb, uint8_t skipFlags, const uint8_t* src, int n,
                         uint8_t* dst, int dstCap) {
  if (n == 0) return 0;
  if (skipFlags == 0xFF) { if (n > dstCap) return -1; memcpy(dst, src, (size_t)n); return n; }
  int cap = dstCap > n ? dstCap : n;
  uint8_t* bufA = (uint8_t*)malloc((size_t)cap + 64);
  uint8_t* bufB = (uint8_t*)malloc((size_t)cap + 64);
  memcpy(bufA, src, (size_t)n);
  uint8_t *in = bufA, *out = bufB;
  int count = n, ok = 1;
  for (int i = nb - 1; i >= 0; i--) {
    if (skipFlags & (1 << (7 - i))) continue;
    int produced = 0;
    /* sa2.length = dst.array.length (Sequence.java:168) */
    ok = kzo_transform_inverse(types[i], in, count, out, dstCap, &produced);
    count = produced;
    if (!ok) break;
    uint8_t* t = in; in = out; out = t;
  }
  int ret = -1;
  if (ok && count <= dstCap) { memcpy(dst, in, (size_t)count); ret = count; }
  free(bufA); free(bufB);
  return ret;
}

/* ---------------- block ---------------- */
static uint32_t mix32(uint32_t c, uint32_t h, uint32_t v) {     /* CompressedOutputStream.java:89-93 */
  c ^= h * ~v;
  c = (c << 13) | (c >> 19);
  return c * 5u + 0x52DCE729u;
}

static uint8_t block_hdr_cksum(uint8_t mode, uint8_t headerSkipFlags, uint32_t postLen, uint64_t written) {
  const uint32_t HASH = 0x1E35A7BDu;                             /* :977-985 */
  uint32_t c = HASH * 0x01030507u;
  c = mix32(c, HASH, mode);
  c = mix32(c, HASH, headerSkipFlags);
  c = mix32(c, HASH, postLen);
  c = mix32(c, HASH, (uint32_t)(written >> 32));
  c = mix32(c, HASH, (uint32_t)written);
  c = (c >> 23) ^ (c >> 3);
  return (uint8_t)c;
}

/* Data-type side channel.  The writer tags a block BIN / MULTIMEDIA / EXE from its first four bytes (K/Magic.java,
 * CompressedOutputStream.java:795-804, kzo_block_data_type); the tag travels through the Sequence as the context
 * entry "dataType": MM (FSDCodec) reads and rewrites it, LZ/LZX reacts to DNA and SMALL_ALPHABET (LZCodec.java:343-352). */
int64_t kzo_encode_block(uint64_t transformType, int entropyType, const uint8_t* data, int n,
                         uint8_t* out, size_t outCap, uint8_t* skipFlagsOut, int* postLenOut) {
  return kzo_encode_block_x(transformType, entropyType, 0, data, n, out, outCap, skipFlagsOut, postLenOut);
}

/* chkKind: 0 none, 1 XXHash32, 2 XXHash64 of the ORIGINAL block (CompressedOutputStream.java:749-755,887-891) */
int64_t kzo_encode_block_x(uint64_t transformType, int entropyType, int chkKind, const uint8_t* data, int n,
                           uint8_t* out, size_t outCap, uint8_t* skipFlagsOut, int* postLenOut) {
  return kzo_encode_block_y(transformType, entropyType, chkKind, 4 * 1024 * 1024, data, n, out, outCap, skipFlagsOut, postLenOut);
}

/* blockSize: the stream's block size = the context entry "blockSize" (only TEXT looks at it) */
int64_t kzo_encode_block_y(uint64_t transformType, int entropyType, int chkKind, int blockSize, const uint8_t* data, int n,
                           uint8_t* out, size_t outCap, uint8_t* skipFlagsOut, int* postLenOut) {
  if (n == 0) return 0;
  kzo_set_transform_ctx(entropyType, blockSize);
  const int skipBlocks = (chkKind >> 8) & 1;                     /* bit 8 of chkKind: the writer's "skipBlocks" option */
  chkKind &= 0xFF;
  uint64_t checksum = 0;
  if (chkKind == 1) checksum = kzo_xxhash32(data, n, 0x4B414E5Au);
  else if (chkKind == 2) checksum = kzo_xxhash64(data, n, 0x4B414E5AULL);
  uint8_t mode = 0;
  int types[8];
  if (n <= 15) { transformType = 0; entropyType = KZO_E_NONE; mode |= 0x80; }   /* :764-767 */
  else if (skipBlocks) {                                                         /* :769-788 */
    int skip = kzo_magic_is_compressed(kzo_magic_type(data));
    if (!skip) {
      int histo[256];
      memset(histo, 0, sizeof(histo));
      for (int i = 0; i < n; i++) histo[data[i]]++;
      skip = kzo_entropy1024(n, histo) >= 973;                                   /* EntropyUtils.INCOMPRESSIBLE_THRESHOLD */
    }
    if (skip) { transformType = 0; entropyType = KZO_E_NONE; mode |= 0x80; }
  }
  int nb = split_types(transformType, types);
  int required = seq_max_encoded_len(types, nb, n);
  uint8_t* buffer = (uint8_t*)malloc((size_t)required + 64);
  uint8_t skipFlags = 0xFF;
  int dataType = kzo_block_data_type(data, n);                                     /* :795-804 */
  int postLen = kzo_sequence_forward(types, nb, &dataType, data, n, buffer, required, &skipFlags);
  if (postLen < 0) { free(buffer); return postLen == -13 ? -13 : -1; }
  int dataSize = (postLen < 256) ? 1 : (ilog2((uint32_t)postLen) >> 3) + 1;       /* :825-826 */
  mode |= (uint8_t)(((dataSize - 1) & 3) << 5);
  kzo_obs os; kzo_obs_init(&os, (size_t)n + (n >> 3) + 1024);
  uint8_t headerSkipFlags = skipFlags;
  if ((mode & 0x80) || (nb <= 4)) {                                               /* :864-878 */
    mode |= (uint8_t)(skipFlags >> 4);
    headerSkipFlags = (mode & 0x80) ? 0 : (uint8_t)(((mode << 4) | 0x0F) & 0xFF);
    kzo_obs_write(&os, mode, 8);
  } else {
    mode |= 0x10;
    kzo_obs_write(&os, mode, 8);
    kzo_obs_write(&os, skipFlags, 8);
  }
  kzo_obs_write(&os, (uint32_t)postLen, 8 * dataSize);
  int headerChecksumIndex = 1 + dataSize;
  if (!(mode & 0x80) && (nb > 4)) headerChecksumIndex++;
  kzo_obs_write(&os, 0, 8);
  if (chkKind == 1) kzo_obs_write(&os, checksum, 32); else if (chkKind == 2) kzo_obs_write(&os, checksum, 64);
  if (kzo_entropy_encode(entropyType, &os, buffer, postLen) != postLen) { kzo_obs_free(&os); free(buffer); return -1; }
  uint64_t written = os.nbits;
  if (!(mode & 0x80)) {                                                           /* :926-973 raw fallback */
    uint64_t entropyBytes = (written + 7) >> 3;
    if ((uint64_t)postLen < entropyBytes) {
      kzo_obs_free(&os); kzo_obs_init(&os, (size_t)postLen + 64);
      uint8_t copyMode = (uint8_t)(mode | 0x80 | 0x10);
      kzo_obs_write(&os, copyMode, 8);
      if (nb > 4) kzo_obs_write(&os, skipFlags, 8);
      kzo_obs_write(&os, (uint32_t)postLen, 8 * dataSize);
      headerChecksumIndex = 1 + dataSize;
      if (nb > 4) { headerChecksumIndex++; headerSkipFlags = skipFlags; }
      else headerSkipFlags = (uint8_t)(((copyMode << 4) | 0x0F) & 0xFF);
      kzo_obs_write(&os, 0, 8);
      if (chkKind == 1) kzo_obs_write(&os, checksum, 32); else if (chkKind == 2) kzo_obs_write(&os, checksum, 64);
      kzo_obs_write_bytes(&os, buffer, (uint64_t)postLen * 8);
      written = os.nbits;
      mode = copyMode;
    }
  }
  os.buf[headerChecksumIndex] = block_hdr_cksum(mode, headerSkipFlags, (uint32_t)postLen, written);
  size_t nbytes = (size_t)((written + 7) >> 3);
  int64_t ret = (int64_t)written;
  if (nbytes > outCap) ret = -1; else memcpy(out, os.buf, nbytes);
  if (skipFlagsOut) *skipFlagsOut = skipFlags;
  if (postLenOut) *postLenOut = postLen;
  kzo_obs_free(&os); free(buffer);
  return ret;
}

/* in = the block's private stream (header included), nbits = W. Returns decoded length or <0. */
int kzo_decode_block(uint64_t transformType, int entropyType, int blockSize, const uint8_t* in,
                     int64_t nbits, uint8_t* out, int outCap) {
  return kzo_decode_block_x(transformType, entropyType, 0, blockSize, in, nbits, out, outCap);
}

static int decode_block_impl(uint64_t transformType, int entropyType, int chkKind, int blockSize, const uint8_t* in,
                             int64_t nbits, uint8_t* out, int outCap, int headerOnly);
int kzo_decode_block_x(uint64_t transformType, int entropyType, int chkKind, int blockSize, const uint8_t* in,
                       int64_t nbits, uint8_t* out, int outCap) {
  return decode_block_impl(transformType, entropyType, chkKind, blockSize, in, nbits, out, outCap, 0);
}
/* headerOnly: the version 7 block header is parsed and verified on the shared stream before the payload is read
   (CompressedInputStream.java:1142-1170), so its faults come before a truncated payload's */
static int decode_block_impl(uint64_t transformType, int entropyType, int chkKind, int blockSize, const uint8_t* in,
                             int64_t nbits, uint8_t* out, int outCap, int headerOnly) {
  if (nbits < 8) return -2;                                      /* ERR_BLOCK_SIZE :1027-1028 */
  kzo_ibs is; kzo_ibs_init(&is, in, (uint64_t)nbits);
  int types[8];
  uint8_t mode = (uint8_t)kzo_ibs_read(&is, 8);
  uint8_t skipFlags = 0;
  int hasSkipFlags = 0, transformedCopy = 0, rawCopy = 0;
  const int copyBlock = (mode & 0x80) != 0;
  if (copyBlock) {                                               /* CompressedInputStream.java:1036-1052 */
    if (mode & 0x10) {
      transformedCopy = 1;
      int nbf = split_types(transformType, types);
      if (nbf > 4) hasSkipFlags = 1; else skipFlags = (uint8_t)((mode << 4) | 0x0F);
    } else rawCopy = 1;
  } else if (mode & 0x10) hasSkipFlags = 1;
  else skipFlags = (uint8_t)((mode << 4) | 0x0F);
  const int dataSize = 1 + ((mode >> 5) & 3);
  const int headerSize = 1 + hasSkipFlags + dataSize + 1;
  if (nbits < (headerSize << 3)) return -2;                      /* ERR_BLOCK_SIZE :1056-1057 */
  if (hasSkipFlags) skipFlags = (uint8_t)kzo_ibs_read(&is, 8);
  uint32_t preLen = (uint32_t)kzo_ibs_read(&is, 8 * dataSize);
  uint8_t ck = (uint8_t)kzo_ibs_read(&is, 8);
  uint8_t hsf = hasSkipFlags ? skipFlags : (rawCopy ? 0 : (uint8_t)(((mode << 4) | 0x0F) & 0xFF));
  if (ck != block_hdr_cksum(mode, hsf, preLen, (uint64_t)nbits)) return -19;   /* ERR_CRC_CHECK */
  int maxTL = blockSize + blockSize / 2; if (maxTL < 2048) maxTL = 2048;
  if ((int)preLen < 0 || (int)preLen > maxTL) return -11;        /* ERR_READ_FILE :1151-1156 */
  {                                                              /* ERR_BLOCK_SIZE :1158-1165 */
    const int64_t checksumSize = chkKind == 2 ? 8 : (chkKind == 1 ? 4 : 0);
    if ((nbits + 7) >> 3 > (int64_t)preLen + headerSize + checksumSize) return -2;
  }
  if (headerOnly) return 0;
  if (preLen == 0) return 0;
  kzo_set_transform_ctx(entropyType, blockSize);
  uint64_t checksum1 = 0;
  if (chkKind == 1) checksum1 = kzo_ibs_read(&is, 32); else if (chkKind == 2) checksum1 = kzo_ibs_read(&is, 64);   /* :1256-1262 */
  if (rawCopy) { transformType = 0; entropyType = KZO_E_NONE; skipFlags = 0xFF; }
  else if (transformedCopy) entropyType = KZO_E_NONE;
  uint8_t* buffer = (uint8_t*)calloc((size_t)preLen + 1024, 1);   /* zeroed: see kzo_ans0_decode on unwritten tails */
  int ret = -13;                                                  /* ERR_PROCESS_BLOCK */
  if (kzo_entropy_decode(entropyType, &is, buffer, (int)preLen) == (int)preLen && !is.error) {
    int nb = split_types(transformType, types);
    /* decoder's data buffer: blockSize + max(512, blockSize/16) (CompressedInputStream.java:694-695) */
    int cap = blockSize + ((blockSize >> 4) > 512 ? (blockSize >> 4) : 512);
    uint8_t* tmp = (uint8_t*)malloc((size_t)cap + 64);
    int r = kzo_sequence_inverse(types, nb, skipFlags, buffer, (int)preLen, tmp, cap);
    if (r >= 0) {
      /* checksum first (:1349-1363), then the reader's "decoded > blockSize" test (:756-759) */
      if (chkKind == 1 && (uint32_t)checksum1 != kzo_xxhash32(tmp, r, 0x4B414E5Au)) ret = -19;      /* ERR_CRC_CHECK */
      else if (chkKind == 2 && checksum1 != kzo_xxhash64(tmp, r, 0x4B414E5AULL)) ret = -19;
      else if (r > blockSize) ret = -13;                                                         /* ERR_PROCESS_BLOCK */
      else if (r > outCap) ret = -12;                                                            /* destination too small (ERR_WRITE_FILE) */
      else { memcpy(out, tmp, (size_t)r); ret = r; }
    }
    free(tmp);
  }
  free(buffer);
  return ret;
}

/* ---------------- stream ---------------- */
int kzo_stream_header(uint64_t transformType, int entropyType, int blockSize, int chkKind,
                      int64_t inputSize, uint8_t* out) {          /* CompressedOutputStream.java:236-313 */
  kzo_obs s; kzo_obs_wrap(&s, out, 40);
  kzo_obs_write(&s, 0x4B414E5A, 32);
  kzo_obs_write(&s, 7, 4);
  kzo_obs_write(&s, (uint64_t)chkKind, 2);
  kzo_obs_write(&s, (uint64_t)entropyType, 5);
  kzo_obs_write(&s, transformType, 48);
  kzo_obs_write(&s, (uint64_t)((uint32_t)blockSize >> 4), 28);
  int szMask = 0;
  if ((inputSize != 0) && (inputSize < (1LL << 48))) {
    if (inputSize >= (1LL << 32)) szMask = 3;
    else {
      int64_t isz = inputSize;
      if (isz > (1LL << 30)) { isz >>= 4; szMask++; }
      szMask += ((ilog2((uint32_t)isz) >> 4) + 1);
    }
  }
  kzo_obs_write(&s, (uint64_t)szMask, 2);
  if (szMask > 0) kzo_obs_write(&s, (uint64_t)inputSize, 16 * szMask);
  kzo_obs_write(&s, 0, 15);
  const uint32_t HASH = 0x1E35A7BDu;
  uint32_t c = HASH * (0x01030507u * 7u);
  c = mix32(c, HASH, (uint32_t)chkKind);
  c = mix32(c, HASH, (uint32_t)entropyType);
  c = mix32(c, HASH, (uint32_t)(transformType >> 32));
  c = mix32(c, HASH, (uint32_t)transformType);
  c = mix32(c, HASH, (uint32_t)blockSize);
  if (szMask > 0) { c = mix32(c, HASH, (uint32_t)((uint64_t)inputSize >> 32)); c = mix32(c, HASH, (uint32_t)inputSize); }
  c = (c >> 23) ^ (c >> 3);
  kzo_obs_write(&s, c, 24);
  return (int)(s.nbits >> 3);
}

typedef struct {
  uint64_t transformType; int entropyType; int blockSize; int chkKind;
  const uint8_t* src; int64_t n; int nblocks;
  uint8_t** outs; int64_t* bits; int* next; pthread_mutex_t* mu; int fail;
} enc_job;

static void* enc_worker(void* arg) {
  enc_job* j = (enc_job*)arg;
  for (;;) {
    pthread_mutex_lock(j->mu);
    int b = (*j->next)++;
    pthread_mutex_unlock(j->mu);
    if (b >= j->nblocks) break;
    int64_t off = (int64_t)b * j->blockSize;
    int len = (int)((j->n - off) < j->blockSize ? (j->n - off) : j->blockSize);
    size_t cap = (size_t)j->blockSize + ((size_t)j->blockSize >> 3) + 1024;   /* its slot of the slab */
    j->bits[b] = kzo_encode_block_y(j->transformType, j->entropyType, j->chkKind, j->blockSize, j->src + off, len, j->outs[b], cap, NULL, NULL);
    if (j->bits[b] < 0) j->fail = (j->bits[b] == -13) ? 13 : 1;
  }
  return NULL;
}

int64_t kzo_compress(uint64_t transformType, int entropyType, int blockSize, const uint8_t* src,
                     int64_t n, uint8_t* dst, int64_t dstCap, int jobs) {
  return kzo_compress_x(transformType, entropyType, blockSize, 0, src, n, dst, dstCap, jobs);
}

int64_t kzo_compress_x(uint64_t transformType, int entropyType, int blockSize, int chkKind, const uint8_t* src,
                       int64_t n, uint8_t* dst, int64_t dstCap, int jobs) {
  int nblocks = (int)((n + blockSize - 1) / blockSize);
  uint8_t** outs = (uint8_t**)calloc((size_t)nblocks + 1, sizeof(uint8_t*));
  int64_t* bits = (int64_t*)calloc((size_t)nblocks + 1, sizeof(int64_t));
  /* one slab for the block streams (only the bytes written are ever touched) instead of one allocation per block */
  const size_t slot = (size_t)blockSize + ((size_t)blockSize >> 3) + 1024;
  uint8_t* slab = (uint8_t*)malloc(slot * (size_t)(nblocks > 0 ? nblocks : 1));
  for (int b = 0; b < nblocks; b++) outs[b] = slab + slot * (size_t)b;
  pthread_mutex_t mu; pthread_mutex_init(&mu, NULL);
  int next = 0;
  enc_job job = { transformType, entropyType, blockSize, chkKind, src, n, nblocks, outs, bits, &next, &mu, 0 };
  if (jobs < 1) jobs = 1;
  if (jobs > 256) jobs = 256;
  pthread_t th[256];
  for (int t = 0; t < jobs; t++) pthread_create(&th[t], NULL, enc_worker, &job);
  for (int t = 0; t < jobs; t++) pthread_join(th[t], NULL);
  int64_t ret = -1;
  if (!job.fail) {
    kzo_obs s; kzo_obs_init(&s, (size_t)(n / 2) + 4096);
    uint8_t hdr[40];
    int hl = kzo_stream_header(transformType, entropyType, blockSize, chkKind & 0xFF, n, hdr);
    kzo_obs_write_bytes(&s, hdr, (uint64_t)hl * 8);
    for (int b = 0; b < nblocks; b++) {                            /* :1024-1035 */
      uint64_t written = (uint64_t)bits[b];
      int lw = (written < 8) ? 3 : ilog2((uint32_t)(written >> 3)) + 4;
      kzo_obs_write(&s, (uint64_t)(lw - 3), 5);
      kzo_obs_write(&s, written, lw);
      kzo_obs_write_bytes(&s, outs[b], written);
    }
    kzo_obs_write(&s, 0, 5); kzo_obs_write(&s, 0, 3);              /* :491-492 end marker */
    int64_t nbytes = (int64_t)((s.nbits + 7) >> 3);
    if (nbytes <= dstCap) { memcpy(dst, s.buf, (size_t)nbytes); ret = nbytes; }
    kzo_obs_free(&s);
  }
  if (job.fail == 13) ret = -13;                                   /* a block's transform threw: ERR_PROCESS_BLOCK */
  free(slab);
  free(outs); free(bits); pthread_mutex_destroy(&mu);
  return ret;
}

typedef struct {
  uint64_t transformType; int entropyType; int blockSize; int chkKind; int nblocks;
  uint8_t** ins; int64_t* bits; uint8_t* dst; int64_t dstCap; int* lens;
  int* next; pthread_mutex_t* mu; int fail;
  uint8_t** spill;                                                 /* blocks decoded aside (see dec_worker) */
} dec_job;

static void* dec_worker(void* arg) {
  dec_job* j = (dec_job*)arg;
  for (;;) {
    pthread_mutex_lock(j->mu);
    int b = (*j->next)++;
    pthread_mutex_unlock(j->mu);
    if (b >= j->nblocks) break;
    int64_t off = (int64_t)b * j->blockSize;
    int64_t cap = j->dstCap - off; if (cap > j->blockSize) cap = j->blockSize;
    if (cap < j->blockSize) {
      /* the destination's tail is shorter than a block.  The reader has no such notion -- it appends what every block produced
         (CompressedInputStream.java:783-785) -- so a block that comes out longer than its slot here (damaged streams: the blocks in
         front of it came out short) is decoded aside and placed when the lengths are known; only the TOTAL can be too long */
      uint8_t* tmpb = (uint8_t*)malloc((size_t)j->blockSize + 64);
      int r = kzo_decode_block_x(j->transformType, j->entropyType, j->chkKind, j->blockSize, j->ins[b], j->bits[b], tmpb, j->blockSize);
      j->lens[b] = r;
      if (r < 0) { j->fail = 1; free(tmpb); }
      else if (r <= cap) { if (r > 0) memcpy(j->dst + off, tmpb, (size_t)r); free(tmpb); }
      else j->spill[b] = tmpb;
      continue;
    }
    int r = kzo_decode_block_x(j->transformType, j->entropyType, j->chkKind, j->blockSize, j->ins[b], j->bits[b], j->dst + off, (int)cap);
    j->lens[b] = r;
    if (r < 0) j->fail = 1;
  }
  return NULL;
}

int64_t kzo_decompress(const uint8_t* src, int64_t n, uint8_t* dst, int64_t dstCap, int jobs) {
  kzo_ibs s; kzo_ibs_init(&s, src, (uint64_t)n * 8);
  /* header checks in the reference's order, returning -(Error.java code) (CompressedInputStream.java:359-478) */
  if (kzo_ibs_read(&s, 32) != 0x4B414E5A) return -15;              /* ERR_INVALID_FILE :367-368 */
  int version = (int)kzo_ibs_read(&s, 4);
  if (version != 7) return -16;                                    /* ERR_STREAM_VERSION :374-377 (older layouts not restated) */
  int chkKind = (int)kzo_ibs_read(&s, 2);
  if (chkKind > 2) return -15;                                     /* :390-392 */
  int entropyType = (int)kzo_ibs_read(&s, 5);
  if (entropyType == 3 || entropyType > 9) return -3;              /* ERR_INVALID_CODEC, EntropyCodecFactory.getName :212-243 */
  uint64_t transformType = kzo_ibs_read(&s, 48);
  for (int i = 0; i < 8; i++) {                                    /* TransformFactory.getName :368-449 */
    int t = (int)((transformType >> (42 - 6 * i)) & 0x3F);
    if (t == 4 || t > 19) return -3;
  }
  int blockSize = (int)(kzo_ibs_read(&s, 28) << 4);
  if (blockSize < 1024 || blockSize > (1 << 30)) return -2;        /* ERR_BLOCK_SIZE :419-422 */
  int szMask = (int)kzo_ibs_read(&s, 2);
  int64_t inputSize = 0;
  if (szMask) inputSize = (int64_t)kzo_ibs_read(&s, 16 * szMask);
  kzo_ibs_read(&s, 15);
  uint32_t ck = (uint32_t)kzo_ibs_read(&s, 24);
  uint8_t hdr[40];
  int hl = kzo_stream_header(transformType, entropyType, blockSize, chkKind, inputSize, hdr);
  uint32_t ck2 = ((uint32_t)hdr[hl - 3] << 16) | ((uint32_t)hdr[hl - 2] << 8) | hdr[hl - 1];
  if (ck != ck2 || s.error) return -19;                            /* ERR_CRC_CHECK :477-478 */
  /* serial index pass over block length prefixes (decodeBlock :1127-1129) */
  int capBlocks = 1024, nblocks = 0;
  uint8_t** ins = (uint8_t**)malloc(sizeof(uint8_t*) * (size_t)capBlocks);
  int64_t* bits = (int64_t*)malloc(sizeof(int64_t) * (size_t)capBlocks);
  int bad = 0, badCode = -11;
  for (;;) {
    int lr = (int)kzo_ibs_read(&s, 5) + 3;
    uint64_t read = kzo_ibs_read(&s, lr);
    if (s.error) { bad = 1; break; }
    if (read == 0) break;
    const uint64_t startPos = s.pos;
    if (nblocks == capBlocks) {
      capBlocks *= 2;
      ins = (uint8_t**)realloc(ins, sizeof(uint8_t*) * (size_t)capBlocks);
      bits = (int64_t*)realloc(bits, sizeof(int64_t) * (size_t)capBlocks);
    }
    if (read > s.nbits - startPos) {                               /* stream ends inside this block: header faults first, else ERR_READ_FILE */
      badCode = -11;
      const uint64_t have = s.nbits - startPos;                    /* bits of the block that exist */
      if (have >= 8) {
        uint8_t hb[16];
        memset(hb, 0, sizeof(hb));
        kzo_ibs_read_bytes(&s, hb, have < 64 ? have : 64);
        int tt[8];
        const uint8_t mode = hb[0];
        int hasSkip = 0;
        if (mode & 0x80) { if ((mode & 0x10) && split_types(transformType, tt) > 4) hasSkip = 1; }
        else if (mode & 0x10) hasSkip = 1;
        const uint64_t headerBits = 8ULL * (uint64_t)(1 + hasSkip + 1 + ((mode >> 5) & 3) + 1);
        if (read < headerBits || have >= headerBits) {             /* the header is whole (or the declared length is too short for one) */
          int hc = decode_block_impl(transformType, entropyType, chkKind, blockSize, hb, (int64_t)read, NULL, 0, 1);
          if (hc < 0) badCode = hc;
        }
      }
      bad = 1; break;
    }
    ins[nblocks] = (uint8_t*)malloc((size_t)((read + 7) >> 3) + 8);
    kzo_ibs_read_bytes(&s, ins[nblocks], read);
    bits[nblocks] = (int64_t)read;
    nblocks++;
  }
  int64_t ret = badCode;                                           /* stream ends inside a block */
  {
    int* lens = (int*)calloc((size_t)nblocks + 1, sizeof(int));
    pthread_mutex_t mu; pthread_mutex_init(&mu, NULL);
    int next = 0;
    uint8_t** spill = (uint8_t**)calloc((size_t)nblocks + 1, sizeof(uint8_t*));
    dec_job job = { transformType, entropyType, blockSize, chkKind, nblocks, ins, bits, dst, dstCap, lens, &next, &mu, 0, spill };
    if (jobs < 1) jobs = 1;
    if (jobs > 256) jobs = 256;
    pthread_t th[256];
    for (int t = 0; t < jobs; t++) pthread_create(&th[t], NULL, dec_worker, &job);
    for (int t = 0; t < jobs; t++) pthread_join(th[t], NULL);
    if (!job.fail) {
      /* blocks were decoded at blockSize strides; the reader simply appends what each block produced (:783-785), so
         a block that came out short (possible only for corrupted streams without checksums) is closed up */
      ret = 0;
      for (int b = 0; b < nblocks; b++) {
        if (ret + lens[b] > dstCap) { ret = -12; break; }            /* destination too small (ERR_WRITE_FILE) */
        if (spill[b]) memcpy(dst + ret, spill[b], (size_t)lens[b]);
        else if (ret != (int64_t)b * blockSize && lens[b] > 0) memmove(dst + ret, dst + (int64_t)b * blockSize, (size_t)lens[b]);
        ret += lens[b];
      }
      if (bad && ret >= 0) ret = badCode;                          /* every whole block decoded; the fault is the truncated one after them */
    } else {                                                       /* code of the first failing block, as the reader would surface it */
      ret = -13;
      for (int b = 0; b < nblocks; b++) if (lens[b] < 0) { ret = lens[b]; break; }
    }
    for (int b = 0; b < nblocks; b++) free(spill[b]);
    free(spill);
    free(lens); pthread_mutex_destroy(&mu);
  }
  for (int b = 0; b < nblocks; b++) free(ins[b]);
  free(ins); free(bits);
  return ret;
}
