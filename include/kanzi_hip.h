/*
 * kanzi_hip.h -- C-ABI of libkanzi_hip.so: the MI355X (gfx950) implementation of Kanzi's per-block
 * hot path.  Plain pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * What each entry point replaces in the reference (K/ = java/src/main/java/io/github/flanglet/kanzi/):
 *
 *   kz_transform_forward / _inverse / _max_encoded_len
 *       K/ByteTransform.java:36,48,56  (boolean forward(SliceByteArray,SliceByteArray), inverse, getMaxEncodedLength)
 *       for the codecs K/transform/BWTBlockCodec.java:71-213, K/transform/SBRT.java:87-214 (RANK, MTFT),
 *       K/transform/ZRLT.java:54-233, K/transform/SRT.java:66-257, K/transform/LZCodec.java:299-756 (LZ, LZX),
 *       K/transform/FSDCodec.java:60-323 (MM), K/transform/AliasCodec.java:76-475 (PACK, DNA), and, as host (CPU) stages in front of
 *       the GPU chain, K/transform/TextCodec.java:482-531 (TEXT) and K/transform/UTFCodec.java:68-305 (UTF).
 *       "false" is a normal outcome (Sequence.java:95-105) -> return 0.
 *   kz_entropy_encode / kz_entropy_decode
 *       K/EntropyEncoder.java:34 (int encode(byte[],int,int)) + dispose(), K/EntropyDecoder.java:33
 *       for K/entropy/ANSRangeEncoder.java:263-305, K/entropy/ANSRangeDecoder.java:189-236,
 *       K/entropy/HuffmanEncoder.java:380-416, K/entropy/HuffmanDecoder.java:353-390,
 *       K/entropy/FPAQEncoder.java:128-238, K/entropy/FPAQDecoder.java:161-335,
 *       K/entropy/NullEntropyEncoder.java:66-81.  The codec's output is a bit string (MSB first,
 *       K/bitstream/DefaultOutputBitStream.java:103-123): the Java adapter calls
 *       obs.writeBits(out, 0, nbits) once.
 *   kz_encode_blocks / kz_decode_blocks  (batched, the form in which the GPU pays off)
 *       the span transform.forward ... ee.dispose of EncodingTask.encodeBlock
 *       (K/io/CompressedOutputStream.java:792-985) resp. ed.decode ... transform.inverse of
 *       DecodingTask.decodeBlock (K/io/CompressedInputStream.java:1286-1344), including the block
 *       header (mode / skipFlags / postTransformLength / header checksum, :861-896,:977-985) and the
 *       raw "transformed copy" fallback (:926-973).  Output per block = the block's private
 *       byte-aligned stream of `bits` bits, ready for the ordered emission loop (:1024-1035).
 *
 * Return convention: >= 0 success (meaning documented per call); negative = -(K/Error.java code).
 * A kz_ctx is single-threaded (one HIP stream, one scratch arena); use one per host thread / GPU,
 * exactly like the reference creates one codec instance per task (CompressedOutputStream.java:792,907).
 */
#ifndef KANZI_HIP_H
#define KANZI_HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KZ_ABI_VERSION 3

/* transform ids: K/transform/TransformFactory.java:36-60 */
enum { KZ_T_NONE = 0, KZ_T_BWT = 1, KZ_T_LZ = 3, KZ_T_ZRLT = 6, KZ_T_MTFT = 7, KZ_T_RANK = 8, KZ_T_TEXT = 10 /* DICT_TYPE */,
       KZ_T_SRT = 13, KZ_T_MM = 15, KZ_T_LZX = 16, KZ_T_UTF = 17, KZ_T_PACK = 18, KZ_T_DNA = 19 };
/* Global.DataType (K/Global.java:40-80): the per-block context entry "dataType" that MM (FSDCodec.java:78-85,160-168)
   and LZ/LZX (LZCodec.java:343-352) read and write.  The KZ_DT_* VALUES ARE THIS LIBRARY'S OWN and are NOT the Java enum's
   ordinals (Java order: UNDEFINED, TEXT, MULTIMEDIA, EXE, NUMERIC, BASE64, DNA, BIN, UTF8, SMALL_ALPHABET); the value never
   reaches the stream, and a binding maps by NAME (integration/java/HipByteTransform.java). */
enum { KZ_DT_UNDEFINED = 0, KZ_DT_DNA = 1, KZ_DT_SMALL_ALPHABET = 2, KZ_DT_TEXT = 3, KZ_DT_MULTIMEDIA = 4, KZ_DT_EXE = 5,
       KZ_DT_NUMERIC = 6, KZ_DT_BASE64 = 7, KZ_DT_BIN = 8, KZ_DT_UTF8 = 9 };
/* entropy ids: K/entropy/EntropyCodecFactory.java */
enum { KZ_E_NONE = 0, KZ_E_HUFFMAN = 1, KZ_E_FPAQ = 2, KZ_E_ANS0 = 5 };
/* error codes: K/Error.java:24-43 (returned negated); KZ_ERR_DEVICE is the one code the reference has no equivalent for */
enum { KZ_ERR_MISSING_PARAM = 1, KZ_ERR_BLOCK_SIZE = 2, KZ_ERR_INVALID_CODEC = 3, KZ_ERR_READ_FILE = 11,
       KZ_ERR_WRITE_FILE = 12, KZ_ERR_PROCESS_BLOCK = 13, KZ_ERR_INVALID_FILE = 15, KZ_ERR_STREAM_VERSION = 16,
       KZ_ERR_INVALID_PARAM = 18, KZ_ERR_CRC_CHECK = 19, KZ_ERR_DEVICE = 126, KZ_ERR_UNKNOWN = 127 };

/* pointer location flags for the batched calls */
enum { KZ_MEM_HOST = 0, KZ_MEM_DEVICE = 1 };

#define KZ_MAX_STAGES 16
enum { KZ_STAGE_BWT_FWD = 0, KZ_STAGE_SBRT_FWD = 1, KZ_STAGE_ZRLT_FWD = 2, KZ_STAGE_ENTROPY_ENC = 3,
       KZ_STAGE_FRAME_ENC = 4, KZ_STAGE_ENTROPY_DEC = 5, KZ_STAGE_ZRLT_INV = 6, KZ_STAGE_SBRT_INV = 7,
       KZ_STAGE_BWT_INV = 8, KZ_STAGE_FRAME_DEC = 9, KZ_STAGE_LZ_FWD = 10, KZ_STAGE_LZ_INV = 11,
       KZ_STAGE_SRT_FWD = 12, KZ_STAGE_SRT_INV = 13, KZ_STAGE_HOST_FWD = 14, KZ_STAGE_HOST_INV = 15 };

typedef struct kz_ctx kz_ctx;

int32_t     kz_abi_version(void);
kz_ctx*     kz_ctx_create(int32_t deviceId);          /* NULL if no HIP device / out of memory */
void        kz_ctx_destroy(kz_ctx* ctx);
const char* kz_last_error(kz_ctx* ctx);
/* per-block checksum of the original data: bits = 0 (none), 32 (XXHash32) or 64 (XXHash64): the "checksum" key of
 * the reference's context map (K/io/CompressedOutputStream.java:190-204, :749-755, :887-891). Applies to the
 * following kz_encode_blocks / kz_decode_blocks / kz_compress calls (kz_decompress reads it from the stream). */
int32_t     kz_ctx_set_checksum(kz_ctx* ctx, int32_t bits);
/* the "dataType" key of the context map a transform instance is built with (KZ_DT_*): kz_transform_forward reads it
 * the way FSDCodec.forward / LZCodec.forward do (FSDCodec.java:78-85, LZCodec.java:343-352) and stores back what the
 * reference would store (FSDCodec.java:160-168).  The batched calls tag every block themselves, like the writer does
 * from the block's first four bytes (K/Magic.java, K/io/CompressedOutputStream.java:795-804). */
/* the "skipBlocks" key of the context map (CLI --skip): kz_encode_blocks / kz_compress store a block as a copy block
 * when its first bytes carry the magic number of a compressed format or its order-0 entropy is >= 0.95 * 8 bits
 * (K/io/CompressedOutputStream.java:769-788, Magic.isCompressed, Global.computeFirstOrderEntropy1024). Default off. */
int32_t     kz_ctx_set_skip_blocks(kz_ctx* ctx, int32_t on);
/* the "blockSize" and "entropy" keys of the context map as the TEXT transform reads them: TextCodec sizes its hash map by the
 * stream's block size (K/transform/TextCodec.java:561-575,1068-1081) and TransformFactory picks TextCodec1 or TextCodec2 by the
 * stream's entropy codec (TransformFactory.java:275-286).  kz_encode_blocks / kz_decode_blocks take the entropy codec from
 * their own argument; kz_encode_blocks takes the block size from here AS OF THE CALL (kz_submit_encode_blocks: as of the
 * submit) and refuses a chain with TEXT (-KZ_ERR_MISSING_PARAM) when it was never set, because the decoder is told the block
 * size explicitly and a silent default would write blocks another block size cannot read; kz_decode_blocks takes it from its
 * blockSize argument; kz_transform_forward / _inverse take both from here; kz_compress / kz_decompress set them from the
 * stream.  kz_ctx_set_entropy refuses TPAQX (9): TEXT's extra hash bit under it (TextCodec.java:561-575) is not modelled. */
/* The library's environment switches (KZ_*: diagnostics, A/B runs, the tests' forced schedules -- DESIGN.md 7; none is needed for
 * normal use) are read ONCE, by kz_ctx_create, into the context; no call reads the environment afterwards.  A test or A/B tool
 * that changes the environment of a live context calls this to have it read again.  (No reference counterpart: the reference
 * takes its options through the context map, K/io/CompressedOutputStream.java:140-190.) */
void        kz_ctx_reload_switches(kz_ctx* ctx);
int32_t     kz_ctx_set_block_size(kz_ctx* ctx, int32_t blockSize);
int32_t     kz_ctx_set_entropy(kz_ctx* ctx, uint32_t entropyType);
int32_t     kz_ctx_set_data_type(kz_ctx* ctx, int32_t dataType);
int32_t     kz_ctx_get_data_type(kz_ctx* ctx);
/* every kz_ctx_set_* value back to a fresh context's (checksum none, skipBlocks off, dataType UNDEFINED, block size unset, entropy
 * NONE): the values are sticky like the entries of the reference's context map, and a caller leaving through an error path may not
 * know which ones it had set */
int32_t     kz_ctx_reset(kz_ctx* ctx);
/* One process per GPU, N on one host: pin this process (and the threads it creates later: TEXT / UTF stages, bit assembly,
 * staging copies) to the CPUs of the GPU's NUMA node (/sys/bus/pci/devices/<bdf>/local_cpulist).  Returns the number of CPUs
 * in the new mask, 0 if the topology is not visible (nothing changed), <0 on error.  PROCESS-WIDE: every thread of the process
 * that exists at the time of the call is re-pinned (HIP's helpers, the application's own threads, the workers of contexts bound
 * to other GPUs), so it assumes ONE GPU PER PROCESS; a process that drives several devices must not call it (the last call would
 * move everything to one NUMA node).  The reference has no equivalent: its task pool is one JVM on one socket
 * (K/app/BlockCompressor.java:199-206). */
int32_t     kz_pin_to_device_numa(int32_t deviceId);
/* host CPUs the library's thread pool (TEXT / UTF stages, staging copies, bit assembly) will use: the process's affinity mask cut
 * down to its cgroup CPU quota, if any */
int32_t     kz_host_cpus(void);
/* N processes (one per GPU) under ONE cgroup quota: each takes 1/N of it (rounded up), so that together they stay inside -- a group
 * that runs more busy threads than its quota is frozen as a whole for the rest of every period, GPU-driving threads included.
 * ranksOnHost <= 1 restores the whole quota.  Returns the new kz_host_cpus(). */
int32_t     kz_host_share(int32_t ranksOnHost);
/* HIP stream the context launches on (as void* = hipStream_t) so callers can bracket it with events */
void*       kz_ctx_stream(kz_ctx* ctx);

/* ---- ByteTransform mirror (host buffers; one block) ------------------------------------------ */
/* returns 1 = applied (*produced bytes written), 0 = declined (dst untouched), <0 = error */
int32_t kz_transform_forward(kz_ctx* ctx, uint32_t type, const uint8_t* src, int32_t n,
                             uint8_t* dst, int32_t dstCap, int32_t* produced);
int32_t kz_transform_inverse(kz_ctx* ctx, uint32_t type, const uint8_t* src, int32_t n,
                             uint8_t* dst, int32_t dstCap, int32_t* produced);
int32_t kz_transform_max_encoded_len(uint32_t type, int32_t n);

/* TEXT and UTF (KZ_T_TEXT, KZ_T_UTF) are host stages: the same ByteTransform contract without a context, no GPU touched.
 * entropyType / blockSize = the "entropy" / "blockSize" entries of the reference's context map (TEXT only), *dataType = its
 * "dataType" entry, read and updated (NULL = a transform built without a context). */
int32_t kz_host_stage_forward(uint32_t type, uint32_t entropyType, int32_t blockSize, int32_t* dataType,
                              const uint8_t* src, int32_t n, uint8_t* dst, int32_t dstCap, int32_t* produced);
int32_t kz_host_stage_inverse(uint32_t type, int32_t blockSize, const uint8_t* src, int32_t n,
                              uint8_t* dst, int32_t dstCap, int32_t* produced);

/* ---- EntropyEncoder / EntropyDecoder mirror (host buffers; one block) ------------------------- */
/* returns number of BITS written to out (MSB first, last byte zero padded), <0 = error */
int64_t kz_entropy_encode(kz_ctx* ctx, uint32_t type, const uint8_t* src, int32_t n,
                          uint8_t* out, int64_t outCapBytes);
/* decodes `count` bytes from the bit string in[0..inBits); returns count, or <0; *bitsConsumed set */
int32_t kz_entropy_decode(kz_ctx* ctx, uint32_t type, const uint8_t* in, int64_t inBits,
                          uint8_t* dst, int32_t count, int64_t* bitsConsumed);

/* ---- fused, batched block API ----------------------------------------------------------------- */
typedef struct {
  int64_t bits;                 /* enc: bit length W of the block's private stream */
  int32_t length;               /* enc: postTransformLength ; dec: decoded length */
  int32_t status;               /* 0 ok, else -(K/Error.java code) */
  uint8_t skipFlags;            /* Sequence.getSkipFlags() (K/transform/Sequence.java:243) */
  uint8_t mode;                 /* block mode byte (CompressedOutputStream.java:861-880) */
  uint8_t pad[6];
} kz_block_result;

/*
 * Encode nBlocks independent blocks.  Block b is in[b*inStride .. +lengths[b]); its private stream
 * is written at out[b*outStride ..] (outStride >= kz_max_block_stream_bytes(maxLength)).
 * `lengths` and `results` are host arrays.  memKind says where in/out live (KZ_MEM_HOST: pageable or
 * pinned host memory, copied over PCIe inside the call; KZ_MEM_DEVICE: HBM pointers, no copies).
 * transformType = 8 x 6-bit ids, first transform in the top slot (TransformFactory.java:29-31).
 */
int32_t kz_encode_blocks(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType,
                         const uint8_t* in, int64_t inStride, const int32_t* lengths, int32_t nBlocks,
                         uint8_t* out, int64_t outStride, kz_block_result* results, int32_t memKind);
/*
 * Decode nBlocks block streams.  Stream b is in[b*inStride ..] with bitLengths[b] bits (W, header
 * included).  Decoded bytes go to out[b*outStride ..] (capacity outStride each, >= blockSize).
 */
int32_t kz_decode_blocks(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType, int32_t blockSize,
                         const uint8_t* in, int64_t inStride, const int64_t* bitLengths, int32_t nBlocks,
                         uint8_t* out, int64_t outStride, kz_block_result* results, int32_t memKind);
int64_t kz_max_block_stream_bytes(int32_t blockLength);
/*
 * The same two calls queued on the context's worker thread (what SURVEY 8b calls kz_submit / kz_wait): they return a job id
 * (> 0) at once, kz_wait(job) blocks until the call has run and gives its return code, kz_poll(job) says whether it has.  Every
 * array passed (lengths / bitLengths / results, and host in / out buffers) stays the caller's and must stay valid until kz_wait.
 * Jobs of one context run one at a time in submission order; use two contexts to overlap batches (the copies and the encode of
 * batch k+1 under the decode of batch k).  Between a submit and its wait only kz_submit_*, kz_wait and kz_poll may be called on
 * that context.
 */
int64_t kz_submit_encode_blocks(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType,
                                const uint8_t* in, int64_t inStride, const int32_t* lengths, int32_t nBlocks,
                                uint8_t* out, int64_t outStride, kz_block_result* results, int32_t memKind);
int64_t kz_submit_decode_blocks(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType, int32_t blockSize,
                                const uint8_t* in, int64_t inStride, const int64_t* bitLengths, int32_t nBlocks,
                                uint8_t* out, int64_t outStride, kz_block_result* results, int32_t memKind);
/* kz_wait hands a job's return code out ONCE: a second kz_wait / kz_poll on the same id returns -KZ_ERR_INVALID_PARAM. */
int32_t kz_wait(kz_ctx* ctx, int64_t job);
int32_t kz_poll(kz_ctx* ctx, int64_t job);

/* ---- whole .knz stream on host memory (K/io/CompressedOutputStream / CompressedInputStream) --- */
/* returns compressed size in bytes or <0 */
int64_t kz_compress(kz_ctx* ctx, uint64_t transformType, uint32_t entropyType, int32_t blockSize,
                    const uint8_t* src, int64_t n, uint8_t* dst, int64_t dstCap);
int64_t kz_decompress(kz_ctx* ctx, const uint8_t* src, int64_t n, uint8_t* dst, int64_t dstCap);
/* a dstCap that kz_compress never exceeds for n input bytes cut into blockSize blocks (a transform may grow a block:
 * SRT puts up to 1024 bytes of frequencies in front, K/transform/SRT.java MAX_HEADER_SIZE, which is 28 % on 1 KiB
 * blocks of random bytes; the reference's writer simply keeps writing to its OutputStream) */
int64_t kz_compress_bound(int64_t n, int32_t blockSize);
uint64_t kz_transform_type(const int32_t* types, int32_t nb);
/* host-only container helpers (no GPU touched): assemble a .knz from per-block streams gathered in
 * block-id order (stream header :236-313, 5+lw bit length prefix + payload per block :1024-1035, end
 * marker :491-492); index = the serial walk of block prefixes a decoder must do first (:1127-1129). */
/* checksumBits = 0 / 32 / 64: the kz_ctx_set_checksum value the block streams were coded with (it goes into the stream
 * header, :244-250; the block streams already carry their hash bytes) */
int64_t kz_knz_assemble(uint64_t transformType, uint32_t entropyType, int32_t blockSize, int64_t inputSize,
                        int32_t checksumBits, const uint8_t* streams, int64_t stride, const int64_t* bits, int32_t nBlocks,
                        uint8_t* dst, int64_t dstCap);
/* the same, block by block: a gatherer appends block streams in block-id order as they arrive (rank 0 of a multi-GPU job: round
 * r brings blocks r*N .. r*N+N-1 from ranks 0..N-1), holding one round instead of the whole stream.  _open writes the stream header
 * into dst, _add the 5 + lw bit length prefix and the bits of the next block, _close the end marker and returns the size in bytes
 * (or -KZ_ERR_WRITE_FILE when dst was too small); the writer is freed by _close. */
typedef struct kz_knz_writer kz_knz_writer;
kz_knz_writer* kz_knz_writer_open(uint64_t transformType, uint32_t entropyType, int32_t blockSize, int64_t inputSize,
                                  int32_t checksumBits, uint8_t* dst, int64_t dstCap);
int32_t kz_knz_writer_add(kz_knz_writer* w, const uint8_t* stream, int64_t bits);
int64_t kz_knz_writer_close(kz_knz_writer* w);
int32_t kz_knz_index(const uint8_t* src, int64_t n, uint64_t* transformType, uint32_t* entropyType,
                     int32_t* blockSize, int64_t* inputSize, int32_t* checksumBits,
                     int64_t* blockBitOff, int64_t* blockBits, int32_t cap);

/* ---- instrumentation (bench.py / roofline) ----------------------------------------------------- */
/* blocks that went through a TEXT / UTF stage on a HOST thread since the last reset (process-wide counter; inverse = 0: forward,
 * 1: inverse).  The stage timers KZ_STAGE_HOST_FWD / _INV measure the TEXT / UTF stage's wall time INCLUDING its device forms
 * (kz_text_fwd_gpu.hip, kz_utf_fwd_gpu.hip, kz_text_gpu.hip): this counter says how much of the stage the host really did. */
int64_t kz_host_stage_blocks(int32_t inverse, int32_t reset);
void    kz_set_timing(kz_ctx* ctx, int32_t enable);     /* hipEvent-bracket every stage of the next calls */
int32_t kz_get_stage_count(kz_ctx* ctx);
float   kz_get_stage_ms(kz_ctx* ctx, int32_t stage);    /* accumulated since last kz_reset_timing */
int64_t kz_get_stage_alg_bytes(kz_ctx* ctx, int32_t stage);
void    kz_reset_timing(kz_ctx* ctx);
/* per-kernel: HIP events recorded on the context's stream around every kernel launch */
void        kz_set_kernel_timing(kz_ctx* ctx, int32_t enable);
int32_t     kz_get_kernel_count(void);
const char* kz_get_kernel_name(int32_t id);
double      kz_get_kernel_ms(kz_ctx* ctx, int32_t id);        /* summed launch durations since reset */
double      kz_get_kernel_max_ms(kz_ctx* ctx, int32_t id);    /* longest single launch since reset (launches on the decoder's side streams overlap) */
int64_t     kz_get_kernel_launches(kz_ctx* ctx, int32_t id);
void        kz_reset_kernel_timing(kz_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
