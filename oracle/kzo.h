/*
 * kzo.h -- CPU ORACLE for the Kanzi per-block hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This directory is a plain-C restatement of the reference algorithms (flanglet/kanzi 2.5.0,
 * bitstream format 7, Java).  It exists so that tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py can check / time the HIP path against an independent CPU
 * implementation.  NOTHING in the product path (kanzi_amd/, include/) may include, link,
 * import or execute anything in oracle/.
 *
 * PARITY UNPINNED: the reference is 100 % Java and this image has no JVM, so the reference
 * cannot be run here; its own tests contain no byte-exact golden outputs (round-trip only).
 * The oracle is therefore pinned only by (1) the javadoc known answer BWT("mississippi") =
 * "ipssmpissii"/primary 5 (K/transform/BWT.java:45-50), (2) hand-derived vectors worked from
 * the cited reference lines (tests/golden/), (3) encode->decode identity on the reference's
 * own test-input generators.  Every function cites the reference file:line it follows;
 * K/ = java/src/main/java/io/github/flanglet/kanzi/.
 */
#ifndef KZO_H
#define KZO_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ids (K/transform/TransformFactory.java:36-60, K/entropy/EntropyCodecFactory.java) ---- */
enum { KZO_T_NONE = 0, KZO_T_BWT = 1, KZO_T_BWTS = 2, KZO_T_LZ = 3, KZO_T_RLT = 5, KZO_T_ZRLT = 6,
       KZO_T_MTFT = 7, KZO_T_RANK = 8, KZO_T_EXE = 9, KZO_T_TEXT = 10, KZO_T_ROLZ = 11,
       KZO_T_ROLZX = 12, KZO_T_SRT = 13, KZO_T_LZP = 14, KZO_T_MM = 15, KZO_T_LZX = 16,
       KZO_T_UTF = 17, KZO_T_PACK = 18, KZO_T_DNA = 19 };
enum { KZO_E_NONE = 0, KZO_E_HUFFMAN = 1, KZO_E_FPAQ = 2, KZO_E_RANGE = 4, KZO_E_ANS0 = 5,
       KZO_E_CM = 6, KZO_E_TPAQ = 7, KZO_E_ANS1 = 8, KZO_E_TPAQX = 9 };

/* Global.DataType (K/Global.java:40-80): the per-block "dataType" context entry that transforms read and write */
enum { KZO_DT_UNDEFINED = 0, KZO_DT_DNA = 1, KZO_DT_SMALL_ALPHABET = 2, KZO_DT_TEXT = 3, KZO_DT_MULTIMEDIA = 4,
       KZO_DT_EXE = 5, KZO_DT_NUMERIC = 6, KZO_DT_BASE64 = 7, KZO_DT_BIN = 8, KZO_DT_UTF8 = 9 };
/* K/Magic.java constants referred to by name */
#define KZO_MAGIC_JPG 0xFFD8FFE0u
#define KZO_MAGIC_RIFF 0x52494646u
#define KZO_MAGIC_BZIP2 0x425A68
#define KZO_MAGIC_MP3_ID3 0x494433
#define KZO_MAGIC_BMP 0x424D
#define KZO_MAGIC_PBM 0x5034
#define KZO_MAGIC_PGM 0x5035
#define KZO_MAGIC_PPM 0x5036

/* ---- MSB-first bit streams (K/bitstream/DefaultOutputBitStream.java:80-205,
 *      K/bitstream/DefaultInputBitStream.java:81-192) ---- */
typedef struct { uint8_t* buf; size_t cap; uint64_t nbits; int owns; int overflow; } kzo_obs;
typedef struct { const uint8_t* buf; uint64_t nbits; uint64_t pos; int error; } kzo_ibs;

void     kzo_obs_init(kzo_obs* s, size_t capBytes);           /* growable, owned   */
void     kzo_obs_wrap(kzo_obs* s, uint8_t* buf, size_t cap);  /* fixed, caller's   */
void     kzo_obs_free(kzo_obs* s);
void     kzo_obs_write(kzo_obs* s, uint64_t v, int count);    /* low `count` bits of v, MSB first */
void     kzo_obs_write_bytes(kzo_obs* s, const uint8_t* p, uint64_t nbits);
void     kzo_ibs_init(kzo_ibs* s, const uint8_t* buf, uint64_t nbits);
uint64_t kzo_ibs_read(kzo_ibs* s, int count);
void     kzo_ibs_read_bytes(kzo_ibs* s, uint8_t* p, uint64_t nbits);

/* ---- EntropyUtils (K/entropy/EntropyUtils.java) ---- */
int  kzo_encode_alphabet(kzo_obs* s, const int* alphabet, int count);
int  kzo_decode_alphabet(kzo_ibs* s, int* alphabet);
int  kzo_normalize_freqs(int* freqs, int* alphabet, int totalFreq, int scale);
void kzo_write_varint(kzo_obs* s, uint32_t v);
uint32_t kzo_read_varint(kzo_ibs* s);

/* ---- entropy codecs: encode appends to obs; decode returns bytes decoded or <0 ---- */
int kzo_ans0_encode(kzo_obs* s, const uint8_t* block, int count);
int kzo_ans0_decode(kzo_ibs* s, uint8_t* block, int count);
int kzo_huffman_encode(kzo_obs* s, const uint8_t* block, int count);
int kzo_huffman_decode(kzo_ibs* s, uint8_t* block, int count);
int kzo_fpaq_encode(kzo_obs* s, const uint8_t* block, int count);   /* includes dispose() */
int kzo_fpaq_decode(kzo_ibs* s, uint8_t* block, int count);
int kzo_null_encode(kzo_obs* s, const uint8_t* block, int count);
int kzo_null_decode(kzo_ibs* s, uint8_t* block, int count);
int kzo_entropy_encode(int type, kzo_obs* s, const uint8_t* block, int count);
int kzo_entropy_decode(int type, kzo_ibs* s, uint8_t* block, int count);

/* ---- transforms: return 1 applied, 0 declined/failed; *produced = bytes written ---- */
int kzo_zrlt_forward(const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced);
int kzo_zrlt_inverse(const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced);
int kzo_sbrt_forward(int mode, const uint8_t* src, int n, uint8_t* dst);
int kzo_sbrt_inverse(int mode, const uint8_t* src, int n, uint8_t* dst);
int kzo_srt_forward(const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced);
int kzo_srt_inverse(const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced);
/* raw BWT: out n bytes + primary[8] (K/transform/BWT.java:148-191) */
int kzo_bwt_forward_raw(const uint8_t* src, int n, uint8_t* dst, int32_t primary[8]);
int kzo_bwt_inverse_raw(const uint8_t* src, int n, uint8_t* dst, const int32_t primary[8]);
/* BWTBlockCodec with header (K/transform/BWTBlockCodec.java:71-213) */
int kzo_bwt_forward(const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced);
int kzo_bwt_inverse(const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced);
int kzo_lz_forward(int lzx, int dataType, const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced);
int kzo_lz_inverse(int lzx, const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced);
void kzo_suffix_array(const uint8_t* t, int32_t* sa, int n);      /* SA-IS */

/* Global / Magic helpers (kzo_global.c) */
int kzo_log2_4096(int x);
int kzo_log2_1024(int x);
int kzo_entropy1024(int length, const int* histo);
int kzo_detect_simple_type(int count, const int* freqs0);
int32_t kzo_magic_type(const uint8_t* src);
int kzo_magic_is_compressed(int32_t magic);
int kzo_magic_is_multimedia(int32_t magic);
int kzo_magic_is_executable(int32_t magic);
int kzo_block_data_type(const uint8_t* data, int n);
/* FSDCodec = transform MM (kzo_fsd.c); dataType in/out, NULL = transform built without a context */
int kzo_fsd_max_encoded_len(int n);
int kzo_fsd_forward(int* dataType, const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced);
int kzo_fsd_inverse(const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced);

/* AliasCodec = transforms PACK and DNA (kzo_alias.c) */
int kzo_alias_max_encoded_len(int n);
int kzo_alias_forward(int onlyDNA, int* dataType, const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced);
int kzo_alias_inverse(const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced);

/* TextCodec = transform TEXT (kzo_text.c): codecType 1 / 2 = TextCodec1 / TextCodec2, blockSize = the context's "blockSize";
   UTFCodec = transform UTF (kzo_utf.c).  Both read and write the block's dataType. */
int kzo_text_forward(int codecType, int blockSize, int* dataType, const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced);
int kzo_text_inverse(int blockSize, const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced);
int kzo_text_static_dict_words(void);
int kzo_utf_forward(int* dataType, const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced);
int kzo_utf_inverse(const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced);
/* the context entries "entropy" and "blockSize" for the calling thread's next transform calls (TEXT reads them) */
void kzo_set_transform_ctx(int entropyType, int blockSize);

int kzo_transform_max_encoded_len(int type, int n);
/* dataType: the block's context entry, read and updated by the stages that use it; NULL = no context */
int kzo_transform_forward(int type, int* dataType, const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced);
int kzo_transform_inverse(int type, const uint8_t* src, int n, uint8_t* dst, int dstCap, int* produced);

/* ---- Sequence (K/transform/Sequence.java:56-207) ---- */
/* types[nb] in chain order. Returns post-transform length; *skipFlags per reference. */
int kzo_sequence_forward(const int* types, int nb, int* dataType, const uint8_t* src, int n,
                         uint8_t* dst, int dstCap, uint8_t* skipFlags);
int kzo_sequence_inverse(const int* types, int nb, uint8_t skipFlags, const uint8_t* src, int n,
                         uint8_t* dst, int dstCap);

/* ---- block + stream (.knz v7) (K/io/CompressedOutputStream.java, CompressedInputStream.java) ---- */
/* Encode one block into its private byte-aligned stream; returns bit length W, bytes in out. */
int64_t kzo_encode_block(uint64_t transformType, int entropyType, const uint8_t* data, int n,
                         uint8_t* out, size_t outCap, uint8_t* skipFlagsOut, int* postLenOut);
int     kzo_decode_block(uint64_t transformType, int entropyType, int blockSize, const uint8_t* in,
                         int64_t nbits, uint8_t* out, int outCap);
/* _x variants: chkKind 0 none / 1 XXHash32 / 2 XXHash64 block checksum (-x32 / -x64); encoders: | 0x100 = the writer's
   "skipBlocks" option (CompressedOutputStream.java:769-788) */
int64_t kzo_encode_block_x(uint64_t transformType, int entropyType, int chkKind, const uint8_t* data, int n,
                           uint8_t* out, size_t outCap, uint8_t* skipFlagsOut, int* postLenOut);
int     kzo_decode_block_x(uint64_t transformType, int entropyType, int chkKind, int blockSize, const uint8_t* in,
                           int64_t nbits, uint8_t* out, int outCap);
int64_t kzo_encode_block_y(uint64_t transformType, int entropyType, int chkKind, int blockSize, const uint8_t* data, int n,
                           uint8_t* out, size_t outCap, uint8_t* skipFlagsOut, int* postLenOut);
int64_t kzo_compress_x(uint64_t transformType, int entropyType, int blockSize, int chkKind, const uint8_t* src,
                       int64_t n, uint8_t* dst, int64_t dstCap, int jobs);
uint32_t kzo_xxhash32(const uint8_t* data, int length, uint32_t seed);
uint64_t kzo_xxhash64(const uint8_t* data, int length, uint64_t seed);
/* whole stream, `jobs` threads over blocks; returns bytes written or <0 */
int64_t kzo_compress(uint64_t transformType, int entropyType, int blockSize, const uint8_t* src,
                     int64_t n, uint8_t* dst, int64_t dstCap, int jobs);
int64_t kzo_decompress(const uint8_t* src, int64_t n, uint8_t* dst, int64_t dstCap, int jobs);
uint64_t kzo_transform_type(const int* types, int nb);
int     kzo_stream_header(uint64_t transformType, int entropyType, int blockSize, int chkKind,
                          int64_t inputSize, uint8_t* out /* >= 26 bytes */);

/* java.util.Random (for the reference's test-input generators) */
typedef struct { uint64_t seed; } kzo_jrandom;
void    kzo_jrandom_init(kzo_jrandom* r, int64_t seed);
int32_t kzo_jrandom_next_int(kzo_jrandom* r, int32_t bound);

/* scratch allocation of the oracle's sources: big buffers are parked per thread instead of going back to the kernel (kzo_alloc.c) */
void*   kzo_malloc(size_t n);
void*   kzo_calloc(size_t a, size_t b);
void*   kzo_realloc(void* p, size_t n);
void    kzo_free(void* p);
void    kzo_alloc_thread_cleanup(void);

#ifdef __cplusplus
}
#endif
#ifndef KZO_NO_ALLOC_MACROS
#include <stdlib.h>
#define malloc kzo_malloc
#define calloc kzo_calloc
#define realloc kzo_realloc
#define free kzo_free
#endif
#endif
