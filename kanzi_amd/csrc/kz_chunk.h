// kz_chunk.h -- per-chunk entropy encoder output descriptors shared by the ANS0 and Huffman stages:
// every 16 KiB chunk produces  [hdrBits bits in hdr] ++ [tailBits bits at scr+tailOff]; a scan over the
// chunk bit lengths and a funnel-shift kernel concatenate them at bit granularity (kz_ans.hip).
#pragma once
#include "kz_internal.h"
typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint8_t u8;
#define ANS_CHUNK 16384
#define ANS_HDR_BYTES 512            // per-chunk header bit buffer
#define ANS_SCRATCH (32 * 1024)      // per-chunk output buffer (rANS worst case 12 bit/sym * 16384 + 19)
struct AnsEnc {
  u8* hdr;          // [B][C][ANS_HDR_BYTES]
  u8* scr;          // [B][C][ANS_SCRATCH]
  u32* hdrBits;     // [B][C]
  u32* tailOff;     // [B][C] offset inside the chunk scratch where varint|states|payload start
  u32* tailBits;    // [B][C]
  u64* bitOff;      // [B][C] exclusive scan of chunk bit lengths
  int C;            // chunk stride per block
};

int kz_chunk_enc_alloc(kz_ctx* ctx, kz_batch& bt, AnsEnc& E, int* chunksOut);
int kz_chunk_enc_finish(kz_ctx* ctx, kz_batch& bt, AnsEnc& E, int chunks, uint8_t* out, int64_t outStride,
                        const int32_t* d_hdrBytes, int64_t* d_bits, int rawLimit);
