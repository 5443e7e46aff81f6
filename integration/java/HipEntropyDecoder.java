package io.github.flanglet.kanzi.hip;

import io.github.flanglet.kanzi.EntropyDecoder;
import io.github.flanglet.kanzi.InputBitStream;

/** EntropyDecoder backed by the HIP library: drop-in for ANSRangeDecoder(order 0) / HuffmanDecoder / FPAQDecoder /
 *  NullEntropyDecoder in EntropyCodecFactory.newDecoder.
 *
 *  The reference decoders pull bits from the block's private InputBitStream as they go
 *  (K/io/CompressedInputStream.java:1286-1330 builds that stream over the block's payload bytes).  The native codec
 *  needs the payload as one bit string, so the patched DecodingTask hands the payload array and its bit length to
 *  this adapter; after decoding, the bits the codec consumed (kz_entropy_decode's bitsConsumed) are skipped on the
 *  InputBitStream so that its position matches what the Java decoder would have left behind. */
public final class HipEntropyDecoder implements EntropyDecoder {
  private final long ctx;
  private final int type;        // EntropyCodecFactory ids: NONE 0, HUFFMAN 1, FPAQ 2, ANS0 5
  private final InputBitStream bitstream;
  private final byte[] payload;  // the block's entropy-coded payload, bit 0 = MSB of payload[0]
  private long bitPos;           // bits already consumed by earlier decode() calls
  private final long bitLen;
  private final long[] used = new long[1];

  public static boolean supports(int type) { return (type == 0) || (type == 1) || (type == 2) || (type == 5); }
  public HipEntropyDecoder(long ctx, int type, InputBitStream bs, byte[] payload, long bitLen) {
    this.ctx = ctx; this.type = type; this.bitstream = bs; this.payload = payload; this.bitLen = bitLen;
    this.bitPos = bs.read();      // the block header (and checksum) in front of the payload have been read from the same stream
  }
  /** built by the patched EntropyCodecFactory.newDecoder from what the patched DecodingTask left in the map:
   *  "hipPayload" = the block's bytes (data.array), "hipPayloadBits" = its bit length W */
  public HipEntropyDecoder(java.util.Map<String, Object> map, int type, InputBitStream bs) {
    this(HipRuntime.context(map), type, bs, (byte[]) map.get("hipPayload"), (Long) map.get("hipPayloadBits"));
  }

  @Override public int decode(byte[] block, int blkptr, int count) {
    if ((block == null) || (blkptr + count > block.length) || (blkptr < 0) || (count < 0)) return -1;
    if (count == 0) return 0;
    if ((this.bitPos & 7) != 0) return -1;    // the codecs in scope start byte aligned inside the private stream
    final int rc = KanziHip.entropyDecode(this.ctx, this.type, this.payload, (int) (this.bitPos >> 3),
        this.bitLen - this.bitPos, block, blkptr, count, this.used);
    if (rc < 0) return -1;
    for (long n = this.used[0]; n > 0; ) {    // keep the shared stream position in step
      final int chunk = (int) Math.min(n, 64);
      this.bitstream.readBits(chunk);
      n -= chunk;
    }
    this.bitPos += this.used[0];
    return count;
  }
  @Override public InputBitStream getBitStream() { return this.bitstream; }
  @Override public void dispose() {}
}
