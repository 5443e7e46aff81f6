package io.github.flanglet.kanzi.hip;

import io.github.flanglet.kanzi.EntropyEncoder;
import io.github.flanglet.kanzi.OutputBitStream;

/** EntropyEncoder backed by the HIP library: drop-in for ANSRangeEncoder(order 0) / NullEntropyEncoder in
 *  EntropyCodecFactory.newEncoder.  The native codec returns a bit string; it is appended to the block's
 *  private stream with writeBits(byte[],int,int) in chunks of <= 2^30 bits. */
public final class HipEntropyEncoder implements EntropyEncoder {
  private final long ctx;
  private final int type;   // EntropyCodecFactory ids: NONE 0, ANS0 5
  private final OutputBitStream bitstream;
  private byte[] buf = new byte[0];

  public static boolean supports(int type) { return (type == 0) || (type == 1) || (type == 2) || (type == 5); }   // NONE, HUFFMAN, FPAQ, ANS0
  public HipEntropyEncoder(long ctx, int type, OutputBitStream bs) { this.ctx = ctx; this.type = type; this.bitstream = bs; }

  @Override public int encode(byte[] block, int blkptr, int count) {
    if ((block == null) || (blkptr + count > block.length) || (blkptr < 0) || (count < 0)) return -1;
    if (count == 0) return 0;
    final int cap = count + (count >> 3) + 1024;
    if (this.buf.length < cap) this.buf = new byte[cap];
    long bits = KanziHip.entropyEncode(this.ctx, this.type, block, blkptr, count, this.buf);
    if (bits < 0) return -1;
    for (int n = 0; bits > 0; ) {
      final int chunk = (int) Math.min(bits, 1L << 30);
      this.bitstream.writeBits(this.buf, n, chunk);
      n += (chunk + 7) >> 3;
      bits -= chunk;
    }
    return count;
  }
  @Override public OutputBitStream getBitStream() { return this.bitstream; }
  @Override public void dispose() {}
}
