package io.github.flanglet.kanzi.hip;

import io.github.flanglet.kanzi.OutputBitStream;
import io.github.flanglet.kanzi.SliceByteArray;

import java.nio.ByteBuffer;
import java.util.Map;

/** The batched hook of integration/kanzi-hip.patch: CompressedOutputStream.processBlock hands its filled buffers (one per
 *  job) to {@link #encode} instead of building one EncodingTask per buffer.  One native call codes the whole batch
 *  (transforms, block header with its checksum, entropy coding, raw fallback: the span
 *  K/io/CompressedOutputStream.java:792-985); what remains here is the reference's ordered emission into the shared stream
 *  (:1024-1035): per block 5 bits of length-of-length, the bit length W, then the W bits. */
public final class HipBlockBatch {
  private HipBlockBatch() {}

  private static final ThreadLocal<ByteBuffer[]> BUFS = ThreadLocal.withInitial(() -> new ByteBuffer[2]);

  private static ByteBuffer direct(int slot, long cap) {
    final ByteBuffer[] b = BUFS.get();

    if ((b[slot] == null) || (b[slot].capacity() < cap))
      b[slot] = ByteBuffer.allocateDirect((int) Math.min(Integer.MAX_VALUE, cap + (cap >> 3)));

    b[slot].clear();
    return b[slot];
  }

  /** @return the K/Error.java code (0 = success) */
  public static int encode(Map<String, Object> ctx, long transformType, int entropyType, SliceByteArray[] buffers,
      int nbBuffers, int blockSize, OutputBitStream obs) {
    int n = 0;

    while ((n < nbBuffers) && (buffers[n].index > 0))
      n++;

    if (n == 0)
      return 0;

    final long h = HipRuntime.context(ctx);
    final long inStride = blockSize;
    final long outStride = KanziHip.maxBlockStreamBytes(blockSize);
    final ByteBuffer in = direct(0, inStride * n);
    final ByteBuffer out = direct(1, outStride * n);
    final int[] lengths = new int[n];

    for (int i = 0; i < n; i++) {
      lengths[i] = buffers[i].index;
      in.position((int) (i * inStride));
      in.put(buffers[i].array, 0, lengths[i]);
      buffers[i].index = 0;
    }

    final long[] bits = new long[n];
    final int[] postLen = new int[n];
    final byte[] skipFlags = new byte[n];
    final int rc = KanziHip.encodeBlocks(h, transformType, entropyType, in, inStride, lengths, n, out, outStride, bits, postLen, skipFlags);

    if (rc != 0)
      return -rc;

    byte[] chunk = new byte[0];

    for (int i = 0; i < n; i++) {
      if (postLen[i] < 0)
        return -postLen[i];

      final long written = bits[i];
      final int lw = (written < 8) ? 3 : (31 - Integer.numberOfLeadingZeros((int) (written >> 3))) + 4;   // Global.log2
      obs.writeBits(lw - 3, 5);
      obs.writeBits(written, lw);
      final int nbytes = (int) ((written + 7) >> 3);

      if (chunk.length < nbytes)
        chunk = new byte[nbytes];

      out.position((int) (i * outStride));
      out.get(chunk, 0, nbytes);
      long remaining = written;

      for (int off = 0; remaining > 0;) {                                   // writeBits(byte[], ...) in chunks of 2^30 bits
        final int c = (int) Math.min(remaining, 1L << 30);
        obs.writeBits(chunk, off, c);
        off += c >> 3;
        remaining -= c;
      }
    }

    return 0;
  }
}
