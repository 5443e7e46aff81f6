package io.github.flanglet.kanzi.hip;

import io.github.flanglet.kanzi.InputBitStream;
import io.github.flanglet.kanzi.OutputBitStream;
import io.github.flanglet.kanzi.SliceByteArray;

import java.nio.ByteBuffer;
import java.util.Map;

/** The batched hooks of integration/kanzi-hip.patch.
 *  Writer: CompressedOutputStream.processBlock hands its filled buffers (one per job) to {@link #encode} instead of building one
 *  EncodingTask per buffer.
 *  Reader: CompressedInputStream.processBlock calls {@link #decode} instead of building one DecodingTask per buffer
 *  (K/io/CompressedInputStream.java:689-790): the serial walk of the 5 + lr bit length prefixes on the shared stream
 *  (DecodingTask.decodeBlock :1127-1129) and the copy of each block's bits (:1177-1187) happen here, the span ed.decode ...
 *  transform.inverse of every block of the batch (:1286-1344, block header and checksum verification included) is ONE native
 *  call, so the serial-per-block inverse stages (RANK / SRT / FPAQ) of the whole batch run side by side on the GPU instead of
 *  one block per call.  One native call codes the whole batch
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

  /** Decodes up to nbBlocks blocks from the shared bit stream into buffers[0..count) (block i at buffers[i].array[0..), index 0).
   *  @param bufSize   the reader's buffer size: max(blockSize + EXTRA_BUFFER_SIZE, blockSize + blockSize / 16)
   *  @param checksum  0, 32 or 64: the stream header's block checksum kind (the reader keeps it in its hashers, not in the map)
   *  @param count     count[0] = blocks decoded by this call (0 at the end marker)
   *  @return total decoded bytes, or -(K/Error.java code) of the first failing block in stream order */
  public static long decode(Map<String, Object> ctx, long transformType, int entropyType, SliceByteArray[] buffers,
      int nbBlocks, int blockSize, int bufSize, InputBitStream ibs, int checksum, int[] count) {
    count[0] = 0;
    final long h = HipRuntime.context(ctx);
    KanziHip.ctxSetChecksum(h, checksum);
    final long inStride = KanziHip.maxBlockStreamBytes(blockSize) + 64;
    final long outStride = bufSize;
    final ByteBuffer in = direct(0, inStride * nbBlocks);
    final long[] bitLengths = new long[nbBlocks];
    byte[] chunk = new byte[0];
    int n = 0;

    // serial walk of the length prefixes (DecodingTask.decodeBlock :1127-1129); a zero length is the end marker (:1131-1134)
    while (n < nbBlocks) {
      final int lr = (int) ibs.readBits(5) + 3;
      final long read = ibs.readBits(lr);

      if (read == 0)
        break;

      final long nbytes = (read + 7) >> 3;

      if (nbytes > inStride - 64)
        return -2;                                                       // Error.ERR_BLOCK_SIZE: no block of this stream is that long

      if (chunk.length < nbytes)
        chunk = new byte[(int) nbytes];

      long remaining = read;

      for (int off = 0; remaining > 0;) {                               // readBits(byte[], ...) in chunks of 2^30 bits (:1181-1186)
        final int c = (int) Math.min(remaining, 1L << 30);
        ibs.readBits(chunk, off, c);
        off += (c + 7) >> 3;
        remaining -= c;
      }

      in.position((int) (n * inStride));
      in.put(chunk, 0, (int) nbytes);
      bitLengths[n++] = read;
    }

    if (n == 0)
      return 0;

    final ByteBuffer out = direct(1, outStride * n);
    final int[] decodedLen = new int[n];
    final byte[] skipFlags = new byte[n];
    final int rc = KanziHip.decodeBlocks(h, transformType, entropyType, blockSize, in, inStride, bitLengths, n, out, outStride,
        decodedLen, skipFlags);

    if (rc != 0)
      return rc;

    long decoded = 0;

    for (int i = 0; i < n; i++) {
      if (decodedLen[i] < 0)
        return decodedLen[i];                                            // the first failing block in stream order, like the reader

      if (decodedLen[i] > blockSize)
        return -13;                                                      // "incorrectly decompressed" (:756-759)

      if (buffers[i].array.length < bufSize) {
        buffers[i].array = new byte[bufSize];
        buffers[i].length = bufSize;
      }

      out.position((int) (i * outStride));
      out.get(buffers[i].array, 0, decodedLen[i]);
      buffers[i].index = 0;
      decoded += decodedLen[i];
    }

    count[0] = n;
    return decoded;
  }
}
