package io.github.flanglet.kanzi.hip;

/** Native entry points of libkanzi_hip_jni.so (integration/jni/kanzi_hip_jni.c). */
public final class KanziHip {
  static { System.loadLibrary("kanzi_hip_jni"); }
  public static final int DECLINED = -1000000;
  public static native long ctxCreate(int device);
  public static native void ctxDestroy(long ctx);
  public static native int ctxSetChecksum(long ctx, int bits);   // 0, 32 or 64
  public static native int ctxSetSkipBlocks(long ctx, boolean on);   // context map key "skipBlocks"
  public static native int ctxSetBlockSize(long ctx, int blockSize);   // context map key "blockSize" (TEXT sizes its hash map by it)
  public static native int ctxSetEntropy(long ctx, int entropyType);   // context map key "entropy" (TEXT: TextCodec1 / TextCodec2)
  public static native int ctxSetDataType(long ctx, int dataType);   // Global.DataType as numbered by KZ_DT_* (kanzi_hip.h)
  public static native int ctxGetDataType(long ctx);
  public static native int ctxReset(long ctx);   // every ctxSet* value back to a fresh context's
  public static native int maxEncodedLength(int type, int n);
  public static native int transform(long ctx, int type, boolean forward, byte[] src, int srcIdx, int n, byte[] dst, int dstIdx, int dstCap);
  public static native long entropyEncode(long ctx, int type, byte[] block, int blkptr, int n, byte[] out);
  public static native int entropyDecode(long ctx, int type, byte[] in, int inOff, long inBits, byte[] block, int blkptr, int count, long[] bitsUsed);
  public static native int encodeBlocks(long ctx, long transformType, int entropyType, java.nio.ByteBuffer in, long inStride,
      int[] lengths, int nBlocks, java.nio.ByteBuffer out, long outStride, long[] bitsOut, int[] postLenOut, byte[] skipFlagsOut);
  /** decodedLenOut[b] = decoded length or -(Error code); streams b = the W bits of block b, header included */
  public static native int decodeBlocks(long ctx, long transformType, int entropyType, int blockSize, java.nio.ByteBuffer in, long inStride,
      long[] bitLengths, int nBlocks, java.nio.ByteBuffer out, long outStride, int[] decodedLenOut, byte[] skipFlagsOut);
  /** kz_max_block_stream_bytes: n + n/8 + 1024 rounded up to 256 */
  public static long maxBlockStreamBytes(int n) { return (((long) n + (n >> 3) + 1024) + 255) & ~255L; }
  private KanziHip() {}
}
