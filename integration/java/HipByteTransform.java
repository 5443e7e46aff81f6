package io.github.flanglet.kanzi.hip;

import io.github.flanglet.kanzi.ByteTransform;
import io.github.flanglet.kanzi.SliceByteArray;

/** ByteTransform backed by the HIP library: drop-in for BWTBlockCodec / SBRT / ZRLT in
 *  TransformFactory.newFunctionToken (K/transform/TransformFactory.java:273-351). One instance per
 *  task and block, like the reference codecs (not thread safe). */
public final class HipByteTransform implements ByteTransform {
  private final long ctx;
  private final int type;   // TransformFactory ids: BWT 1, ZRLT 6, MTFT 7, RANK 8

  public HipByteTransform(long ctx, int type) { this.ctx = ctx; this.type = type; }

  @Override public boolean forward(SliceByteArray src, SliceByteArray dst) { return run(true, src, dst); }
  @Override public boolean inverse(SliceByteArray src, SliceByteArray dst) { return run(false, src, dst); }
  @Override public int getMaxEncodedLength(int srcLen) { return KanziHip.maxEncodedLength(this.type, srcLen); }

  private boolean run(boolean fwd, SliceByteArray src, SliceByteArray dst) {
    if (src.length == 0) return true;
    if (src.array == dst.array) return false;                       // every reference codec refuses aliasing
    final int cap = (fwd ? dst.length : dst.array.length) - dst.index;
    final int r = KanziHip.transform(this.ctx, this.type, fwd, src.array, src.index, src.length, dst.array, dst.index, cap);
    if (r == KanziHip.DECLINED) return false;                       // Sequence sets the skip flag (Sequence.java:95-105)
    if (r < 0) throw new IllegalStateException("kanzi-hip error " + (-r));
    src.index += src.length;
    dst.index += r;
    return true;
  }
}
