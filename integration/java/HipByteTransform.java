package io.github.flanglet.kanzi.hip;

import io.github.flanglet.kanzi.ByteTransform;
import io.github.flanglet.kanzi.SliceByteArray;

/** ByteTransform backed by the HIP library: drop-in for BWTBlockCodec / SBRT / ZRLT in
 *  TransformFactory.newFunctionToken (K/transform/TransformFactory.java:273-351). One instance per
 *  task and block, like the reference codecs (not thread safe). */
public final class HipByteTransform implements ByteTransform {
  // Global.DataType in the order of KZ_DT_* (include/kanzi_hip.h)
  private static final io.github.flanglet.kanzi.Global.DataType[] DT = {
      io.github.flanglet.kanzi.Global.DataType.UNDEFINED, io.github.flanglet.kanzi.Global.DataType.DNA,
      io.github.flanglet.kanzi.Global.DataType.SMALL_ALPHABET, io.github.flanglet.kanzi.Global.DataType.TEXT,
      io.github.flanglet.kanzi.Global.DataType.MULTIMEDIA, io.github.flanglet.kanzi.Global.DataType.EXE,
      io.github.flanglet.kanzi.Global.DataType.NUMERIC, io.github.flanglet.kanzi.Global.DataType.BASE64,
      io.github.flanglet.kanzi.Global.DataType.BIN, io.github.flanglet.kanzi.Global.DataType.UTF8};
  private final long ctx;
  private final int type;   // TransformFactory ids: BWT 1, LZ 3, ZRLT 6, MTFT 7, RANK 8, TEXT (DICT) 10, SRT 13, MM 15, LZX 16, UTF 17, PACK 18, DNA 19
  private final java.util.Map<String, Object> map;   // the task's context map (may be null, like the reference codecs)

  /** what TransformFactory.newFunctionToken asks before it builds the Java codec (integration/kanzi-hip.patch) */
  public static boolean supports(int type) {
    switch (type) {
      case 1: case 3: case 6: case 7: case 8: case 10: case 13: case 15: case 16: case 17: case 18: case 19: return true;
      default: return false;
    }
  }
  public HipByteTransform(java.util.Map<String, Object> map, int type) { this(HipRuntime.context(map), type, map); }
  public HipByteTransform(long ctx, int type) { this(ctx, type, null); }
  public HipByteTransform(long ctx, int type, java.util.Map<String, Object> map) { this.ctx = ctx; this.type = type; this.map = map; }

  @Override public boolean forward(SliceByteArray src, SliceByteArray dst) { return run(true, src, dst); }
  @Override public boolean inverse(SliceByteArray src, SliceByteArray dst) { return run(false, src, dst); }
  @Override public int getMaxEncodedLength(int srcLen) { return KanziHip.maxEncodedLength(this.type, srcLen); }

  private boolean run(boolean fwd, SliceByteArray src, SliceByteArray dst) {
    if (src.length == 0) return true;
    if (src.array == dst.array) return false;                       // every reference codec refuses aliasing
    final int cap = (fwd ? dst.length : dst.array.length) - dst.index;
    // "dataType" travels through the context map (FSDCodec.java:78-85,160-168; LZCodec.java:343-352)
    if (fwd && this.map != null) {
      final Object dt = this.map.getOrDefault("dataType", io.github.flanglet.kanzi.Global.DataType.UNDEFINED);
      KanziHip.ctxSetDataType(this.ctx, java.util.Arrays.asList(DT).indexOf(dt));
    }
    final int r = KanziHip.transform(this.ctx, this.type, fwd, src.array, src.index, src.length, dst.array, dst.index, cap);
    if (fwd && this.map != null) {
      final int after = KanziHip.ctxGetDataType(this.ctx);
      if (after > 0 || this.map.containsKey("dataType")) this.map.put("dataType", DT[after]);
    }
    if (r == KanziHip.DECLINED) return false;                       // Sequence sets the skip flag (Sequence.java:95-105)
    if (r < 0) throw new IllegalStateException("kanzi-hip error " + (-r));
    src.index += src.length;
    dst.index += r;
    return true;
  }
}
