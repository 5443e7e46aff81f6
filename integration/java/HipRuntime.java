package io.github.flanglet.kanzi.hip;

import java.util.Map;

/** Glue for the patched factories (integration/kanzi-hip.patch): whether a context map asks for the HIP back end
 *  ("hip" = Boolean.TRUE, e.g. set by the application next to "transform" / "entropy") and the native context of the calling
 *  pool thread.  The reference creates codec instances per task and block on pool threads
 *  (K/io/CompressedOutputStream.java:541-566, :792, :907); a native context is single-threaded (one HIP stream, one arena),
 *  hence one per thread, on the device given by the map's "hipDevice" (default 0). */
public final class HipRuntime {
  private static final ThreadLocal<long[]> CTX = ThreadLocal.withInitial(() -> new long[] {0L, -1L});

  private HipRuntime() {}

  public static boolean enabled(Map<String, Object> ctx) {
    return (ctx != null) && Boolean.TRUE.equals(ctx.get("hip"));
  }

  /** the calling thread's native context, configured from the map's "checksum", "skipBlocks", "blockSize" and "entropy" */
  public static long context(Map<String, Object> ctx) {
    final int device = (ctx == null) ? 0 : (Integer) ctx.getOrDefault("hipDevice", 0);
    final long[] slot = CTX.get();

    if ((slot[0] == 0L) || (slot[1] != device)) {
      if (slot[0] != 0L)
        KanziHip.ctxDestroy(slot[0]);

      slot[0] = KanziHip.ctxCreate(device);
      slot[1] = device;

      if (slot[0] == 0L)
        throw new IllegalStateException("kanzi-hip: no usable HIP device " + device);
    }

    if (ctx != null) {
      KanziHip.ctxSetChecksum(slot[0], (Integer) ctx.getOrDefault("checksum", 0));
      KanziHip.ctxSetSkipBlocks(slot[0], (Boolean) ctx.getOrDefault("skipBlocks", false));
      KanziHip.ctxSetBlockSize(slot[0], (Integer) ctx.getOrDefault("blockSize", 4 * 1024 * 1024));
      final Object e = ctx.get("entropy");

      if (e instanceof String)
        KanziHip.ctxSetEntropy(slot[0], entropyId((String) e));
    }

    return slot[0];
  }

  /** EntropyCodecFactory ids of the codecs the library has; anything else -> -1 */
  public static int entropyId(String name) {
    switch (name.toUpperCase()) {
      case "NONE": return 0;
      case "HUFFMAN": return 1;
      case "FPAQ": return 2;
      case "ANS0": return 5;
      default: return 9;   // TPAQX and the rest only matter to TEXT as "not NONE / ANS0 / HUFFMAN / RANGE"
    }
  }
}
