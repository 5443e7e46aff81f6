package io.github.flanglet.kanzi.hip;

import java.util.Map;

/** Glue for the patched factories (integration/kanzi-hip.patch): whether a context map asks for the HIP back end
 *  ("hip" = Boolean.TRUE, e.g. set by the application next to "transform" / "entropy") and the native context of the calling
 *  pool thread.  The reference creates codec instances per task and block on pool threads
 *  (K/io/CompressedOutputStream.java:541-566, :792, :907); a native context is single-threaded (one HIP stream, one arena),
 *  hence one per thread, on the device given by the map's "hipDevice" (default 0). */
public final class HipRuntime {
  private static final ThreadLocal<long[]> CTX = ThreadLocal.withInitial(() -> new long[] {0L, -1L});

  // The decoder's widest schedule keeps four HIP streams busy side by side and HIP multiplexes ALL streams of a process over
  // GPU_MAX_HW_QUEUES hardware queues (4 by default), fixed when the HIP runtime starts.  A JVM cannot change its own
  // environment (and the native library deliberately does not: setenv under a running JVM races with getenv): start the JVM
  // with GPU_MAX_HW_QUEUES=8 in its environment.  Without it nothing breaks: the library measures that its streams share
  // queues and takes the three-stream schedule (about 10 % slower on large decode batches).  Checked once, before the
  // native library is loaded by the first KanziHip call.
  static {
    final String q = System.getenv("GPU_MAX_HW_QUEUES");
    int queues = 4;

    try {
      if (q != null)
        queues = Integer.parseInt(q.trim());
    } catch (NumberFormatException e) {
      queues = 4;
    }

    if (queues < 8)
      System.err.println("kanzi-hip: GPU_MAX_HW_QUEUES=" + ((q == null) ? "(unset)" : q)
          + "; export GPU_MAX_HW_QUEUES=8 before starting the JVM for the four-stream decoder schedule"
          + " (falling back to three streams)");
  }

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

      if (e instanceof String) {
        final int rc = KanziHip.ctxSetEntropy(slot[0], entropyId((String) e));

        if (rc < 0)
          throw new IllegalArgumentException("kanzi-hip: entropy codec " + e + " is not supported with the HIP transforms (" + rc + ")");
      }
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
      case "RANGE": return 4;
      case "CM": return 6;
      case "TPAQ": return 7;
      case "ANS1": return 8;
      default: return 9;   // TPAQX: refused by kz_ctx_set_entropy (TEXT's extra hash bit under it is not modelled) -> the caller sees -3
    }
  }
}
