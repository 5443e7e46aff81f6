package io.github.flanglet.kanzi.hip;

import io.github.flanglet.kanzi.io.CompressedOutputStream;

import java.io.ByteArrayOutputStream;
import java.nio.charset.StandardCharsets;
import java.nio.file.Files;
import java.nio.file.Path;
import java.nio.file.Paths;
import java.util.Arrays;
import java.util.HashMap;
import java.util.List;
import java.util.Map;

/**
 * Pins the repo's fixtures to the REAL reference: for every line of tests/golden/manifest.tsv
 * (input, output, transform chain, entropy codec, block size, checksum bits) it runs the reference's own
 * CompressedOutputStream on the input with one job and writes the stream next to the committed fixture as
 * &lt;output&gt;.ref, printing whether the two are byte-identical.  tools/promote_goldens.py does the diff / promotion.
 *
 * <pre>
 *   javac -cp kanzi.jar -d out integration/java/GoldenDump.java
 *   java -cp kanzi.jar:out io.github.flanglet.kanzi.hip.GoldenDump tests/golden
 * </pre>
 * Not compiled in the build image (no JDK there); it only uses the reference's public API
 * (K/io/CompressedOutputStream.java:140-227: the context keys "transform", "entropy", "blockSize", "checksum",
 * "jobs", "fileSize").
 */
public final class GoldenDump {
  public static void main(String[] args) throws Exception {
    final Path dir = Paths.get(args.length > 0 ? args[0] : "tests/golden");
    final List<String> lines = Files.readAllLines(dir.resolve("manifest.tsv"), StandardCharsets.UTF_8);
    int same = 0, diff = 0;

    for (String line : lines) {
      if (line.isEmpty() || line.startsWith("#"))
        continue;

      final String[] f = line.split("\t");
      final byte[] input = Files.readAllBytes(dir.resolve(f[0]));
      final Map<String, Object> ctx = new HashMap<>();
      ctx.put("transform", f[2]);
      ctx.put("entropy", f[3]);
      ctx.put("blockSize", Integer.parseInt(f[4]));
      ctx.put("checksum", Integer.parseInt(f[5]));
      ctx.put("jobs", 1);
      ctx.put("fileSize", (long) input.length);
      final ByteArrayOutputStream bos = new ByteArrayOutputStream(input.length / 2 + 1024);

      try (CompressedOutputStream cos = new CompressedOutputStream(bos, ctx)) {
        cos.write(input, 0, input.length);
      }

      final byte[] ref = bos.toByteArray();
      Files.write(dir.resolve(f[1] + ".ref"), ref);
      final boolean eq = Arrays.equals(ref, Files.readAllBytes(dir.resolve(f[1])));
      System.out.println((eq ? "SAME  " : "DIFF  ") + f[1] + "  (" + f[2] + " & " + f[3] + ", " + ref.length + " bytes)");

      if (eq)
        same++;
      else
        diff++;
    }

    System.out.println(same + " fixtures identical to the reference, " + diff + " different");
    System.exit(diff == 0 ? 0 : 1);
  }
}
