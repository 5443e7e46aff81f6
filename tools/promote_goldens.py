#!/usr/bin/env python3
"""Pinning the fixtures to the real reference (one command each side, as soon as a JVM exists somewhere):

  1. python tools/promote_goldens.py --tsv          writes tests/golden/manifest.tsv from manifest.json (committed)
  2. on a host with a JDK and kanzi.jar:            javac -cp kanzi.jar -d out integration/java/GoldenDump.java
                                                    java -cp kanzi.jar:out io.github.flanglet.kanzi.hip.GoldenDump tests/golden
     -> tests/golden/<fixture>.knz.ref  (the reference's own output for every fixture input)
  3. python tools/promote_goldens.py                diffs every .ref against the committed fixture; with --promote the
                                                    .ref files replace the fixtures and manifest.json's provenance becomes
                                                    "reference-generated (kanzi <version>)".
A difference is a lead, not a verdict: the first differing byte and the block it falls in are printed."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tsv", action="store_true", help="(re)write manifest.tsv for integration/java/GoldenDump.java")
    ap.add_argument("--promote", action="store_true", help="replace the fixtures by the reference's outputs")
    ap.add_argument("--version", default="2.5.0", help="reference version the .ref files came from")
    args = ap.parse_args()
    man = json.load(open(os.path.join(GOLD, "manifest.json")))
    if args.tsv:
        with open(os.path.join(GOLD, "manifest.tsv"), "w") as f:
            f.write("# input\toutput\ttransform\tentropy\tblockSize\tchecksum   (for integration/java/GoldenDump.java)\n")
            for e in man["entries"]:
                f.write("%s\t%s\t%s\t%s\t%d\t%d\n" % (e["input"], e["output"], e["chain"], e["entropy"], e["blockSize"], e.get("checksum", 0)))
        print("wrote manifest.tsv (%d entries)" % len(man["entries"]))
        return 0
    missing = same = diff = 0
    for e in man["entries"]:
        ref = os.path.join(GOLD, e["output"] + ".ref")
        if not os.path.exists(ref):
            missing += 1
            continue
        a, b = open(os.path.join(GOLD, e["output"]), "rb").read(), open(ref, "rb").read()
        if a == b:
            same += 1
        else:
            diff += 1
            k = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
            print("DIFF %s: fixture %d bytes, reference %d bytes, first difference at byte %d (bit %d)" % (e["output"], len(a), len(b), k, 8 * k))
        if args.promote:
            os.replace(ref, os.path.join(GOLD, e["output"]))
    print("%d identical, %d different, %d without a .ref" % (same, diff, missing))
    if args.promote and missing == 0:
        man["provenance"] = "reference-generated (flanglet/kanzi %s, integration/java/GoldenDump.java, jobs=1)" % args.version
        json.dump(man, open(os.path.join(GOLD, "manifest.json"), "w"), indent=1)
        print("fixtures promoted")
    return 1 if diff else 0


if __name__ == "__main__":
    sys.exit(main())
