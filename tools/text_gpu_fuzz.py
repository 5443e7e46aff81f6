"""The device forms of the TEXT inverse (and the device UTF inverse) against the oracle's decoder on damaged streams (needs a GPU).  Streams of the chain TEXT (and
TEXT+UTF) with entropy NONE carry the TEXT-coded bytes as they are, so flipped bits land in tokens, numbers and escapes; every copy
is decoded with KZ_TEXT_GPU = 1 / 2 / 3 (and the host stage, 0) and must give the oracle's verdict and bytes.
   python tools/text_gpu_fuzz.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import kanzi_amd as kz
import oracle, textgen

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
big4 = len(sys.argv) > 3 and sys.argv[3] == "big4"      # the first stream: three blocks of 4 MiB (BASELINE's block size), four damaged copies
rng = np.random.default_rng(seed)
print("seed", seed, flush=True)
ctx = kz.Context(0)
t0 = time.time(); cases = bad = refused = big = 0
while time.time() - t0 < budget:
    chain = ["TEXT", "TEXT+UTF", "UTF"][int(rng.choice(3, p=[0.5, 0.25, 0.25]))]
    ent = "NONE" if rng.random() < 0.8 else "FPAQ"
    bs = int(rng.choice([32768, 65536, 1 << 18, 1 << 20]))
    parts = []
    first_big = big4 and cases == 0
    if first_big:
        chain, ent, bs = "TEXT+UTF", "NONE", 4 << 20
        parts = [textgen.bulk_text(bs, 71, "english").tobytes(), textgen.vocab_words(bs, 72, 150000).tobytes(), textgen.bulk_text(bs - 4321, 73, "utf8").tobytes()]
        big += 3
    for _ in range(0 if first_big else int(rng.integers(1, 5))):
        n = int(rng.integers(2000, 3 * bs))
        k = int(rng.integers(0, 7)) if chain != "UTF" else int(rng.choice([0, 6, 6, 6])); s = int(rng.integers(0, 1 << 30))
        if k == 0: parts.append(textgen.english(n, s))
        elif k == 1: parts.append(textgen.english(n, s, crlf=True))
        elif k == 2: parts.append(textgen.xml(n, s))
        elif k == 3: parts.append(textgen.many_words(n, s, alphabet=int(rng.integers(4, 26))))
        elif k == 4: parts.append(textgen.english(n, s, sprinkle=bytes([0x0F, 0x0E, 0x80, 0xFF])))
        elif k == 5: parts.append(textgen.english(n, s, invented=int(rng.integers(10, 20000))))
        else: parts.append(textgen.utf8(n, s, bom=bool(s & 1)))
    data = b"".join(bytes(p) for p in parts)
    ref = oracle.compress(chain, ent, bs, data, jobs=4)
    copies = [ref]
    for _ in range(4 if first_big else 10):
        b = bytearray(ref)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(24, len(b) - 4))
            if rng.random() < 0.5: b[pos] ^= 1 << int(rng.integers(0, 8))
            else: b[pos] = int(rng.integers(0, 256))
        copies.append(bytes(b))
    for c in copies:
        try:
            want = oracle.decompress(c, len(data) + 4 * bs, jobs=2)
        except Exception:
            want = None
        refused += want is None
        for form in ("1", "3", "2", "0"):
            os.environ["KZ_TEXT_GPU"] = form
            ctx.reload_switches()
            try:
                got = kz.CompressedInputStream(ctx, c).read()
            except Exception:
                got = None
            cases += 1
            if (got is None) != (want is None) or (got is not None and got != want):
                bad += 1
                print("MISMATCH", chain, ent, "bs", bs, "n", len(data), "form", form, "oracle", None if want is None else len(want), "hip", None if got is None else len(got), flush=True)
    if bad > 20: break
print("4MiB cases: %d" % big)
print("%d decodes (%d of the streams refused by the oracle), %d mismatches in %.0f s" % (cases, refused, bad, time.time() - t0))
sys.exit(1 if bad else 0)
