"""Entropy decoders asked for the WRONG number of symbols, or given fewer bits than the encoder wrote: valid (and corrupted) streams of
n symbols decoded with count = n - 1, n + 1, n -+ one chunk, 1, 32, 33 ... and with the bit length cut or padded; the verdict
(decode() == count), the bytes and the bits consumed must be the oracle's.   python tools/entropy_count_fuzz.py [seconds] [seed]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kanzi_amd as kz
import oracle, datagen, refinputs

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
ctx = kz.Context(0)
DEC = {"ANS0": kz.ANSRangeDecoder, "HUFFMAN": kz.HuffmanDecoder, "FPAQ": kz.FPAQDecoder, "NONE": kz.NullEntropyDecoder}
print("seed", seed, flush=True)
t0 = time.time(); cases = bad = 0; per = {}
while time.time() - t0 < budget:
    ent = ["ANS0", "HUFFMAN", "FPAQ", "NONE"][int(rng.integers(0, 4))]
    base = int(rng.choice([16384, 32768, 49152, int(rng.integers(1, 400)), int(rng.integers(400, 70000))]))
    n = max(1, base + int(rng.integers(-3, 4)))
    data = datagen.block(int(rng.integers(0, 1 << 16)), n, int(rng.integers(0, 5))).tobytes()
    good, nbits = oracle.entropy_encode(ent, data)
    stream = good if (rng.random() < 0.7 or len(good) < 4) else bytes(refinputs.corrupt(rng, good, int(rng.integers(0, 8))))
    if len(stream) == 0: continue
    nbmax = min(nbits, len(stream) * 8)
    counts = sorted(set(max(1, c) for c in (n - 1, n, n + 1, n - 16384, n + 16384, n - 16385, 1, 31, 32, 33, n // 2, 2 * n)))
    for count in counts:
        for nb in sorted(set(max(0, b) for b in (nbmax, nbmax - 1, nbmax - 8, nbmax - 64, nbmax // 2))):
            if rng.random() < 0.5 and (count != n or nb != nbmax): continue
            r, o, used = oracle.entropy_decode(ent, stream, nb, count)
            buf = np.zeros(count, dtype=np.uint8)
            d = DEC[ent](ctx, stream, nb)
            ok_p = d.decode(buf, 0, count) == count
            exact = count == n and nb == nbits and stream is good          # the position behind the block is only defined for the call the encoder's output asks for
            ok = ok_p == (r == count) and (not ok_p or (bytes(buf) == o and (not exact or d.bits_consumed == used)))
            cases += 1; per[ent] = per.get(ent, 0) + 1
            if not ok:
                bad += 1
                if bad <= 6: np.savez(os.path.join(ROOT, "gpurun_out", "entcount_fail_%d_%d.npz" % (seed, bad)), stream=np.frombuffer(stream, np.uint8), good=np.frombuffer(good, np.uint8), n=n, count=count, nb=nb, nbits=nbits, ent=ent)
                print("MISMATCH", ent, "n", n, "count", count, "bits", nb, "of", nbits, "oracle", r, used, "hip", ok_p, getattr(d, "bits_consumed", None), flush=True)
                if bad > 40: break
        if bad > 40: break
    if bad > 40: break
print("%d cases %s, %d mismatches in %.0f s" % (cases, per, bad, time.time() - t0))
sys.exit(1 if bad else 0)
