"""The writer's "skipBlocks" decision at its threshold (CompressedOutputStream.java:769-788: order-0 entropy against 0.95 * 8 bits, from
an integer histogram estimate): blocks of k equiprobable symbols for k around 2^7.6 = 194, biased mixtures, several block sizes; the
stream must be the oracle's byte for byte.   python tools/skip_threshold_fuzz.py [seconds] [seed]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kanzi_amd as kz
import oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
ctx = kz.Context(0)
print("seed", seed, flush=True)
t0 = time.time(); cases = bad = 0; copies = 0
while time.time() - t0 < budget:
    bs = int(rng.choice([1024, 4096, 16384, 65536]))
    parts = []
    for _ in range(int(rng.integers(3, 10))):
        n = bs if rng.random() < 0.8 else int(rng.integers(16, bs))
        k = int(rng.integers(170, 225))
        x = rng.integers(0, k, n, dtype=np.uint8)
        if rng.random() < 0.4:                               # a biased tail: shifts the entropy by a few hundredths of a bit
            m = int(rng.integers(1, max(2, n // 8)))
            x[rng.integers(0, n, m)] = rng.integers(0, 8)
        parts.append(x.tobytes())
    data = b"".join(parts)
    chain, ent = [("BWT+RANK+ZRLT", "ANS0"), ("LZ", "HUFFMAN"), ("NONE", "ANS0")][int(rng.integers(0, 3))]
    ref = oracle.compress(chain, ent, bs, data, jobs=1, checksum=0, skip_blocks=True)
    cos = kz.CompressedOutputStream(ctx, chain, ent, bs, checksum=0, skipBlocks=True)
    cos.write(data); cos.close()
    cases += 1
    modes = [kz.extract_bits(ref, off, 8)[0] for off, nb in kz.knz_index(ref)["blocks"]]
    copies += sum(1 for m in modes if (m & 0x80) and not (m & 0x10))
    if cos.output != ref:
        bad += 1
        print("MISMATCH", chain, ent, "bs", bs, "len", len(data), flush=True)
        if bad > 10: break
print("%d streams (%d copy blocks), %d mismatches in %.0f s" % (cases, copies, bad, time.time() - t0))
sys.exit(1 if bad else 0)
