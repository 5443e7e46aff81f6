"""The block writer's raw fallback at its threshold (CompressedOutputStream.java:926-973: the entropy-coded block against the
post-transform bytes): nearly incompressible blocks (k equiprobable symbols for k = 240 .. 256, a few biased positions) whose coded size
lands within bytes of their length, through kz_encode_blocks against the oracle's encode_block (bits, skip flags, length, bytes), and back.
   python tools/rawfallback_fuzz.py [seconds] [seed]"""
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
CH = [("NONE", "ANS0"), ("NONE", "HUFFMAN"), ("NONE", "FPAQ"), ("ZRLT", "ANS0"), ("RANK+ZRLT", "HUFFMAN"), ("BWT+RANK+ZRLT", "ANS0"), ("LZ", "HUFFMAN")]
t0 = time.time(); cases = bad = raw = 0
while time.time() - t0 < budget:
    chain, ent = CH[int(rng.integers(0, len(CH)))]
    bs = int(rng.choice([256, 1024, 4096, 16384, 32768]))
    B = int(rng.integers(4, 17))
    lens = np.array([bs if rng.random() < 0.6 else int(rng.integers(16, bs + 1)) for _ in range(B)], np.int32)
    inp = np.zeros((B, bs), np.uint8)
    for b in range(B):
        n = int(lens[b]); k = int(rng.integers(236, 257))
        x = rng.integers(0, k, n, dtype=np.uint8)
        if rng.random() < 0.5:
            m = int(rng.integers(1, max(2, n // 16)))
            x[rng.integers(0, n, m)] = rng.integers(0, 4)
        inp[b, :n] = x
    ostride = kz.max_block_stream_bytes(bs)
    out = np.zeros((B, ostride), np.uint8)
    res = kz.encode_blocks(ctx, chain, ent, inp, bs, lens, out, ostride)
    bits = np.array([r.bits for r in res], np.int64)
    dec = np.zeros((B, bs), np.uint8)
    r2 = kz.decode_blocks(ctx, chain, ent, bs, out, ostride, bits, dec, bs)
    for b in range(B):
        src = inp[b, :lens[b]].tobytes()
        so, w, sf, pl = oracle.encode_block(chain, ent, src, block_size=bs)
        ok = res[b].status == 0 and (res[b].bits, res[b].skipFlags, res[b].length) == (w, sf, pl) and out[b, :(w + 7) // 8].tobytes() == so
        ok = ok and r2[b].status == 0 and dec[b, :lens[b]].tobytes() == src
        raw += (so[0] & 0x10) != 0 if len(so) else 0
        cases += 1
        if not ok:
            bad += 1
            print("MISMATCH", chain, ent, "bs", bs, "n", int(lens[b]), "oracle bits", w, "hip", res[b].bits, res[b].status, flush=True)
    if bad > 20: break
print("%d blocks, %d mismatches in %.0f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
