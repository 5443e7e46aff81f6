"""Differential run of the ZRLT stage (forward through the block API, chain "ZRLT" & NONE, and back) against the oracle on inputs built
to stress the row / wave / tile seams of the forward kernels: zero runs of every length class (inside a row, across rows, waves, 4 KiB
tiles, the whole block), escapes (0xFE / 0xFF) dense and sparse, blocks that expand (the transform declines), ragged batches.
   python tools/zrlt_fuzz.py [seconds] [seed]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kanzi_amd as kz
import oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
ctx = kz.Context(0)
print("seed", seed, flush=True)
RUNS = [1, 2, 3, 4, 7, 8, 15, 16, 31, 62, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 8192, 8193, 65535, 65536, 300000]


def block():
    n = int(rng.choice([rng.integers(16, 300), rng.integers(300, 9000), rng.integers(9000, 200000), rng.integers(200000, 1 << 21),
                        rng.choice([63, 64, 65, 1023, 1024, 1025, 4095, 4096, 4097, 8192, 12288, 65536])]))
    kind = int(rng.integers(0, 8))
    if kind >= 6:                                   # output length within a few bytes of n: the forward's three bound checks decide
        x = rng.integers(1, 0xFE, n, dtype=np.uint8)                    # n literals -> n bytes
        e = int(rng.integers(0, 6))                                      # e escapes: + e
        z = int(rng.integers(0, 6))                                      # z runs of 2..3 zeros: - 1 each
        where = rng.permutation(max(1, n // 4))[:e + z] * 4              # disjoint slots of 4 bytes
        early = kind == 7                                                # escapes first, runs last: the offset runs ahead of the input
        slots = np.sort(where)
        if not early: slots = rng.permutation(slots)
        for j, p0 in enumerate(slots):
            if p0 + 3 >= n: continue
            if j < e: x[p0] = rng.choice([0xFE, 0xFF])
            else: x[p0:p0 + int(rng.integers(2, 4))] = 0
        return np.ascontiguousarray(x)
    if kind == 0:                                   # runs of chosen lengths separated by a few literals
        out, left = [], n
        esc = rng.random() < 0.5
        while left > 0:
            L = min(int(rng.choice(RUNS)), left)
            out.append(np.zeros(L, np.uint8)); left -= L
            m = min(int(rng.integers(1, 6)), left)
            lit = rng.integers(1, 256, m, dtype=np.uint8)
            if esc: lit[rng.random(m) < 0.5] = rng.choice([0xFE, 0xFF])
            out.append(lit); left -= m
        x = np.concatenate(out)[:n]
    elif kind == 1:                                 # sparse non-zero bytes
        x = np.zeros(n, np.uint8)
        m = max(1, n // int(rng.integers(2, 3000)))
        x[rng.integers(0, n, m)] = rng.integers(1, 256, m, dtype=np.uint8)
    elif kind == 2:                                 # mostly escapes: expands, the transform declines
        x = rng.choice(np.array([0xFE, 0xFF, 1, 0], np.uint8), n, p=[0.4, 0.4, 0.1, 0.1])
    elif kind == 3:                                 # all zero / zero except the ends
        x = np.zeros(n, np.uint8)
        if rng.random() < 0.5: x[0] = rng.integers(1, 256)
        if rng.random() < 0.5: x[-1] = rng.integers(1, 256)
    elif kind == 4:                                 # random with a bias to small values (post-RANK look)
        x = np.minimum(rng.geometric(0.45, n) - 1, 255).astype(np.uint8)
    else:
        x = rng.integers(0, 256, n, dtype=np.uint8)
    return np.ascontiguousarray(x.astype(np.uint8))


t0 = time.time(); calls = blocks = bad = 0
while time.time() - t0 < budget:
    B = int(rng.integers(1, 9))
    bl = [block() for _ in range(B)]
    bs = max(len(b) for b in bl)
    inp = np.zeros((B, bs), np.uint8); lens = np.array([len(b) for b in bl], np.int32)
    for i, b in enumerate(bl): inp[i, :len(b)] = b
    ostride = kz.max_block_stream_bytes(bs)
    out = np.zeros((B, ostride), np.uint8)
    res = kz.encode_blocks(ctx, "ZRLT", "NONE", inp, bs, lens, out, ostride)
    bits = np.array([r.bits for r in res], np.int64)
    dec = np.zeros((B, bs), np.uint8)
    res2 = kz.decode_blocks(ctx, "ZRLT", "NONE", bs, out, ostride, bits, dec, bs)
    for i, b in enumerate(bl):
        so, w, sf, pl = oracle.encode_block("ZRLT", "NONE", b)
        ok = res[i].status == 0 and (res[i].bits, res[i].skipFlags, res[i].length) == (w, sf, pl) and out[i, :(w + 7) // 8].tobytes() == so
        ok = ok and res2[i].status == 0 and dec[i, :len(b)].tobytes() == b.tobytes()
        if not ok:
            bad += 1
            np.save(os.path.join(ROOT, "gpurun_out", "zrlt_fuzz_fail_%d_%d.npy" % (seed, blocks + i)), b)
            print("MISMATCH block of", len(b), "bytes in a batch of", B, "status", res[i].status, res2[i].status, flush=True)
    calls += 1; blocks += B
print("%d calls, %d blocks, %d mismatches in %.0f s" % (calls, blocks, bad, time.time() - t0))
sys.exit(1 if bad else 0)
