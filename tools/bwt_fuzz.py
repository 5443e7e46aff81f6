"""Randomised differential run of the forward BWT (trie rounds, key rounds, LSD rounds) against the oracle: ragged batches of
blocks of random size and structure through kz_encode_blocks("BWT", "NONE").   python tools/bwt_fuzz.py [seconds] [seed]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kanzi_amd as kz
import oracle, datagen

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
ctx = kz.Context(0)
print("seed", seed, flush=True)


def piece(n):
    k = int(rng.integers(0, 11))
    if k < 5:
        return datagen.block(int(rng.integers(0, 1 << 20)), n, k)
    if k == 5:
        return np.zeros(n, np.uint8) + np.uint8(rng.integers(0, 256))
    if k == 6:
        per = int(rng.integers(1, 300))
        return np.resize(rng.integers(0, 256, per, dtype=np.uint8), n)
    if k == 7:
        return rng.integers(0, int(rng.integers(2, 9)), n, dtype=np.uint8)
    if k == 8:
        nw, wl = int(rng.integers(2, 2000)), int(rng.integers(3, 60))
        words = rng.integers(0, 256, (nw, wl), dtype=np.uint8)
        return words[rng.integers(0, nw, n // wl + 1)].reshape(-1)[:n]
    if k == 9:
        x = np.zeros(n, np.uint8)
        m = max(1, n // int(rng.integers(5, 5000)))
        x[rng.integers(0, n, m)] = rng.integers(1, 256, m, dtype=np.uint8)
        return x
    a = piece(n // 2)
    return np.concatenate([a, a[:n - len(a)]])                 # the second half repeats the first: repeats of n / 2


def block():
    n = int(rng.choice([rng.integers(1, 70000), rng.integers(60000, 300000), rng.integers(300000, (4 << 20) + 60000)], p=[0.15, 0.45, 0.4]))
    parts, left = [], n
    while left > 0:
        m = left if rng.random() < 0.5 else int(rng.integers(1, left + 1))
        parts.append(piece(m)[:m]); left -= m
    return np.concatenate(parts).astype(np.uint8)


t0 = time.time(); calls = blocks = bad = 0
while time.time() - t0 < budget:
    B = int(rng.integers(1, 6))
    bl = [block() for _ in range(B)]
    bs = max(len(b) for b in bl)
    inp = np.zeros((B, bs), np.uint8); lens = np.array([len(b) for b in bl], np.int32)
    for i, b in enumerate(bl): inp[i, :len(b)] = b
    ostride = kz.max_block_stream_bytes(bs)
    out = np.zeros((B, ostride), np.uint8)
    res = kz.encode_blocks(ctx, "BWT", "NONE", inp, bs, lens, out, ostride)
    for i, b in enumerate(bl):
        so, w, sf, pl = oracle.encode_block("BWT", "NONE", b)
        ok = res[i].status == 0 and (res[i].bits, res[i].skipFlags, res[i].length) == (w, sf, pl) and out[i, :(w + 7) // 8].tobytes() == so
        if not ok:
            bad += 1
            np.save(os.path.join(ROOT, "gpurun_out", "bwt_fuzz_fail_%d_%d.npy" % (seed, blocks + i)), b)
            print("MISMATCH block of", len(b), "bytes in a batch of", B, "status", res[i].status, flush=True)
    calls += 1; blocks += B
print("%d calls, %d blocks, %d mismatches in %.0f s" % (calls, blocks, bad, time.time() - t0))
sys.exit(1 if bad else 0)
