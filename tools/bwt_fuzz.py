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
    if n <= 0:
        return np.zeros(0, np.uint8)
    k = int(rng.integers(0, 14))
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
    if k == 10:
        a = piece(n // 2)
        return np.concatenate([a, a[:n - len(a)]])             # the second half repeats the first: repeats of n / 2
    if k == 13:
        # noise with planted repeats: no 2-byte prefix reaches the LDS sort's capacity (the trie has no expanded node: the block is
        # a candidate for round 0's lazy ranks) and yet some suffixes stay live after round 0, few or many
        x = rng.integers(0, 256, n, dtype=np.uint8)
        for _ in range(int(rng.integers(0, 6))):
            L = int(min(n, rng.choice([6, 7, 8, 40, 1000, 70000])))
            a, b = int(rng.integers(0, n - L + 1)), int(rng.integers(0, n - L + 1))
            x[b:b + L] = x[a:a + L].copy()
        return x
    return structured(n)


def structured(n):
    """strings that are hard for a sort by prefix doubling: every group stays large for many rounds"""
    if n <= 0:
        return np.zeros(0, np.uint8)
    k = int(rng.integers(0, 5))
    a, b = (int(v) for v in rng.choice(256, 2, replace=False))
    if k == 0:                                                  # Fibonacci word
        x, y = np.array([a], np.uint8), np.array([a, b], np.uint8)
        while len(y) < n: x, y = y, np.concatenate([y, x])
        return y[:n]
    if k == 1:                                                  # Thue-Morse
        i = np.arange(n, dtype=np.uint32)
        bits = i ^ (i >> 16); bits ^= bits >> 8; bits ^= bits >> 4; bits ^= bits >> 2; bits ^= bits >> 1
        return np.where(bits & 1, b, a).astype(np.uint8)
    if k == 2:                                                  # runs whose lengths sit around the sorter's capacities and powers of two
        out, left = [], n
        while left > 0:
            L = int(rng.choice([1, 2, 3, 5, 6, 7, 63, 64, 65, 255, 256, 3839, 3840, 7679, 7680, 7681, 8192, 65535, 65536, 100000]))
            L = min(L + int(rng.integers(0, 2)), left)
            out.append(np.full(L, rng.choice([a, b, 0, 255]), np.uint8)); left -= L
        return np.concatenate(out)
    if k == 3:                                                  # a long period with sparse mutations
        per = int(rng.integers(2, 70000))
        x = np.resize(rng.integers(0, int(rng.integers(2, 257)), per, dtype=np.uint8), n).copy()
        m = int(rng.integers(0, 1 + n // 1000))
        if m: x[rng.integers(0, n, m)] ^= np.uint8(1)
        return x
    x = np.resize(np.arange(int(rng.integers(2, 257)), dtype=np.uint8), n).copy()   # a counter (every bigram distinct, every period equal)
    return x[::-1].copy() if rng.random() < 0.5 else x


def block():
    n = int(rng.choice([rng.integers(1, 70000), rng.integers(60000, 300000), rng.integers(300000, (4 << 20) + 60000),
                        rng.choice([65535, 65536, 65537, (4 << 20) - 1, 4 << 20, (4 << 20) + 1, (4 << 20) + 65536, (4 << 20) + 65537])], p=[0.15, 0.4, 0.35, 0.1]))
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
