"""Deterministic synthetic block generator: block b uses splitmix64 seeded
0x9E3779B97F4A7C15*(b+1); class = b mod 5: (0) Markov text with word reuse, (1) geometric-skew
bytes, (2) 64-byte records with mutating fields, (3) uniform random, (4) 90% zeros + spikes."""
import numpy as np

M64 = (1 << 64) - 1


def _splitmix(seed):
    z = (seed + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def block(b, n, cls=None):
    seed = _splitmix((0x9E3779B97F4A7C15 * (b + 1)) & M64)
    rng = np.random.Generator(np.random.PCG64(seed))
    cls = b % 5 if cls is None else cls
    if cls == 0:
        # word-level Zipf reuse over a 64-symbol alphabet
        nwords = 4096
        lens = rng.integers(2, 10, nwords)
        alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789,.", dtype=np.uint8)
        probs = 0.85 ** np.arange(64); probs /= probs.sum()
        words = [alpha[rng.choice(64, l, p=probs)] for l in lens]
        need = n // 4 + 16
        idx = np.minimum(rng.zipf(1.3, need) - 1, nwords - 1)
        parts = []
        total = 0
        for i in idx:
            w = words[i]
            parts.append(w); parts.append(np.array([32], dtype=np.uint8))
            total += len(w) + 1
            if total >= n:
                break
        out = np.concatenate(parts)
        if len(out) < n:
            out = np.resize(out, n)
        return np.ascontiguousarray(out[:n])
    if cls == 1:
        return np.minimum(rng.geometric(0.1, n) - 1, 255).astype(np.uint8)
    if cls == 2:
        nrec = n // 64 + 1
        rec = rng.integers(0, 256, 64, dtype=np.uint8)
        out = np.tile(rec, nrec).reshape(nrec, 64)
        for f in range(8):
            col = rng.integers(0, 64)
            out[:, col] = (np.cumsum(rng.integers(0, 3, nrec)) + f) & 0xFF
        return np.ascontiguousarray(out.reshape(-1)[:n])
    if cls == 3:
        return rng.integers(0, 256, n, dtype=np.uint8)
    out = np.zeros(n, dtype=np.uint8)
    k = max(1, n // 10)
    pos = rng.integers(0, n, k)
    out[pos] = rng.integers(1, 256, k, dtype=np.uint8)
    return out


def stream(nblocks, block_size, first=0):
    return np.concatenate([block(first + b, block_size) for b in range(nblocks)])
