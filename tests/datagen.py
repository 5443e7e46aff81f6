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


# ---- stand-ins for silesia.tar's binary members (bench.py shapes.silesia_mix_level5_exact) ----
def exe_like(n, seed):
    """Executable-like, mid entropy (silesia: mozilla, ooffice): instruction-like records -- a Zipf opcode, a ModRM-like byte tied to
    it, then 0 / 1 / 4 bytes of displacement or immediate (small little-endian values: many 0x00 / 0xFF bytes; call targets with
    random low bytes) -- with pointer tables (ascending 4-byte addresses) and short strings in between."""
    rng = np.random.Generator(np.random.PCG64(_splitmix(seed * 2 + 1)))
    m = n // 3 + 64                                                    # instructions of a fresh stream (>= 2 bytes each)
    ops = rng.permutation(256)[np.minimum(rng.zipf(1.25, m) - 1, 199)].astype(np.uint8)
    modrm = ((ops.astype(np.int64) * 7 + rng.integers(0, 8, m) * 8 + rng.integers(0, 4, m) * 64) & 0xFF).astype(np.uint8)
    kind = rng.choice(4, m, p=[0.45, 0.25, 0.2, 0.1])                  # 0: none, 1: disp8, 2: imm32 small, 3: rel32 call
    L = np.array([2, 3, 6, 6], dtype=np.int64)[kind]
    tot = int(L.sum())
    off = np.cumsum(L) - L
    fresh = np.zeros(tot + 8, dtype=np.uint8)
    fresh[off] = ops
    fresh[off + 1] = modrm
    k1 = kind == 1
    fresh[off[k1] + 2] = (rng.integers(-16, 17, int(k1.sum())) * 4) & 0xFF
    k2 = kind == 2
    v2 = rng.integers(-64, 512, int(k2.sum())).astype(np.int64) & 0xFFFFFFFF
    k3 = kind == 3
    v3 = rng.integers(-(1 << 18), 1 << 18, int(k3.sum())).astype(np.int64) & 0xFFFFFFFF
    for j in range(4):
        fresh[off[k2] + 2 + j] = (v2 >> (8 * j)) & 0xFF
        fresh[off[k3] + 2 + j] = (v3 >> (8 * j)) & 0xFF
    fresh = fresh[:tot]
    # compiled code repeats itself (prologues, inlined helpers, template instances): the stream is a Zipf mix of 6000 recurring
    # fragments of 8 .. 96 bytes cut from the fresh stream, and fresh bytes in between
    nfrag = 6000
    fs = rng.integers(0, tot - 128, nfrag)
    fl = rng.integers(8, 97, nfrag)
    pieces = n // 24 + 64
    pick = np.minimum(rng.zipf(1.15, pieces) - 1, nfrag - 1)
    isfresh = rng.random(pieces) < 0.35
    ps = np.where(isfresh, rng.integers(0, tot - 128, pieces), fs[pick])
    pl = np.where(isfresh, rng.integers(4, 40, pieces), fl[pick])
    total = int(pl.sum())
    po = np.cumsum(pl) - pl
    src = np.repeat(ps - po, pl) + np.arange(total)
    out = fresh[src].copy()
    tot = total
    # every 64 KiB: a 4 KiB pointer table and a 2 KiB string table
    words = [b"GetProcAddress\0", b"LoadLibraryA\0", b"kernel32.dll\0", b"memcpy\0", b"malloc\0", b".text\0", b".rdata\0", b"Error: %s\n\0", b"%d.%d.%d\0", b"__imp_\0"]
    for base in range(32768, min(tot, n) - 8192, 65536):
        addr = (0x00401000 + np.cumsum(rng.integers(4, 64, 1024)) * 4).astype(np.int64)
        tb = np.zeros(4096, dtype=np.uint8)
        for j in range(4):
            tb[j::4] = (addr >> (8 * j)) & 0xFF
        out[base:base + 4096] = tb
        s = b"".join(words[int(i)] for i in rng.integers(0, len(words), 260))[:2048]
        out[base + 4096:base + 4096 + len(s)] = np.frombuffer(s, dtype=np.uint8)
    if len(out) < n:
        out = np.resize(out, n)
    return np.ascontiguousarray(out[:n])


def sensor_like(n, seed):
    """Poorly compressible binary (silesia: x-ray, mr, sao): 16-bit little-endian samples of a smooth signal plus noise -- the high
    bytes move slowly, the low bytes are close to random -- in rows with a short header."""
    rng = np.random.Generator(np.random.PCG64(_splitmix(seed * 2 + 2)))
    m = n // 2 + 8
    walk = np.cumsum(rng.integers(-40, 41, m)) + 20000
    noise = rng.normal(0, 40, m).astype(np.int64)
    v = (walk + noise) & 0xFFFF
    out = np.empty(2 * m, dtype=np.uint8)
    out[0::2] = v & 0xFF
    out[1::2] = v >> 8
    for base in range(0, 2 * m - 16, 4096):                            # row header: magic, row number, zeros
        out[base:base + 4] = np.frombuffer(b"ROW\0", dtype=np.uint8)
        out[base + 4] = (base >> 12) & 0xFF
        out[base + 5] = (base >> 20) & 0xFF
        out[base + 6:base + 16] = 0
    return np.ascontiguousarray(out[:n])
