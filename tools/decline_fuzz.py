"""Forward transforms at the point where they decline ("would not shrink enough"): LZ / LZX on blocks that are random for a fraction f
of their length and repeats for the rest, f swept finely around 0.99 (LZCodec.java:310, :561-566); PACK / DNA on alphabets of 1 .. 40
and 60 .. 70 and 250 .. 256 distinct symbols (AliasCodec's 16- / 64-symbol packing rules); MM on signals whose delta gain is marginal;
verdict, bytes and the block's data type against the oracle.   python tools/decline_fuzz.py [seconds] [seed]"""
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


def codec(name):
    if name == "LZ": return kz.LZCodec(ctx, kz.LZ_TYPE)
    if name == "LZX": return kz.LZCodec(ctx, kz.LZX_TYPE)
    if name in ("PACK", "DNA"): return kz.AliasCodec(ctx, onlyDNA=(name == "DNA"))
    return kz.FSDCodec(ctx)


t0 = time.time(); cases = bad = 0; per = {}
while time.time() - t0 < budget:
    name = ["LZ", "LZX", "PACK", "DNA", "MM"][int(rng.integers(0, 5))]
    n = int(rng.choice([rng.integers(64, 1100), rng.integers(1100, 9000), rng.integers(9000, 70000)]))
    if name in ("LZ", "LZX"):
        f = 1.0 - float(rng.choice([0.0, 0.002, 0.005, 0.008, 0.0095, 0.0099, 0.01, 0.0101, 0.0105, 0.011, 0.012, 0.015, 0.02, 0.03, 0.05]))
        r = int(n * f)
        x = rng.integers(0, 256, n, dtype=np.uint8)
        if r < n:
            rep = rng.integers(0, 256, int(rng.integers(4, 40)), dtype=np.uint8)
            x[r:] = np.resize(rep, n - r)
    elif name in ("PACK", "DNA"):
        k = int(rng.choice([rng.integers(1, 41), rng.integers(60, 71), rng.integers(250, 257)]))
        syms = rng.permutation(256)[:k].astype(np.uint8)
        if name == "DNA" and rng.random() < 0.7: syms = np.frombuffer(b"ACGTNacgtn\n>", np.uint8)[:max(1, min(k, 12))]
        x = syms[rng.integers(0, len(syms), n)]
    else:
        ch = int(rng.choice([1, 2, 3, 4]))
        base = np.cumsum(rng.integers(-3, 4, n)).astype(np.int64)
        noise = rng.integers(0, int(rng.choice([1, 2, 4, 16, 64, 256])), n)
        x = ((base + noise) & 0xFF).astype(np.uint8)
        if ch > 1: x = np.repeat(x[: n // ch + 1], ch)[:n]
    data = np.ascontiguousarray(x).tobytes()
    try:
        ok_o, o, dt_o = oracle.transform_forward(name, data, None, 0)
    except oracle.TransformThrows:
        continue
    c = codec(name)
    cap = c.getMaxEncodedLength(len(data))
    src = kz.SliceByteArray(np.frombuffer(data, dtype=np.uint8).copy(), len(data), 0)
    dst = kz.SliceByteArray(np.zeros(cap, dtype=np.uint8), cap, 0)
    ctx.set_data_type(0)
    try:
        ok_p = c.forward(src, dst)
    except kz.KanziError:
        ok_p = False
    dt_p = ctx.get_data_type()
    cases += 1; per[name] = per.get(name, (0, 0)); per[name] = (per[name][0] + 1, per[name][1] + int(bool(ok_o)))
    good = bool(ok_p) == bool(ok_o) and (not ok_o or bytes(dst.array[:dst.index]) == o) and dt_p == dt_o
    if not good:
        bad += 1
        print("MISMATCH", name, "n", n, "oracle", ok_o, len(o), dt_o, "hip", bool(ok_p), dst.index, dt_p, flush=True)
        if bad > 20: break
print("%d cases %s (calls, applied), %d mismatches in %.0f s" % (cases, per, bad, time.time() - t0))
sys.exit(1 if bad else 0)
