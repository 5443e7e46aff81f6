"""The ZRLT INVERSE on arbitrary bytes against the oracle (same verdict, same output bytes): strings dense in the bytes the decoder
branches on (0x00 / 0x01 digits, 0xFF escapes, 0xFE), runs of escapes and of digits of every length (parity of the escape runs,
over-long digit runs with Java int wrap-around), lengths at the row / wave / tile seams.   python tools/zrlt_inv_fuzz.py [seconds] [seed]"""
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
codec = kz.ZRLT(ctx)
print("seed", seed, flush=True)
LENS = [1, 2, 3, 5, 8, 15, 16, 17, 29, 30, 31, 32, 33, 34, 40, 62, 63, 64, 65, 66, 100, 127, 128, 129, 200, 1023, 1024, 1025, 4095, 4096, 4097]


def garbage():
    n = int(rng.choice([rng.integers(1, 200), rng.integers(200, 5000), rng.integers(5000, 70000),
                        rng.choice([63, 64, 65, 127, 128, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 8192, 8193, 12288])]))
    kind = int(rng.integers(0, 5))
    if kind == 0:                                   # segments: digit runs, escape runs, literals
        out, left = [], n
        while left > 0:
            L = min(int(rng.choice(LENS)), left)
            k = int(rng.integers(0, 4))
            if k == 0: seg = rng.integers(0, 2, L, dtype=np.uint8)
            elif k == 1: seg = np.full(L, 0xFF, np.uint8)
            elif k == 2: seg = rng.integers(2, 256, min(L, 6), dtype=np.uint8)
            else: seg = rng.choice(np.array([0, 1, 0xFF, 0xFE, 2, 7], np.uint8), L)
            out.append(seg); left -= len(seg)
        x = np.concatenate(out)[:n]
    elif kind == 1:
        x = rng.choice(np.array([0, 1, 0xFF, 0xFE, 2, 200], np.uint8), n, p=[0.3, 0.2, 0.2, 0.1, 0.1, 0.1])
    elif kind == 2:
        x = rng.integers(0, 256, n, dtype=np.uint8)
    elif kind == 3:                                 # a valid forward output, then corrupted in a few places
        src = np.minimum(rng.geometric(0.4, max(16, n)) - 1, 255).astype(np.uint8)
        ok, enc = oracle.transform_forward("ZRLT", src.tobytes())
        x = np.frombuffer(enc if ok else src.tobytes(), np.uint8).copy()
        m = int(rng.integers(0, 6))
        if m and len(x): x[rng.integers(0, len(x), m)] = rng.choice(np.array([0, 1, 0xFF, 0xFE], np.uint8), m)
    else:
        x = rng.integers(0, 2, n, dtype=np.uint8)   # digits only
        if n > 40 and rng.random() < 0.5: x[rng.integers(0, n, max(1, n // 40))] = 0xFF
    return np.ascontiguousarray(x.astype(np.uint8))


t0 = time.time(); cases = bad = applied = 0
while time.time() - t0 < budget:
    x = garbage()
    if len(x) == 0: continue
    cap = int(rng.choice([len(x) * 2 + 64, len(x) + 16, 1 << 16, 1 << 20, 4 << 20]))
    ok_o, o = oracle.transform_inverse("ZRLT", x.tobytes(), cap)
    src = kz.SliceByteArray(x.copy(), len(x), 0)
    dst = kz.SliceByteArray(np.zeros(cap, dtype=np.uint8), cap, 0)
    ok_p = codec.inverse(src, dst)
    good = bool(ok_p) == bool(ok_o) and (not ok_o or bytes(dst.array[:dst.index]) == o)
    applied += bool(ok_o)
    if not good:
        bad += 1
        np.save(os.path.join(ROOT, "gpurun_out", "zrlt_inv_fail_%d_%d.npy" % (seed, cases)), x)
        print("MISMATCH n", len(x), "cap", cap, "oracle ok", ok_o, "hip ok", bool(ok_p), "len", (len(o) if ok_o else -1), dst.index, flush=True)
    cases += 1
print("%d cases (%d applied), %d mismatches in %.0f s" % (cases, applied, bad, time.time() - t0))
sys.exit(1 if bad else 0)
