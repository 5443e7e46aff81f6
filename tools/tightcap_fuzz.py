"""Inverse transforms with the output buffer cut to the byte: for valid and corrupted inputs the oracle's output length L is found
with a roomy buffer, then oracle and HIP run with capacities L - 1, L, L + 1 (and a few around): same verdict, same bytes.  Exact-fit
is where the reference's loops leave through a different door (ZRLT.java:214 was found this way).
   python tools/tightcap_fuzz.py [seconds] [seed] [names...]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kanzi_amd as kz
import oracle, datagen, refinputs

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
names = sys.argv[3:] or ["ZRLT", "RANK", "MTFT", "SRT", "LZ", "LZX", "BWT", "MM", "PACK", "DNA"]
rng = np.random.default_rng(seed)
ctx = kz.Context(0)
print("seed", seed, names, flush=True)


def codec(name):
    if name == "RANK": return kz.SBRT(ctx, 2)
    if name == "MTFT": return kz.SBRT(ctx, 1)
    if name == "LZ": return kz.LZCodec(ctx, kz.LZ_TYPE)
    if name == "LZX": return kz.LZCodec(ctx, kz.LZX_TYPE)
    if name in ("PACK", "DNA"): return kz.AliasCodec(ctx, onlyDNA=(name == "DNA"))
    return {"BWT": kz.BWTBlockCodec, "ZRLT": kz.ZRLT, "SRT": kz.SRT, "MM": kz.FSDCodec}[name](ctx)


def source(name, n):
    k = int(rng.integers(0, 5))
    if name == "MM": return refinputs.multimedia_like(k, n, seed=int(rng.integers(0, 1000)))
    if name in ("PACK", "DNA"):
        al = refinputs.alias_inputs()
        b = al[int(rng.integers(0, len(al)))][1]
        return (b * (n // max(1, len(b)) + 1))[:n]
    pre = datagen.block(int(rng.integers(0, 1 << 16)), n, k).tobytes()
    if name in ("SRT", "RANK", "MTFT", "ZRLT"):
        pre = oracle.transform_forward("BWT", pre)[1]
        if name == "ZRLT": pre = oracle.transform_forward("RANK", pre)[1]
    return pre


t0 = time.time(); cases = bad = 0
per = {}
while time.time() - t0 < budget:
    name = names[int(rng.integers(0, len(names)))]
    n = int(rng.choice([rng.integers(16, 400), rng.integers(400, 6000), rng.integers(6000, 40000)]))
    pre = source(name, n)
    if len(pre) < 16: continue
    ok, good = oracle.transform_forward(name, pre)
    if not ok: continue
    # forward with the output buffer at, just below and just above the reference's getMaxEncodedLength and the actual output length
    mx = codec(name).getMaxEncodedLength(len(pre))
    for cap in sorted(set(max(1, c) for c in (mx - 1, mx, mx + 1, len(good) - 1, len(good), len(good) + 1, len(pre)))):
        try:
            ok_o, o = oracle.transform_forward(name, pre, cap)
        except oracle.TransformThrows:
            continue
        src = kz.SliceByteArray(np.frombuffer(pre, dtype=np.uint8).copy(), len(pre), 0)
        dst = kz.SliceByteArray(np.zeros(cap, dtype=np.uint8), cap, 0)
        ctx.set_data_type(0)                              # the block's "dataType" entry: LZ / MM read it, MM and others leave theirs behind
        try:
            ok_p = codec(name).forward(src, dst)
        except kz.KanziError:
            ok_p = False
        cases += 1; per[name + ">"] = per.get(name + ">", 0) + 1
        if not (bool(ok_p) == bool(ok_o) and (not ok_o or bytes(dst.array[:dst.index]) == o)):
            bad += 1
            np.save(os.path.join(ROOT, "gpurun_out", "tightcap_fwd_fail_%s_%d_%d.npy" % (name, seed, cases)), np.frombuffer(pre, np.uint8))
            print("MISMATCH forward", name, "n", len(pre), "max", mx, "out", len(good), "cap", cap, "oracle", ok_o, (len(o) if ok_o else -1), "hip", bool(ok_p), dst.index, flush=True)
    data = good if rng.random() < 0.4 else bytes(refinputs.corrupt(rng, good, int(rng.integers(0, 8))))
    if len(data) == 0: continue
    okL, outL = oracle.transform_inverse(name, data, 4 * len(pre) + 65536)
    L = len(outL) if okL else len(pre)
    for cap in sorted(set(max(1, L + d) for d in (-17, -2, -1, 0, 1, 2, 16))):
        ok_o, o = oracle.transform_inverse(name, data, cap)
        src = kz.SliceByteArray(np.frombuffer(data, dtype=np.uint8).copy(), len(data), 0)
        dst = kz.SliceByteArray(np.zeros(cap, dtype=np.uint8), cap, 0)
        try:
            ok_p = codec(name).inverse(src, dst)
        except kz.KanziError as e:
            ok_p = False
        goodcase = bool(ok_p) == bool(ok_o) and (not ok_o or bytes(dst.array[:dst.index]) == o)
        cases += 1; per[name] = per.get(name, 0) + 1
        if not goodcase:
            bad += 1
            np.save(os.path.join(ROOT, "gpurun_out", "tightcap_fail_%s_%d_%d.npy" % (name, seed, cases)), np.frombuffer(data, np.uint8))
            print("MISMATCH", name, "n", len(data), "L", L, "cap", cap, "oracle", ok_o, (len(o) if ok_o else -1), "hip", bool(ok_p), dst.index, flush=True)
print("%d cases %s, %d mismatches in %.0f s" % (cases, per, bad, time.time() - t0))
sys.exit(1 if bad else 0)
