"""Inverse stages on corrupted input: product vs oracle, status and bytes.  Diagnostic."""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, kanzi_amd as kz, oracle, datagen

ctx = kz.Context(0)

def mk(name):
    if name == "RANK": return kz.SBRT(ctx, 2)
    if name == "MTFT": return kz.SBRT(ctx, 1)
    if name == "LZ": return kz.LZCodec(ctx, kz.LZ_TYPE)
    if name == "LZX": return kz.LZCodec(ctx, kz.LZX_TYPE)
    if name in ("PACK", "DNA"): return kz.AliasCodec(ctx, onlyDNA=(name == "DNA"))
    return {"BWT": kz.BWTBlockCodec, "ZRLT": kz.ZRLT, "SRT": kz.SRT, "MM": kz.FSDCodec}[name](ctx)

DEC = {"ANS0": kz.ANSRangeDecoder, "HUFFMAN": kz.HuffmanDecoder, "FPAQ": kz.FPAQDecoder}

def mutate(rng, good, kind):
    bad = bytearray(good)
    if kind == 0:
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(0, len(bad))); bad[pos] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:
        bad = bad[:int(rng.integers(1, len(bad)))]
    elif kind == 2:
        a = int(rng.integers(0, max(1, len(bad) - 64))); bad[a:a + 64] = bytes(rng.integers(0, 256, 64, dtype=np.uint8))
    elif kind == 3:
        a = int(rng.integers(0, max(1, len(bad) - 8))); del bad[a:a + int(rng.integers(1, 8))]
    elif kind == 4:   # header-weighted flips
        for _ in range(int(rng.integers(1, 3))):
            pos = int(rng.integers(0, min(len(bad), 300))); bad[pos] ^= 1 << int(rng.integers(0, 8))
    elif kind == 5:
        bad = bytearray(rng.integers(0, 256, len(bad), dtype=np.uint8).tobytes())
    elif kind == 6:   # zeroed tail (what a stopped entropy decode leaves behind)
        a = int(rng.integers(0, len(bad))); bad[a:] = bytes(len(bad) - a)
    else:             # a long run of 0/1 digits somewhere
        a = int(rng.integers(0, max(1, len(bad) - 80))); k = int(rng.integers(28, 70))
        bad[a:a + k] = bytes(rng.integers(0, 2, k, dtype=np.uint8))
    return bytes(bad)

stats = collections.Counter()
N = int(os.environ.get("N", "20000"))
import refinputs
rng = np.random.default_rng(int(os.environ.get("SEED", "123")))
alias = [d for _, d in refinputs.alias_inputs()]
for name in ["SRT", "ZRLT", "RANK", "MTFT", "BWT", "LZ", "LZX", "MM", "PACK"]:
    for src_kind in range(8):
        data = datagen.block(src_kind, N).tobytes()
        if name == "MM": data = refinputs.multimedia_like(src_kind % 5, N, seed=src_kind)
        if name == "PACK": data = alias[(0, 1, 5, 9, 11, 13, 15, 17)[src_kind]][:N]
        pre = data
        if name in ("SRT", "RANK", "MTFT", "ZRLT"):
            ok, pre = oracle.transform_forward("BWT", data)
            if name == "ZRLT": ok, pre = oracle.transform_forward("RANK", pre)
        ok, good = oracle.transform_forward(name, pre)
        if not ok: continue
        cap = N + max(512, N >> 4)
        for trial in range(40):
            bad = mutate(rng, good, trial % 8)
            ok_o, o = oracle.transform_inverse(name, bad, cap)
            if os.environ.get("TRACE"):
                os.makedirs("gpurun_out", exist_ok=True)
                open("gpurun_out/trace_last.bin", "wb").write(bad)
                open("gpurun_out/trace_last.txt", "w").write("%s src %d trial %d kind %d cap %d len %d ok_o %d\n" % (name, src_kind, trial, trial % 8, cap, len(bad), ok_o))
            src = kz.SliceByteArray(np.frombuffer(bad, dtype=np.uint8).copy(), len(bad), 0)
            dst = kz.SliceByteArray(np.zeros(cap, dtype=np.uint8), cap, 0)
            try: ok_p = mk(name).inverse(src, dst); p = bytes(dst.array[:dst.index])
            except kz.KanziError: ok_p, p = False, b""
            if bool(ok_o) != bool(ok_p): verdict = "status o=%d p=%d" % (ok_o, ok_p)
            elif ok_o and bytes(o) != p: verdict = "bytes differ (len o=%d p=%d)" % (len(o), len(p))
            else: verdict = "same"
            stats[(name, verdict)] += 1
            if verdict != "same" and stats[(name, verdict)] <= 3:
                print("DIFF", name, "src", src_kind, "trial", trial, "kind", trial % 8, verdict)
for ent in ["ANS0", "HUFFMAN", "FPAQ"]:
    for src_kind in (3, 1, 6):
        data = datagen.block(src_kind, 40000).tobytes()
        good, nbits = oracle.entropy_encode(ent, data)
        for trial in range(60):
            bad = mutate(rng, good, trial % 8)
            nb = min(nbits, len(bad) * 8)
            r, o, used = oracle.entropy_decode(ent, bad, nb, len(data))
            if os.environ.get("TRACE"):
                os.makedirs("gpurun_out", exist_ok=True)
                open("gpurun_out/trace_last.bin", "wb").write(bad)
                open("gpurun_out/trace_last.txt", "w").write("%s src %d trial %d kind %d nb %d count %d r %d\n" % (ent, src_kind, trial, trial % 8, nb, len(data), r))
            ok_o = (r == len(data))
            buf = np.zeros(len(data), dtype=np.uint8)
            try: ok_p = DEC[ent](ctx, bad, nb).decode(buf, 0, len(data)) == len(data)
            except kz.KanziError: ok_p = False
            p = bytes(buf)
            if ok_o != ok_p: verdict = "status o=%d p=%d" % (ok_o, ok_p)
            elif ok_o and o != p: verdict = "bytes differ"
            else: verdict = "same"
            stats[(ent, verdict)] += 1
            if verdict != "same" and stats[(ent, verdict)] <= 3:
                print("DIFF", ent, "src", src_kind, "trial", trial, "kind", trial % 8, verdict)
                os.makedirs("gpurun_out", exist_ok=True)
                open("gpurun_out/diff_%s_%d_%d.bin" % (ent, src_kind, trial), "wb").write(bad)
                print("   nb", nb, "count", len(data), "oracle r", r, "used", used)
for k, v in sorted(stats.items()): print(k, v)
