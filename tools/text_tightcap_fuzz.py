"""TEXT and UTF (the host stages: no GPU) against the oracle with buffers cut to the byte and damaged inputs: forward with the
destination at getMaxEncodedLength and at the output length (-1 / 0 / +1), inverse of valid and corrupted stage outputs with the
destination at the output length - 17 .. + 16.  Runs anywhere.   python tools/text_tightcap_fuzz.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kanzi_amd as kz
import oracle, textgen, refinputs

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
print("seed", seed, flush=True)
L = kz.load_library()
t0 = time.time(); cases = bad = 0; per = {}
while time.time() - t0 < budget:
    name = "TEXT" if rng.random() < 0.6 else "UTF"
    ent = ["ANS0", "FPAQ", "HUFFMAN", "NONE"][int(rng.integers(0, 4))]
    bsz = int(rng.choice([65536, 1 << 20, 4 << 20]))
    n = int(rng.choice([rng.integers(64, 600), rng.integers(600, 20000), rng.integers(20000, 120000)]))
    k = int(rng.integers(0, 6))
    s = int(rng.integers(0, 1 << 30))
    if name == "UTF": src = textgen.utf8(n, s, bom=bool(k & 1)) if k < 4 else textgen.english(n, s)
    elif k == 0: src = textgen.english(n, s)
    elif k == 1: src = textgen.english(n, s, crlf=True)
    elif k == 2: src = textgen.xml(n, s)
    elif k == 3: src = textgen.many_words(n, s)
    elif k == 4: src = textgen.english(n, s, sprinkle=bytes([0x0F, 0x0E, 0x80, 0xFF]))
    else: src = textgen.utf8(n, s)
    src = bytes(src)[:n]
    if len(src) < 16: continue
    oracle.set_transform_ctx(ent, bsz)
    try:
        ok0, good, _ = oracle.transform_forward(name, src, None, 0)
    except oracle.TransformThrows:
        continue
    mx = int(L.kz_transform_max_encoded_len(kz.TRANSFORM_IDS[name], len(src)))
    caps = sorted(set(max(1, c) for c in (mx - 1, mx, mx + 1, len(good) - 1, len(good), len(good) + 1, len(src)))) if ok0 else [mx, mx - 1]
    for cap in caps:
        try:
            ro = oracle.transform_forward(name, src, cap, 0)
        except oracle.TransformThrows:
            continue
        try:
            rp = kz.host_stage_forward(name, src, ent, bsz, 0, cap)
        except kz.KanziError:
            rp = (False, b"", 0)
        cases += 1; per[name + ">"] = per.get(name + ">", 0) + 1
        if not (bool(rp[0]) == bool(ro[0]) and (not ro[0] or (rp[1] == ro[1] and rp[2] == ro[2]))):
            bad += 1
            print("MISMATCH forward", name, ent, "n", len(src), "max", mx, "out", len(good), "cap", cap, "oracle", ro[0], len(ro[1]), "hip", rp[0], len(rp[1]), flush=True)
    if not ok0: continue
    data = good if rng.random() < 0.4 else bytes(refinputs.corrupt(rng, good, int(rng.integers(0, 8))))
    if not data: continue
    okL, outL = oracle.transform_inverse(name, data, 4 * len(src) + 65536)
    Ln = len(outL) if okL else len(src)
    for cap in sorted(set(max(1, Ln + d) for d in (-17, -2, -1, 0, 1, 2, 16))):
        ro = oracle.transform_inverse(name, data, cap)
        try:
            rp = kz.host_stage_inverse(name, data, cap, bsz)
        except kz.KanziError:
            rp = (False, b"")
        cases += 1; per[name] = per.get(name, 0) + 1
        if not (bool(rp[0]) == bool(ro[0]) and (not ro[0] or rp[1] == ro[1])):
            bad += 1
            print("MISMATCH inverse", name, "n", len(data), "L", Ln, "cap", cap, "oracle", ro[0], len(ro[1]), "hip", rp[0], len(rp[1]), flush=True)
    if bad > 30: break
print("%d cases %s, %d mismatches in %.0f s" % (cases, per, bad, time.time() - t0))
sys.exit(1 if bad else 0)
