"""Forward BWT of the HIP path against the oracle on inputs that steer the suffix sort through its paths; reports every mismatch.
   python tools/bwt_diag.py [quick]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes
import numpy as np
import kanzi_amd as kz
import oracle, datagen

ctx = kz.Context(0)
rng = np.random.default_rng(5)

def fwd(data):
    tid = kz.BWT_TYPE
    cap = ctx.lib.kz_transform_max_encoded_len(tid, len(data))
    out = np.zeros(cap + 64, dtype=np.uint8)
    p = ctypes.c_int32(0)
    a = np.frombuffer(data, dtype=np.uint8)
    rc = ctx.lib.kz_transform_forward(ctx.h, tid, a.ctypes.data, len(data), out.ctypes.data, cap, ctypes.addressof(p))
    assert rc >= 0, ctx.error()
    return rc == 1, out[:p.value].tobytes()

def repeats(n, nwords, wlen):
    words = rng.integers(0, 256, (nwords, wlen), dtype=np.uint8)
    idx = rng.integers(0, nwords, n // wlen + 1)
    return words[idx].reshape(-1)[:n].tobytes()

cases = []
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
for n in (65536, 65537, 100000, 262144):
    for c in range(5):
        cases.append(("class%d n=%d" % (c, n), datagen.block(c, n, c).tobytes()))
cases.append(("zeros 300000", bytes(300000)))
cases.append(("zeros 65536", bytes(65536)))
cases.append(("ab period", (b"ab" * 200000)))
cases.append(("period 25", (b"abcdefghijklmnopqrstuvwxy" * 40000)[:999983]))
cases.append(("3 symbols", rng.integers(0, 3, 500000, dtype=np.uint8).tobytes()))
cases.append(("2 symbols", rng.integers(0, 2, 400000, dtype=np.uint8).tobytes()))
cases.append(("words 3x33", repeats(900000, 3, 33)))
cases.append(("words 12x40", repeats(1 << 20, 12, 40)))
cases.append(("words 900x16", repeats(300000, 900, 16)))
cases.append(("ends in zeros", rng.integers(0, 256, 200000, dtype=np.uint8).tobytes() + bytes(70000)))
cases.append(("zeros then one", bytes(99999) + b"\x01"))
cases.append(("ff run", b"\xff" * 150000 + rng.integers(0, 256, 50000, dtype=np.uint8).tobytes()))
if not quick:
    for c in range(5):
        cases.append(("class%d 4MiB" % c, datagen.block(c, 4 << 20, c).tobytes()))
    cases.append(("class0 4MiB+1", datagen.block(7, (4 << 20) + 1, 0).tobytes()))
bad = 0
for name, d in cases:
    t0 = time.time()
    ok_o, enc_o = oracle.transform_forward("BWT", d)
    t1 = time.time()
    ok_g, enc_g = fwd(d)
    t2 = time.time()
    if ok_o == ok_g and enc_o == enc_g:
        print("ok    %-22s (oracle %.2fs hip %.2fs)" % (name, t1 - t0, t2 - t1), flush=True)
    else:
        bad += 1
        a = np.frombuffer(enc_o, np.uint8); g = np.frombuffer(enc_g, np.uint8)
        m = min(len(a), len(g))
        diff = np.nonzero(a[:m] != g[:m])[0]
        print("FAIL  %-22s ok %s/%s len %d/%d first diff %s, %d differ" % (name, ok_o, ok_g, len(a), len(g), diff[:5], len(diff)), flush=True)
print("%d of %d failed" % (bad, len(cases)))
sys.exit(1 if bad else 0)
