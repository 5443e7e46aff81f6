"""Single-block transform API vs the oracle over random sizes and contents (status, bytes, "dataType" entry, inverse).
SEEDS=1,2 CASES=400 python tools/transform_campaign.py   (diagnostic)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
import kanzi_amd as kz, oracle, refinputs, datagen
from test_gpu_parity import _fuzz_input, _codec

ctx = kz.Context(0)
names = ["BWT", "RANK", "MTFT", "ZRLT", "SRT", "LZ", "LZX", "MM", "PACK", "DNA"]
alias = [d for _, d in refinputs.alias_inputs()]
seeds = [int(x) for x in os.environ.get("SEEDS", "1,2").split(",")]
cases = int(os.environ.get("CASES", "400"))
bad = 0; done = 0
for seed in seeds:
    rng = np.random.default_rng(seed)
    for case in range(cases):
        n = int(rng.choice([int(rng.integers(0, 40)), int(rng.integers(1000, 1100)), int(rng.integers(0, 6000)), int(rng.integers(0, 70000)), int(rng.integers(0, 600000))]))
        pick = int(rng.integers(0, 4))
        if pick == 0: data = _fuzz_input(rng, n).tobytes()
        elif pick == 1: data = refinputs.multimedia_like(int(rng.integers(0, 5)), n, seed=case) if n else b""
        elif pick == 2:
            srcb = alias[int(rng.integers(0, len(alias)))]; data = (srcb * (n // len(srcb) + 1))[:n]
        else: data = datagen.block(int(rng.integers(0, 40)), n).tobytes() if n else b""
        if rng.integers(0, 8) == 0 and n >= 4:                                   # a magic number in front
            data = bytes(rng.choice([b"RIFF", b"BM\x00\x00", b"P6\n3", b"\x89PNG", b"\x7fELF", b"PK\x03\x04", b"\xff\xd8\xff\xe0"])) + data[4:]
        name = names[int(rng.integers(0, len(names)))]
        dt0 = int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9]))
        ok_o, out_o, dt_o = oracle.transform_forward(name, data, data_type=dt0)
        ctx.set_data_type(dt0)
        codec = _codec(ctx, name)
        cap = codec.getMaxEncodedLength(len(data))
        dst = kz.SliceByteArray(np.zeros(max(cap, 1), dtype=np.uint8), cap, 0)
        ok_p = codec.forward(kz.SliceByteArray(np.frombuffer(data, dtype=np.uint8).copy(), len(data), 0), dst)
        good = bool(ok_p) == bool(ok_o) and ctx.get_data_type() == dt_o and (not ok_o or bytes(dst.array[:dst.index]) == out_o)
        if good and ok_o and len(out_o):
            back = kz.SliceByteArray(np.zeros(len(data) + 64, dtype=np.uint8), len(data) + 64, 0)
            okb = _codec(ctx, name).inverse(kz.SliceByteArray(np.frombuffer(out_o, dtype=np.uint8).copy(), len(out_o), 0), back)
            good = bool(okb) and bytes(back.array[:back.index]) == data
        done += 1
        if not good:
            bad += 1
            print("FAIL seed", seed, "case", case, name, "n", n, "pick", pick, "dt0", dt0, "oracle", ok_o, dt_o, "hip", bool(ok_p), ctx.get_data_type(), flush=True)
ctx.set_data_type(0)
print("cases", done, "failures", bad)
