"""Single-block entropy API vs the oracle over random sizes (chunk boundaries included): bits, bytes, bits consumed.
SEEDS=1,2 CASES=300 python tools/entropy_campaign.py   (diagnostic)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
import kanzi_amd as kz, oracle, datagen
from test_gpu_parity import _fuzz_input

ctx = kz.Context(0)
ENC = {"ANS0": kz.ANSRangeEncoder, "HUFFMAN": kz.HuffmanEncoder, "FPAQ": kz.FPAQEncoder, "NONE": kz.NullEntropyEncoder}
DEC = {"ANS0": kz.ANSRangeDecoder, "HUFFMAN": kz.HuffmanDecoder, "FPAQ": kz.FPAQDecoder, "NONE": kz.NullEntropyDecoder}
seeds = [int(x) for x in os.environ.get("SEEDS", "1,2").split(",")]
cases = int(os.environ.get("CASES", "300"))
bad = 0; done = 0
for seed in seeds:
    rng = np.random.default_rng(seed)
    for case in range(cases):
        k = int(rng.integers(0, 6))
        base = int(rng.choice([0, 16384, 32768, 65536, 4 * 16384 * int(rng.integers(1, 20))]))
        n = max(0, base + int(rng.integers(-40, 41))) if k < 3 else int(rng.integers(0, 300000))
        data = (_fuzz_input(rng, n) if rng.integers(0, 2) else datagen.block(int(rng.integers(0, 40)), n)).tobytes() if n else b""
        ent = ["ANS0", "HUFFMAN", "FPAQ", "NONE"][int(rng.integers(0, 4))]
        ref, nbits = oracle.entropy_encode(ent, data)
        e = ENC[ent](ctx)
        arr = np.frombuffer(data, dtype=np.uint8).copy() if n else np.zeros(1, dtype=np.uint8)
        good = True
        try:
            e.encode(arr, 0, n); e.dispose()
            got, gbits = e.bits[-1]
        except Exception as ex:
            good = False; got, gbits = None, -1; print("encode exception", ex)
        if good: good = (gbits == nbits and got == ref[:(nbits + 7) // 8])
        if good and n:
            buf = np.zeros(n, dtype=np.uint8)
            d = DEC[ent](ctx, ref, nbits)
            good = d.decode(buf, 0, n) == n and bytes(buf) == data
            r2, _, used = oracle.entropy_decode(ent, ref, nbits, n)
            if good and hasattr(d, "bits_consumed"): good = d.bits_consumed == used
        done += 1
        if not good:
            bad += 1; print("FAIL seed", seed, "case", case, ent, "n", n, flush=True)
print("cases", done, "failures", bad)
