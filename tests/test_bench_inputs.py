"""bench.py's synthetic inputs (no GPU): the silesia.tar stand-in keeps its composition and is deterministic, the binary-member
generators sit in the compressibility range of the members they stand for."""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import datagen


def test_silesia_mix_composition_and_determinism():
    import bench
    bs = 64 * 1024
    a = bench.silesia_mix(bs)
    b = bench.silesia_mix(bs)
    assert a.shape == (51, bs) and a.dtype == np.uint8 and np.array_equal(a, b)
    assert sum(cnt for _, cnt in bench.SILESIA_MIX) == 51
    ratio = [len(zlib.compress(a[i].tobytes(), 6)) / bs for i in range(51)]
    text, exe, binary, records, redundant = ratio[:24], ratio[24:38], ratio[38:45], ratio[45:48], ratio[48:]
    assert max(text) < 0.6 and 0.25 < np.mean(exe) < 0.75 and min(binary) > 0.6 and max(records) < 0.3 and max(redundant) < 0.3


def test_binary_member_generators():
    n = 1 << 20
    for f, lo, hi in ((datagen.exe_like, 0.3, 0.6), (datagen.sensor_like, 0.65, 0.9)):
        x = f(n, 7)
        assert x.shape == (n,) and x.dtype == np.uint8 and np.array_equal(x, f(n, 7)) and not np.array_equal(x, f(n, 8))
        r = len(zlib.compress(x.tobytes(), 6)) / n
        assert lo < r < hi, (f.__name__, r)
    assert datagen.exe_like(1000, 1).shape == (1000,) and datagen.sensor_like(17, 1).shape == (17,)
