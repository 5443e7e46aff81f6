#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (kz_compress / kz_decompress): pageable host memory in,
.knz bytes in host memory out.  Not the headline metric (bench.py times device-resident buffers); reported in
DESIGN.md 5.   usage: tools/host_rate.py [blocks] [chain] [entropy] [mix|text]   (text = bench.py's text-heavy mix, for the level-exact chains)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import kanzi_amd as kz  # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    bs = 4 * 1024 * 1024
    D = min(64, nb)
    chain = sys.argv[2] if len(sys.argv) > 2 else "BWT+RANK+ZRLT"
    ent = sys.argv[3] if len(sys.argv) > 3 else "ANS0"
    if len(sys.argv) > 4 and sys.argv[4] == "text":
        import bench
        D = min(16, nb)
        host = bench.text_mix(D, bs)
    else:
        host = np.empty((D, bs), dtype=np.uint8)
        for i in range(D):
            host[i] = datagen.block(i, bs)
    data = np.ascontiguousarray(np.tile(host, ((nb + D - 1) // D, 1))[:nb]).reshape(-1)
    n = data.size
    ctx = kz.Context(0)
    tt, et = kz.transform_type(chain), kz.ENTROPY_IDS[ent.upper()]
    cap = n + n // 4 + 65536
    knz = np.empty(cap, dtype=np.uint8)
    back = np.empty(n, dtype=np.uint8)
    for rep in range(2):
        t0 = time.perf_counter()
        m = ctx.lib.kz_compress(ctx.h, tt, et, bs, data.ctypes.data, n, knz.ctypes.data, cap)       # C-ABI, plain host pointers
        ctx.check(m)
        t1 = time.perf_counter()
        r = ctx.lib.kz_decompress(ctx.h, knz.ctypes.data, m, back.ctypes.data, n)
        ctx.check(r)
        t2 = time.perf_counter()
        assert r == n and np.array_equal(back, data)
        print("rep %d: %d blocks, %.1f MB -> %.1f MB: kz_compress %.0f MB/s, kz_decompress %.0f MB/s, enc+dec %.0f MB/s (pageable host buffers, PCIe inclusive)"
              % (rep, nb, n / 1e6, m / 1e6, n / (t1 - t0) / 1e6, n / (t2 - t1) / 1e6, n / (t2 - t0) / 1e6))


if __name__ == "__main__":
    main()
