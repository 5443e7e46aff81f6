#!/usr/bin/env python3
"""Host rate of the TEXT / UTF inverse stages per block and core (no GPU: kz_host_stage_*): what a GPU form would have to beat.
   python tools/text_inverse_rate.py [blocks]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import textgen  # noqa: E402
import kanzi_amd as kz  # noqa: E402

lib = kz.load_library()
bs = 4 << 20
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
TEXT, ANS0 = kz.TEXT_TYPE, kz.ENTROPY_IDS["ANS0"]
tot_in = tot_out = 0
t_inv = 0.0
for i in range(nb):
    src = np.ascontiguousarray(textgen.bulk_text(bs, 1000 + i, "english"))
    dst = np.empty(bs + 1024, dtype=np.uint8)
    dt = ctypes.c_int32(0)
    prod = ctypes.c_int32(0)
    rc = lib.kz_host_stage_forward(TEXT, ANS0, bs, ctypes.addressof(dt), src.ctypes.data, bs, dst.ctypes.data, dst.size, ctypes.addressof(prod))
    assert rc == 1, rc
    enc = np.ascontiguousarray(dst[:prod.value])
    back = np.empty(bs, dtype=np.uint8)
    p2 = ctypes.c_int32(0)
    t0 = time.perf_counter()
    rc = lib.kz_host_stage_inverse(TEXT, bs, enc.ctypes.data, enc.size, back.ctypes.data, bs, ctypes.addressof(p2))
    t_inv += time.perf_counter() - t0
    assert rc == 1 and p2.value == bs and np.array_equal(back, src)
    tot_in += enc.size
    tot_out += bs
    words = int(np.count_nonzero(src == 32))
print("TEXT inverse, one thread: %.2f ms per 4 MiB block (%.0f MB/s of output), coded/plain %.3f, ~%d words per block -> %.1f ns per word"
      % (t_inv / nb * 1e3, tot_out / t_inv / 1e6, tot_in / tot_out, words, t_inv / nb / max(words, 1) * 1e9))
