"""Per synthetic class (4 MiB block): groups of equal d-byte prefixes, live suffixes, and the share of suffixes in groups above 4096 / 8192 / 16384
(what a radix sort of the first d bytes cannot separate: the reason round 0 of the suffix sort counts before it moves)."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import datagen
n = 4 << 20
for cls in range(5):
    x = datagen.block(cls, n, cls)
    xp = np.concatenate([x, np.zeros(16, np.uint8)]).astype(np.uint64)
    key = np.zeros(n, np.uint64)
    print("class", cls)
    for d in range(1, 9):
        key = (key << np.uint64(8)) | xp[d-1:d-1+n]
        u, c = np.unique(key, return_counts=True)
        row = [f"d={d} groups={len(u):8d} live={(c[c>1].sum())/n:5.3f}"]
        for cap in (4096, 8192, 16384):
            big = c[c > cap]
            row.append(f">{cap}: {big.sum()/n:5.3f} ({len(big)})")
        print("  ", "  ".join(row))
