import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, kanzi_amd as kz, oracle, datagen, collections
ctx = kz.Context(0)
stats = collections.Counter()
for chain, ent in [("BWT+RANK+ZRLT","ANS0"),("LZ","HUFFMAN"),("BWT+SRT+ZRLT","FPAQ"),("LZX","NONE")]:
    rng = np.random.default_rng(77)
    data = datagen.stream(3, 32768*6).tobytes()
    good = oracle.compress(chain, ent, 32768, data, jobs=2, checksum=64)
    for trial in range(60):
        bad = bytearray(good); kind = trial % 4; hdr = 24
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))):
                pos = int(rng.integers(hdr, len(bad))); bad[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1: bad = bad[:int(rng.integers(hdr, len(bad)))]
        elif kind == 2:
            a = int(rng.integers(hdr, len(bad) - 64)); bad[a:a + 64] = bytes(rng.integers(0, 256, 64, dtype=np.uint8))
        else:
            a = int(rng.integers(hdr, len(bad) - 8)); del bad[a:a + int(rng.integers(1, 8))]
        try: p = ("ok", len(kz.CompressedInputStream(ctx, bytes(bad)).read(len(data))))
        except kz.KanziError as e: p = ("err", e.code)
        try: o = ("ok", len(oracle.decompress(bytes(bad), len(data))))
        except oracle.OracleError as e: o = ("err", e.code)
        if p != o: print('MISMATCH', chain, ent, 'trial', trial, 'kind', kind, p, o)
        stats[(kind, p == o, p if p[0]=="err" else "ok", o if o[0]=="err" else "ok")] += 1
for k, v in sorted(stats.items(), key=str): print(k, v)
