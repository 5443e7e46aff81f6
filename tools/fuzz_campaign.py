"""Long randomised differential run: HIP streams vs the oracle over many seeds, chains, sizes, options; and agreement
of the decoders on corrupted streams.  SEEDS=a,b,c CASES=n python tools/fuzz_campaign.py   (diagnostic, not a test)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
import kanzi_amd as kz, oracle, refinputs, datagen
from test_gpu_parity import _fuzz_input

ctx = kz.Context(0)
chains = ["BWT+RANK+ZRLT", "BWT+MTFT+ZRLT", "BWT+SRT+ZRLT", "BWT", "RANK", "MTFT", "ZRLT", "SRT", "LZ", "LZX", "RANK+ZRLT", "LZ+ZRLT", "NONE",
          "PACK", "DNA", "MM", "PACK+MM+LZX", "DNA+LZ", "MM+LZX", "PACK+LZ", "PACK+BWT+RANK+ZRLT", "PACK+ZRLT", "MM+PACK", "DNA+MM+LZX",
          "LZX+BWT+RANK+ZRLT", "MM+BWT+SRT+ZRLT",
          # round 3: chains led by the host stages (levels 5, 6, 3 and shorter ones); run with KZ_STREAM_CHUNK=8 KZ_HOST_CHUNK=8
          # KZ_HOST_CHUNK_DEC=8 to send every stream through the pipelines
          "TEXT+UTF+BWT+RANK+ZRLT", "TEXT+UTF+BWT+SRT+ZRLT", "TEXT+UTF+PACK+MM+LZX", "TEXT+LZ", "UTF+BWT+RANK+ZRLT", "TEXT+UTF"]
ents = ["ANS0", "HUFFMAN", "FPAQ", "NONE"]
alias = [d for _, d in refinputs.alias_inputs()]
seeds = [int(x) for x in os.environ.get("SEEDS", "1,2,3").split(",")]
cases = int(os.environ.get("CASES", "300"))
t0 = time.time(); done = 0; bad = 0
for seed in seeds:
    rng = np.random.default_rng(seed)
    for case in range(cases):
        n = int(rng.choice([0, 1, 15, 16, 17, 255, 1023, 1024, 1025, 4096, int(rng.integers(1, 70000)), int(rng.integers(1, 400000)), int(rng.integers(1, 3000000))]))
        pick = int(rng.integers(0, 4))
        if pick == 0: data = _fuzz_input(rng, n).tobytes()
        elif pick == 1: data = refinputs.multimedia_like(int(rng.integers(0, 5)), n, seed=case) if n else b""
        elif pick == 2:
            srcb = alias[int(rng.integers(0, len(alias)))]; data = (srcb * (n // len(srcb) + 1))[:n]
        else: data = datagen.block(int(rng.integers(0, 40)), n).tobytes() if n else b""
        chain, ent = chains[int(rng.integers(0, len(chains)))], ents[int(rng.integers(0, len(ents)))]
        bs = int(rng.choice([1024, 4096, 16384, 65536, 1 << 20, 4 << 20]))
        chk = int(rng.choice([0, 0, 32, 64])); skip = bool(rng.integers(0, 4) == 0)
        ref = oracle.compress(chain, ent, bs, data, jobs=4, checksum=chk, skip_blocks=skip)
        cos = kz.CompressedOutputStream(ctx, chain, ent, bs, checksum=chk, skipBlocks=skip); cos.write(data); cos.close()
        ok = cos.output == ref and kz.CompressedInputStream(ctx, ref).read(max(n, 1)) == data
        if ok and n > 200 and rng.integers(0, 3) == 0:            # decoders agree on a corrupted copy
            bad_s = bytearray(ref); k = int(rng.integers(0, 4))
            badb = refinputs.corrupt(rng, bytes(bad_s[24:]), k if k < 4 else 6)
            stream = bytes(bad_s[:24]) + badb
            try: p = ("ok", kz.CompressedInputStream(ctx, stream).read(len(data)))
            except kz.KanziError as e: p = ("err", e.code)
            try: o = ("ok", oracle.decompress(stream, len(data)))
            except oracle.OracleError as e: o = ("err", e.code)
            ok = p == o
            if not ok:
                print("DECODE MISMATCH", seed, case, chain, ent, bs, chk, skip, n, p[0], p[1] if p[0] == "err" else len(p[1]), o[0], o[1] if o[0] == "err" else len(o[1]), flush=True)
                os.makedirs(os.path.join(ROOT, "gpurun_out", "fuzz_fail"), exist_ok=True)      # the stream the decoders disagree on, for a replay
                open(os.path.join(ROOT, "gpurun_out", "fuzz_fail", "stream_%d_%d.knz" % (seed, case)), "wb").write(stream)
        if not ok:
            bad += 1
            print("FAIL seed", seed, "case", case, chain, ent, "bs", bs, "chk", chk, "skip", skip, "n", n, "pick", pick, flush=True)
        done += 1
print("cases", done, "failures", bad, "in %.0f s" % (time.time() - t0))
