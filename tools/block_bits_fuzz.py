"""kz_decode_blocks against the oracle's decode_block on block streams whose HEADER or bit length is off: bit flips in the first six
bytes (mode byte, skip flags, the post-transform length, the checksum), bit lengths W - 1, W - 8, W + 8, W / 2, small blocks (copy
blocks of <= 15 bytes, raw entropy tails below 32 symbols) next to chunk-sized ones; every chain family and entropy coder.
   python tools/block_bits_fuzz.py [seconds] [seed]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kanzi_amd as kz
import oracle, datagen

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
ctx = kz.Context(0)
print("seed", seed, flush=True)
CHAINS = [("BWT+RANK+ZRLT", "ANS0"), ("BWT+SRT+ZRLT", "FPAQ"), ("LZ", "HUFFMAN"), ("LZ", "ANS0"), ("LZX", "NONE"), ("ZRLT", "HUFFMAN"), ("RANK+ZRLT", "ANS0"),
          ("NONE", "HUFFMAN"), ("NONE", "ANS0"), ("NONE", "FPAQ"), ("BWT", "NONE"), ("PACK+LZ", "HUFFMAN"), ("MM+LZX", "ANS0")]
t0 = time.time(); cases = bad = 0
while time.time() - t0 < budget:
    chain, ent = CHAINS[int(rng.integers(0, len(CHAINS)))]
    bs = int(rng.choice([1024, 4096, 65536]))
    B = int(rng.integers(4, 24))
    lens = np.array([int(rng.choice([rng.integers(1, 16), rng.integers(16, 40), rng.integers(40, 400), rng.integers(400, bs + 1), bs])) for _ in range(B)], np.int32)
    inp = np.zeros((B, bs), np.uint8)
    for b in range(B):
        if lens[b]: inp[b, :lens[b]] = datagen.block(int(rng.integers(0, 1 << 16)), int(lens[b]), int(rng.integers(0, 5)))
    ostride = kz.max_block_stream_bytes(bs)
    out = np.zeros((B, ostride), np.uint8)
    res = kz.encode_blocks(ctx, chain, ent, inp, bs, lens, out, ostride)
    bits = np.array([r.bits for r in res], np.int64)
    bad_streams = out.copy(); bits2 = bits.copy()
    for b in range(B):
        k = int(rng.integers(0, 6))
        nby = int((bits[b] + 7) // 8)
        if nby == 0: continue
        if k == 0:                                   # header bit flips
            for _ in range(int(rng.integers(1, 3))):
                pos = int(rng.integers(0, min(nby, 6)))
                bad_streams[b, pos] ^= np.uint8(1 << int(rng.integers(0, 8)))
        elif k == 1: bits2[b] = max(0, bits[b] - 1)
        elif k == 2: bits2[b] = max(0, bits[b] - 8)
        elif k == 3: bits2[b] = bits[b] + 8
        elif k == 4: bits2[b] = bits[b] // 2
    dec = np.zeros((B, bs), np.uint8)
    r2 = kz.decode_blocks(ctx, chain, ent, bs, bad_streams, ostride, bits2, dec, bs)
    for b in range(B):
        nby = int((bits2[b] + 7) // 8)
        ro, oo = oracle.decode_block(chain, ent, bs, bytes(bad_streams[b, :nby]), int(bits2[b]), bs)
        ok = (r2[b].status == 0 and r2[b].length == ro and bytes(dec[b, :ro]) == oo) if ro >= 0 else (r2[b].status == ro)
        cases += 1
        if not ok:
            bad += 1
            if bad <= 6: np.savez(os.path.join(ROOT, "gpurun_out", "blockbits_fail_%d_%d.npz" % (seed, bad)), stream=bad_streams[b, :nby + 8], bits=bits2[b], wbits=bits[b], chain=chain, ent=ent, bs=bs, n=lens[b])
            print("MISMATCH", chain, ent, "bs", bs, "n", int(lens[b]), "bits", int(bits2[b]), "of", int(bits[b]), "oracle", ro, "hip", r2[b].status, r2[b].length, flush=True)
            if bad > 30: break
    if bad > 30: break
print("%d cases, %d mismatches in %.0f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
