"""The device UTF forward (kz_utf_fwd_gpu.hip) against the oracle's encoder (needs a GPU): batches of UTF-8 blocks -- code points of
two and three units from varying alphabets (a few up to 20 000 distinct), some with four-unit code points (host stage), a byte order
mark, blocks cut inside a code point at either end -- with mutations that keep the byte-pair statistics of UTF-8 (so that the device
TEXT forward still declines the block as UTF8 and UTFCodec skips its validation) but break the walk: a continuation byte inserted or
dropped, the third unit of a three-unit code point replaced by a letter, a first unit of two units in front of a letter.  Chain
TEXT+UTF (entropy NONE / ANS0 with the level-5 transforms): block streams, bit counts, skip flags and lengths must equal the oracle's.
   python tools/utf_fwd_gpu_fuzz.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ["KZ_TEXT_FWD_GPU"] = "1"
os.environ["KZ_TEXT_GPU_TRACE"] = "1"
import numpy as np
import kanzi_amd as kz
import oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
print("seed", seed, flush=True)
ctx = kz.Context(0)


def utf8_block(n):
    kind = int(rng.integers(0, 6))
    pools = [list(range(0x400, 0x460)), list(range(0x370, 0x400)), list(range(0x4E00, 0x4E00 + int(rng.integers(50, 20000)))), list(range(0x3040, 0x30FF)),
             list(range(0x80, 0x800, 7)), list(range(0x800, 0xFFFF, int(rng.integers(3, 400))))]
    cps = pools[kind] + list(range(0x20, 0x7F)) * int(rng.integers(0, 4)) + [0x0A]
    if rng.random() < 0.15: cps += [0x1F600, 0x1F601, 0x10000]                       # four units: the host stage's
    cps = [c for c in cps if not (0xD800 <= c <= 0xDFFF)]
    w = rng.random(len(cps)) ** float(rng.choice([1.0, 3.0, 8.0]))                  # flat or skewed use
    idx = rng.choice(len(cps), size=n // 2 + 8, p=w / w.sum())
    b = bytearray("".join(chr(cps[i]) for i in idx).encode("utf-8"))
    if rng.random() < 0.2: b = bytearray(b"\xef\xbb\xbf") + b
    cut = int(rng.integers(0, 4)) if rng.random() < 0.4 else 0
    b = b[cut:cut + n]
    for _ in range(int(rng.integers(0, 3)) if rng.random() < 0.5 else 0):           # mutations that detectType's pair rules do not see
        pos = int(rng.integers(8, len(b) - 8))
        m = int(rng.integers(0, 4))
        if m == 0: b.insert(pos, 0x80 + int(rng.integers(0, 64)))
        elif m == 1:
            while pos < len(b) - 4 and (b[pos] & 0xC0) != 0x80: pos += 1
            del b[pos]
        elif m == 2:
            while pos < len(b) - 4 and not (0xE0 <= b[pos] < 0xF0): pos += 1
            if pos < len(b) - 4: b[pos + 2] = 0x41
        else:
            while pos < len(b) - 4 and not (0xC2 <= b[pos] < 0xE0): pos += 1
            if pos < len(b) - 4: b[pos + 1] = 0x80 + int(rng.integers(0, 64))      # (still a continuation byte: another code point)
    return bytes(b[:n])


t0 = time.time(); cases = bad = applied = 0
while time.time() - t0 < budget:
    chain, ent = [("TEXT+UTF", "NONE"), ("TEXT+UTF", "HUFFMAN"), ("TEXT+UTF+BWT+RANK+ZRLT", "ANS0")][int(rng.integers(0, 3))]
    bs = int(rng.choice([16384, 65536, 1 << 18, 1 << 20]))
    ctx.set_block_size(bs)
    nblk = int(rng.integers(4, 24)) if bs < (1 << 20) else int(rng.integers(1, 5))
    blocks = [utf8_block(int(rng.integers(1100, bs + 1)) if rng.random() < 0.4 else bs) for _ in range(nblk)]
    B = len(blocks)
    inp = np.zeros((B, bs), dtype=np.uint8)
    lens = np.array([len(d) for d in blocks], dtype=np.int32)
    for i, d in enumerate(blocks):
        inp[i, :len(d)] = np.frombuffer(d, dtype=np.uint8)
    ostride = kz.max_block_stream_bytes(bs)
    out = np.zeros((B, ostride), dtype=np.uint8)
    res = kz.encode_blocks(ctx, chain, ent, inp, bs, lens, out, ostride)
    for i, d in enumerate(blocks):
        so, w, sf, pl = oracle.encode_block(chain, ent, d, block_size=bs)
        cases += 1
        applied += (sf & 0x40) == 0
        if res[i].status != 0 or (res[i].bits, res[i].skipFlags, res[i].length) != (w, sf, pl) or out[i, :(w + 7) // 8].tobytes() != so:
            bad += 1
            print("MISMATCH", chain, ent, "bs", bs, "block", i, "n", len(d), "oracle", (w, sf, pl), "hip", (res[i].status, res[i].bits, res[i].skipFlags, res[i].length), flush=True)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            open(os.path.join(ROOT, "gpurun_out", "utffwd_fail_%d_%d.bin" % (seed, cases)), "wb").write(d)
    if bad > 10: break
print("%d blocks (UTF applied to %d of them), %d mismatches in %.0f s" % (cases, applied, bad, time.time() - t0))
sys.exit(1 if bad else 0)
