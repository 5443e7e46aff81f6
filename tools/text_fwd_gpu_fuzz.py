"""The device TEXT forward (kz_text_fwd_gpu.hip) against the oracle's encoder (needs a GPU): batches of blocks cut from generated
English / CRLF / XML / invented-word / escape-byte / UTF-8 / binary material, with mutations that aim at the forward's rules (runs of
spaces, words of two and three letters around the 16 384-word threshold, flipped first letters, words of 31 and 32 letters, bytes
>= 0x80 and escape bytes inside text, CR / LF mixes, a Magic number in front, blocks that barely shrink), go through
kz_encode_blocks with KZ_TEXT_FWD_GPU=1; block streams, bit counts, skip flags and lengths must equal oracle.encode_block's.
   python tools/text_fwd_gpu_fuzz.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ["KZ_TEXT_FWD_GPU"] = "1"
import numpy as np
import kanzi_amd as kz
import oracle, textgen, datagen

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
big4 = len(sys.argv) > 3 and sys.argv[3] == "big4"                         # the first batch is three 4 MiB blocks (BASELINE's block size), the rest as usual
small = (len(sys.argv) > 3 and sys.argv[3] == "small") or (budget <= 20 and not big4)   # many blocks of 16 - 64 KiB from the word generator: more cases per second
rng = np.random.default_rng(seed)
print("seed", seed, flush=True)
ctx = kz.Context(0)
LETTERS = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ", dtype=np.uint8)


def words(n, lo, hi, vocab):
    """n bytes of random words of lo..hi letters from a vocabulary of `vocab` words, random delimiters"""
    voc = [bytes(LETTERS[rng.integers(0, 52 if rng.random() < 0.3 else 26, int(rng.integers(lo, hi + 1)))]) for _ in range(vocab)]
    delims = [b" ", b" ", b" ", b"  ", b". ", b",", b"\n", b"\r\n", b"_", b"|", b"-", b"\t", b"'", b"0", b"\x80", b"\x0f", b"\x0e", b"(", b"]"]
    out = []; tot = 0
    while tot < n:
        w = voc[int(rng.integers(0, vocab))]
        if rng.random() < 0.15: w = bytes([w[0] ^ 0x20]) + w[1:]
        d = delims[int(rng.integers(0, len(delims)))] if rng.random() < 0.35 else b" "
        out.append(w); out.append(d); tot += len(w) + len(d)
    return b"".join(out)[:n]


def material(n):
    k = int(rng.integers(0, 12)); s = int(rng.integers(0, 1 << 30))
    if n > (1 << 19) and k in (0, 1, 2, 4, 5, 11):                           # the per-line generators take seconds per MiB: array-built text instead
        j = int(rng.integers(0, 4))
        if j == 3: return textgen.vocab_words(n, s, int(rng.integers(1000, 200000))).tobytes()
        return textgen.bulk_text(n, s, ["english", "xml", "utf8"][j]).tobytes()
    if small and k in (0, 1, 2, 4, 5, 11) and rng.random() < 0.7: k = int(rng.choice([7, 8, 9]))   # textgen's generators are slow: mostly words()
    if k == 0: return bytes(textgen.english(n, s))
    if k == 1: return bytes(textgen.english(n, s, crlf=True))
    if k == 2: return bytes(textgen.xml(n, s))
    if k == 3: return bytes(textgen.many_words(n, s, alphabet=int(rng.integers(4, 26))))
    if k == 4: return bytes(textgen.english(n, s, sprinkle=bytes([0x0F, 0x0E, 0x80, 0xFF, 0x0D])))
    if k == 5: return bytes(textgen.english(n, s, invented=int(rng.integers(10, 30000))))
    if k == 6: return bytes(textgen.utf8(n, s, bom=bool(s & 1)))
    if k == 7: return words(n, 2, 4, int(rng.integers(50, 40000)))          # two- and three-letter words, many of them: the 16 384 rule
    if k == 8: return words(n, 28, 34, int(rng.integers(20, 400)))          # around the longest word
    if k == 9: return words(n, 3, 9, int(rng.integers(2000, 60000)))        # the word list doubles
    if k == 10:
        j = int(rng.integers(0, 9))                                          # not text: detectType's verdicts (on the device: k_tf_pairs)
        if j < 5: return datagen.block(s & 0xFFFF, n, j).tobytes()
        alpha = [b"acgtn", b"0123456789+-*/=,.:; ", b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/", b"ab\n"][j - 5]
        return bytes(np.frombuffer(alpha, dtype=np.uint8)[rng.integers(0, len(alpha), n)])
    return b" " * int(rng.integers(1, 300)) + bytes(textgen.english(n, s))


t0 = time.time(); cases = bad = taken = big = 0
while time.time() - t0 < budget:
    chain, ent = [("TEXT", "NONE"), ("TEXT+UTF", "NONE"), ("TEXT+UTF+BWT+RANK+ZRLT", "ANS0"), ("TEXT", "HUFFMAN")][int(rng.integers(0, 4))]
    bs = int(rng.choice([16384, 32768, 65536])) if small else int(rng.choice([16384, 65536, 1 << 18, 1 << 20, 4 << 20]))
    ctx.set_block_size(bs)
    nblk = int(rng.integers(20, 64)) if small else (int(rng.integers(2, 12)) if bs < (1 << 20) else int(rng.integers(1, 4)))
    if big4 and cases == 0:
        bs, nblk = 4 << 20, 3
        ctx.set_block_size(bs)
    big += nblk if bs == (4 << 20) else 0
    blocks = []
    for _ in range(nblk):
        n = int(rng.integers(900, bs + 1)) if rng.random() < 0.5 else bs
        parts = []; tot = 0
        while tot < n:
            m = min(n - tot, int(rng.integers(500, n + 1)))
            parts.append(material(m)[:m]); tot += len(parts[-1])
        d = bytearray(b"".join(parts)[:n])
        for _ in range(int(rng.integers(0, 6))):                             # point mutations
            pos = int(rng.integers(0, len(d)))
            d[pos] = int(rng.choice([0x20, 0x0D, 0x0A, 0x0F, 0x80, 0x41, 0x7A, int(rng.integers(0, 256))]))
        if rng.random() < 0.05: d[:4] = b"GIF8"
        if rng.random() < 0.05: d[:2] = b"\x1f\x8b"
        blocks.append(bytes(d))
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
        taken += (sf & 0x80) == 0
        if res[i].status != 0 or (res[i].bits, res[i].skipFlags, res[i].length) != (w, sf, pl) or out[i, :(w + 7) // 8].tobytes() != so:
            bad += 1
            print("MISMATCH", chain, ent, "bs", bs, "block", i, "n", len(d), "oracle", (w, sf, pl), "hip", (res[i].status, res[i].bits, res[i].skipFlags, res[i].length), flush=True)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            open(os.path.join(ROOT, "gpurun_out", "textfwd_fail_%d_%d.bin" % (seed, cases)), "wb").write(d)
    if bad > 10: break
print("4MiB cases: %d" % big)
print("%d blocks (%d of them TEXT-coded), %d mismatches in %.0f s" % (cases, taken, bad, time.time() - t0))
sys.exit(1 if bad else 0)
