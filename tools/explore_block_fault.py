"""Find which stage disagrees (product vs oracle) on corrupted blocks.  Diagnostic only."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, kanzi_amd as kz, oracle, datagen

ctx = kz.Context(0)
CODECS = {"BWT": kz.BWTBlockCodec, "ZRLT": kz.ZRLT, "SRT": kz.SRT}

def mk(name):
    if name == "RANK": return kz.SBRT(ctx, 2)
    if name == "MTFT": return kz.SBRT(ctx, 1)
    if name == "LZ": return kz.LZCodec(ctx, kz.LZ_TYPE)
    if name == "LZX": return kz.LZCodec(ctx, kz.LZX_TYPE)
    return CODECS[name](ctx)

DEC = {"ANS0": kz.ANSRangeDecoder, "HUFFMAN": kz.HuffmanDecoder, "FPAQ": kz.FPAQDecoder, "NONE": kz.NullEntropyDecoder}

def stage_walk(chain, ent, bsz, blk, nbits, chk):
    names = chain.split("+")
    mode = blk[0]
    pos = 1
    if mode & 0x80: return "copy block"
    skip = ((mode << 4) | 0x0F) & 0xFF
    if mode & 0x10: skip = blk[pos]; pos += 1
    ds = 1 + ((mode >> 5) & 3)
    pre = int.from_bytes(blk[pos:pos + ds], "big"); pos += ds
    pos += 1 + (8 if chk == 64 else 4 if chk == 32 else 0)
    payload = blk[pos:]
    pbits = nbits - 8 * pos
    # entropy
    r_, eo, _ = oracle.entropy_decode(ent, payload, pbits, pre)
    if r_ != pre: eo = None
    buf = np.zeros(pre + 64, dtype=np.uint8)
    try:
        d = DEC[ent](ctx, payload, pbits); r = d.decode(buf, 0, pre); ep = bytes(buf[:pre]) if r == pre else None
    except kz.KanziError: ep = None
    if (eo is None) != (ep is None) or (eo is not None and bytes(eo) != ep):
        return "entropy: oracle %s product %s pre=%d" % (eo is not None, ep is not None, pre)
    if eo is None: return "both entropy fail"
    cur = bytes(eo)
    for i in reversed(range(len(names))):
        if skip & (0x80 >> i): continue
        cap = bsz + max(512, bsz >> 4)
        ok_, o = oracle.transform_inverse(names[i], cur, cap)
        if not ok_: o = None
        src = kz.SliceByteArray(np.frombuffer(cur, dtype=np.uint8).copy(), len(cur), 0)
        dst = kz.SliceByteArray(np.zeros(cap, dtype=np.uint8), cap, 0)
        try: ok = mk(names[i]).inverse(src, dst); p = bytes(dst.array[:dst.index]) if ok else None
        except kz.KanziError: p = None
        if (o is None) != (p is None) or (o is not None and bytes(o) != p):
            return "%s inverse: oracle %s product %s inlen=%d" % (names[i], "ok(%d)" % len(o) if o is not None else "fail", "ok(%d)" % len(p) if p is not None else "fail", len(cur))
        if o is None: return "both fail at " + names[i]
        cur = bytes(o)
    return "all stages agree"

for chain, ent in [("BWT+RANK+ZRLT", "ANS0"), ("LZ", "HUFFMAN"), ("BWT+SRT+ZRLT", "FPAQ"), ("LZX", "NONE")]:
    rng = np.random.default_rng(77)
    data = datagen.stream(3, 32768 * 6).tobytes()
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
        if p == o: continue
        print("MISMATCH", chain, ent, trial, kind, p, o)
        try: idx = kz.knz_index(bytes(bad))
        except kz.KanziError as e: print("  index fails", e.code); continue
        for off, nb in idx["blocks"]:
            blk = kz.extract_bits(bytes(bad), off, nb)
            print("   block@%d: %s" % (off, stage_walk(chain, ent, 32768, bytes(blk), nb, 64)))
