"""GPU parity tests (-m gpu): every HIP stage and the fused block/stream paths, called through the
C-ABI, compared bit-for-bit with the CPU oracle on the same inputs; committed fixtures; and
size-independent properties at the full 4 MiB block size."""
import ctypes
import json
import os

import numpy as np
import pytest

import kanzi_amd as kz
import datagen
import oracle
import refinputs
import textgen

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _fwd(ctx, tid, data, cap=None):
    cap = ctx.lib.kz_transform_max_encoded_len(tid, len(data)) if cap is None else cap
    out = np.zeros(cap + 64, dtype=np.uint8)
    p = ctypes.c_int32(0)
    a = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
    rc = ctx.lib.kz_transform_forward(ctx.h, tid, a.ctypes.data, len(data), out.ctypes.data, cap, ctypes.addressof(p))
    assert rc >= 0, ctx.error()
    return rc == 1, out[:p.value].tobytes()


def _inv(ctx, tid, data, cap):
    out = np.zeros(cap + 64, dtype=np.uint8)
    p = ctypes.c_int32(0)
    a = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
    rc = ctx.lib.kz_transform_inverse(ctx.h, tid, a.ctypes.data, len(data), out.ctypes.data, cap, ctypes.addressof(p))
    assert rc >= 0, ctx.error()
    return rc == 1, out[:p.value].tobytes()


def _all_inputs():
    ins = refinputs.transform_inputs()[:20] + refinputs.bwt_inputs() + refinputs.entropy_inputs() + refinputs.edge_inputs()
    for c in range(5):
        ins.append(datagen.block(c, 50000).tobytes())
    ins.append(bytes((np.arange(20000) & 0xFF).astype(np.uint8)))        # T/test/TestBWT ramp, scaled down
    return ins


TIDS = {"BWT": kz.BWT_TYPE, "RANK": kz.RANK_TYPE, "MTFT": kz.MTFT_TYPE, "ZRLT": kz.ZRLT_TYPE,
        "SRT": kz.SRT_TYPE, "LZ": kz.LZ_TYPE, "LZX": kz.LZX_TYPE}


@pytest.mark.parametrize("name", ["BWT", "RANK", "MTFT", "ZRLT", "SRT", "LZ", "LZX"])
def test_transform_forward_and_inverse_match_oracle(ctx, name):
    for data in _all_inputs():
        ok_o, enc_o = oracle.transform_forward(name, data)
        ok_g, enc_g = _fwd(ctx, TIDS[name], data)
        assert ok_o == ok_g, (name, len(data))
        if not ok_o:
            continue
        assert enc_g == enc_o, (name, len(data))
        ok_i, back = _inv(ctx, TIDS[name], enc_o, len(data) + 1024)
        assert ok_i and back == data, (name, len(data))


def _group_shape_inputs():
    """Inputs that steer the later suffix-sort rounds of the forward BWT through each of their paths (kz_bwt_fwd.hip):
    groups of a few suffixes (counted in LDS), groups of several hundred to several thousand (sorted in LDS), groups
    larger than a bucket (LSD window), and mixes of them in one block."""
    rng = np.random.default_rng(77)
    out = []
    def repeats(n, nwords, wlen):
        words = rng.integers(0, 256, (nwords, wlen), dtype=np.uint8)
        idx = rng.integers(0, nwords, n // wlen + 1)
        return words[idx].reshape(-1)[:n].tobytes()
    out.append(repeats(300000, 900, 16))          # ~20 copies per word: groups of tens
    out.append(repeats(700001, 64, 24))           # ~450 copies: groups of hundreds
    out.append(repeats(1 << 20, 12, 40))          # ~2000 copies: groups of thousands, buckets beyond the count path
    out.append(repeats(900000, 3, 33))            # ~9000 copies: groups larger than a bucket
    out.append((b"abcdefghijklmnopqrstuvwxy" * 40000)[:999983])            # one period: n/25 per group until the end decides
    z = bytearray(1 << 20)
    for p_ in rng.integers(0, 1 << 20, 3000):
        z[p_] = int(rng.integers(1, 256))
    out.append(bytes(z))                           # long zero runs: one group over most of the block for many rounds
    out.append(repeats(200000, 900, 16) + bytes(z[:300000]) + repeats(400000, 12, 40) + bytes(rng.integers(0, 256, 100000, dtype=np.uint8)))
    out.append(repeats(4099, 5, 7))                # just above the list length where the bucket path starts
    return out


@pytest.mark.gpu
def test_zrlt_forward_seams_match_oracle(ctx):
    """The forward ZRLT works on rows of 64 bytes, waves of 1 KiB and tiles of 4 KiB: zero runs of every length class across those
    seams, dense escapes (the transform declines), all-zero blocks, ragged batches (tools/zrlt_fuzz.py is the long form)."""
    rng = np.random.default_rng(20260929)
    runs = [1, 2, 3, 7, 8, 62, 63, 64, 65, 127, 128, 1023, 1024, 1025, 4095, 4096, 4097, 8192, 70000]

    def block(n, kind):
        if kind == 0:
            out, left = [], n
            while left > 0:
                L = min(int(rng.choice(runs)), left)
                out.append(np.zeros(L, np.uint8)); left -= L
                m = min(int(rng.integers(1, 5)), left)
                lit = rng.integers(1, 256, m, dtype=np.uint8)
                lit[rng.random(m) < 0.3] = 0xFF
                out.append(lit); left -= m
            return np.concatenate(out)[:n]
        if kind == 1:
            x = np.zeros(n, np.uint8)
            x[rng.integers(0, n, max(1, n // 700))] = 0xFE
            return x
        if kind == 2:
            return rng.choice(np.array([0xFE, 0xFF, 1, 0], np.uint8), n, p=[0.4, 0.4, 0.1, 0.1])
        x = np.zeros(n, np.uint8)
        if kind == 3:
            x[-1] = 7
        return x

    sizes = [63, 64, 65, 1023, 1024, 1025, 4095, 4096, 4097, 8192, 12289, 65536, 200001, 1 << 20]
    blocks = [np.ascontiguousarray(block(n, k)) for n in sizes for k in range(5)]
    for b0 in range(0, len(blocks), 7):
        bl = blocks[b0:b0 + 7]
        bs = max(len(b) for b in bl)
        inp = np.zeros((len(bl), bs), np.uint8)
        lens = np.array([len(b) for b in bl], np.int32)
        for i, b in enumerate(bl):
            inp[i, :len(b)] = b
        ostride = kz.max_block_stream_bytes(bs)
        out = np.zeros((len(bl), ostride), np.uint8)
        res = kz.encode_blocks(ctx, "ZRLT", "NONE", inp, bs, lens, out, ostride)
        bits = np.array([r.bits for r in res], np.int64)
        dec = np.zeros((len(bl), bs), np.uint8)
        res2 = kz.decode_blocks(ctx, "ZRLT", "NONE", bs, out, ostride, bits, dec, bs)
        for i, b in enumerate(bl):
            so, w, sf, pl = oracle.encode_block("ZRLT", "NONE", b)
            assert res[i].status == 0 and (res[i].bits, res[i].skipFlags, res[i].length) == (w, sf, pl), (len(b), b0 + i)
            assert out[i, :(w + 7) // 8].tobytes() == so, (len(b), b0 + i)
            assert res2[i].status == 0 and dec[i, :len(b)].tobytes() == b.tobytes()


@pytest.mark.gpu
def test_zrlt_inverse_output_that_fits_exactly(ctx):
    """ZRLT.java:166-233 leaves its loop the moment the output is full and then reports whether the input was used up: with a
    lone escape (or a run that wraps to nothing) left behind the token that filled the buffer, an output that fits EXACTLY is an
    error, one byte more of room makes it a success (found by tools/zrlt_inv_fuzz.py; tools/tightcap_fuzz.py does this for every
    inverse transform)."""
    cases = [bytes([5, 7, 0xFF]), bytes([9, 0, 1, 3, 0xFF]), bytes([2] * 70 + [0xFF]), bytes([0xFF, 1, 0, 0, 0, 8, 0xFF]),
             bytes([3, 0xFF, 0xFF, 0xFF]), bytes([7] * 4100 + [0xFF])]
    for data in cases:
        ok_big, full = oracle.transform_inverse("ZRLT", data, 1 << 16)
        assert ok_big
        for cap in (len(full) - 1, len(full), len(full) + 1):
            if cap < 1:
                continue
            ok_o, o = oracle.transform_inverse("ZRLT", data, cap)
            src = kz.SliceByteArray(np.frombuffer(data, dtype=np.uint8).copy(), len(data), 0)
            dst = kz.SliceByteArray(np.zeros(cap, dtype=np.uint8), cap, 0)
            ok_p = kz.ZRLT(ctx).inverse(src, dst)
            assert bool(ok_p) == bool(ok_o), (data[:8], cap)
            if ok_o:
                assert bytes(dst.array[:dst.index]) == o
        assert not oracle.transform_inverse("ZRLT", data, len(full))[0] and oracle.transform_inverse("ZRLT", data, len(full) + 1)[0]


@pytest.mark.gpu
def test_bwt_forward_group_shapes_match_oracle(ctx, monkeypatch):
    """BWT is unique: every path of the suffix sort must give the oracle's bytes and primary indexes.
    The same inputs are run with the bucket path switched off (KZ_BWT_BUCKETS=0: LSD passes only), batched."""
    inputs = _group_shape_inputs()
    want = [oracle.transform_forward("BWT", d) for d in inputs]
    for d, (ok_o, enc_o) in zip(inputs, want):
        ok_g, enc_g = _fwd(ctx, kz.BWT_TYPE, d)
        assert ok_g == ok_o and enc_g == enc_o, len(d)
    # one batch with blocks of different lengths: the bucket geometry follows the longest block
    bs = max(len(d) for d in inputs)
    B = len(inputs)
    inp = np.zeros((B, bs), dtype=np.uint8)
    lens = np.array([len(d) for d in inputs], dtype=np.int32)
    for i, d in enumerate(inputs):
        inp[i, :len(d)] = np.frombuffer(d, dtype=np.uint8)
    ostride = kz.max_block_stream_bytes(bs)
    out = np.zeros((B, ostride), dtype=np.uint8)
    res = kz.encode_blocks(ctx, "BWT", "NONE", inp, bs, lens, out, ostride)
    for i, d in enumerate(inputs):
        so, w, sf, pl = oracle.encode_block("BWT", "NONE", d)
        assert res[i].status == 0 and (res[i].bits, res[i].skipFlags, res[i].length) == (w, sf, pl), len(d)
        assert out[i, :(w + 7) // 8].tobytes() == so, len(d)
    monkeypatch.setenv("KZ_BWT_BUCKETS", "0")
    for d, (ok_o, enc_o) in zip(inputs[:4], want[:4]):
        ok_g, enc_g = _fwd(ctx, kz.BWT_TYPE, d)
        assert ok_g == ok_o and enc_g == enc_o, len(d)


def test_python_mirror_slice_semantics(ctx):
    # ByteTransform contract (K/ByteTransform.java:36-56): indexes advance on success, False = declined
    src = kz.SliceByteArray(np.frombuffer(b"mississippi", dtype=np.uint8).copy(), 11, 0)
    dst = kz.SliceByteArray(np.zeros(64, dtype=np.uint8), 64, 0)
    codec = kz.BWTBlockCodec(ctx)
    assert codec.getMaxEncodedLength(11) == 44
    assert codec.forward(src, dst) is True
    assert src.index == 11 and dst.index == 13
    assert dst.array[:13].tobytes() == bytes([0, 4]) + b"ipssmpissii"      # K/transform/BWT.java:45-50
    z = kz.ZRLT(ctx)
    s2 = kz.SliceByteArray(np.array([0, 0, 0, 5, 0xFE, 0], dtype=np.uint8), 6, 0)
    d2 = kz.SliceByteArray(np.zeros(16, dtype=np.uint8), 16, 0)
    assert z.forward(s2, d2) is False and s2.index == 0 and d2.index == 0   # ZRLT.java:94 -> skipped
    r = kz.SBRT(ctx, kz.SBRT.MODE_MTF)
    s3 = kz.SliceByteArray(np.frombuffer(b"abca", dtype=np.uint8).copy(), 4, 0)
    d3 = kz.SliceByteArray(np.zeros(4, dtype=np.uint8), 4, 0)
    assert r.forward(s3, d3) and list(d3.array) == [97, 98, 99, 2]


def _huffman_limit_input():
    # Fibonacci-like frequencies force code lengths > 12 -> limitCodeLengths (HuffmanEncoder.java:191-273)
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    d = b"".join(bytes([i]) * f for i, f in enumerate(fib))[:16384]
    return bytes(np.random.default_rng(0).permutation(np.frombuffer(d, dtype=np.uint8)))


@pytest.mark.parametrize("ent", ["ANS0", "HUFFMAN", "FPAQ"])
def test_entropy_encode_decode_match_oracle(ctx, ent):
    Enc, Dec = {"ANS0": (kz.ANSRangeEncoder, kz.ANSRangeDecoder), "HUFFMAN": (kz.HuffmanEncoder, kz.HuffmanDecoder),
                "FPAQ": (kz.FPAQEncoder, kz.FPAQDecoder)}[ent]
    for data in _all_inputs() + [_huffman_limit_input(), _huffman_limit_input()[:5000] + bytes(range(256)) * 4]:
        if len(data) == 0:
            continue
        bits_o, nb_o = oracle.entropy_encode(ent, data)
        enc = Enc(ctx)
        assert enc.encode(np.frombuffer(data, dtype=np.uint8), 0, len(data)) == len(data)
        bits_g, nb_g = enc.bits[0]
        assert nb_g == nb_o and bits_g == bits_o, len(data)
        dec = Dec(ctx, bits_o, nb_o)
        out = np.zeros(len(data), dtype=np.uint8)
        assert dec.decode(out, 0, len(data)) == len(data)
        assert out.tobytes() == data
        assert dec.bits_consumed == nb_o, (ent, len(data))          # EntropyDecoder contract: exactly the encoder's bits


@pytest.mark.parametrize("chain,ent", [("BWT+RANK+ZRLT", "ANS0"), ("BWT+MTFT+ZRLT", "ANS0"), ("ZRLT", "NONE"),
                                        ("BWT", "ANS0"), ("RANK+ZRLT", "ANS0"), ("NONE", "ANS0"),
                                        ("BWT+RANK+ZRLT", "HUFFMAN"), ("NONE", "HUFFMAN"),
                                        ("BWT+SRT+ZRLT", "FPAQ"), ("LZ", "HUFFMAN"), ("LZ", "ANS0"), ("LZX", "NONE")])
def test_block_streams_match_oracle(ctx, chain, ent):
    """kz_encode_blocks output == oracle encode_block (header, skip flags, raw fallback, copy blocks)."""
    bs = 40000
    blocks = [datagen.block(c, bs).tobytes() for c in range(5)]
    blocks += [b"0123456789abcde", b"0123456789abcdef", bytes(bs), b"ab" * 300, bytes(np.random.default_rng(1).integers(0, 256, 777, dtype=np.uint8))]
    B = len(blocks)
    inp = np.zeros((B, bs), dtype=np.uint8)
    lens = np.zeros(B, dtype=np.int32)
    for i, b in enumerate(blocks):
        inp[i, :len(b)] = np.frombuffer(b, dtype=np.uint8)
        lens[i] = len(b)
    ostride = kz.max_block_stream_bytes(bs)
    out = np.zeros((B, ostride), dtype=np.uint8)
    res = kz.encode_blocks(ctx, chain, ent, inp, bs, lens, out, ostride)
    bits = []
    for i, b in enumerate(blocks):
        s, w, sf, pl = oracle.encode_block(chain, ent, b)
        assert res[i].status == 0
        assert (res[i].bits, res[i].skipFlags, res[i].length) == (w, sf, pl), (i, len(b))
        assert out[i, :(w + 7) // 8].tobytes() == s, (i, len(b))
        bits.append(w)
    dec = np.zeros((B, bs), dtype=np.uint8)
    res2 = kz.decode_blocks(ctx, chain, ent, bs, out, ostride, np.array(bits, dtype=np.int64), dec, bs)
    for i, b in enumerate(blocks):
        assert res2[i].status == 0 and res2[i].length == len(b)
        assert dec[i, :len(b)].tobytes() == b


def test_knz_stream_matches_oracle_and_goldens(ctx):
    with open(os.path.join(GOLD, "manifest.json")) as f:
        man = json.load(f)
    for e in man["entries"]:
        inp = open(os.path.join(GOLD, e["input"]), "rb").read()
        exp = open(os.path.join(GOLD, e["output"]), "rb").read()
        cos = kz.CompressedOutputStream(ctx, e["chain"], e["entropy"], e["blockSize"])
        cos.write(inp)
        cos.close()
        assert cos.output == exp, e
        assert kz.CompressedInputStream(ctx, exp).read() == inp


def test_corrupt_block_header_is_rejected(ctx):
    # T/test/TestCompressedStream.java:177-291 (tamper tests): block header checksum / length
    data = datagen.block(0, 30000).tobytes()
    knz = bytearray(oracle.compress("BWT+RANK+ZRLT", "ANS0", 16384, data, jobs=1))
    idx = kz.knz_index(bytes(knz))
    off = idx["blocks"][0][0]
    knz[off // 8 + 1] ^= 0x40                      # inside the first block's header
    with pytest.raises(kz.KanziError):
        kz.CompressedInputStream(ctx, bytes(knz)).read(len(data))


@pytest.mark.parametrize("chain,ent", [("BWT+RANK+ZRLT", "ANS0"), ("LZ", "ANS0"), ("BWT+SRT+ZRLT", "FPAQ"), ("LZ", "HUFFMAN")])
def test_full_size_blocks_properties(ctx, chain, ent):
    """Every BASELINE config at its own size (4 MiB blocks): encode -> decode identity for a batch of six blocks (one of each
    synthetic class, the uniform random one ends as a raw "transformed copy"), and the HIP block streams equal the oracle's on
    two of them (the oracle needs seconds per 4 MiB block)."""
    bs = 4 * 1024 * 1024
    B = 6
    inp = np.stack([datagen.block(i, bs) for i in range(B)])
    lens = np.full(B, bs, dtype=np.int32)
    ostride = kz.max_block_stream_bytes(bs)
    out = np.zeros((B, ostride), dtype=np.uint8)
    res = kz.encode_blocks(ctx, chain, ent, inp, bs, lens, out, ostride)
    assert all(r.status == 0 for r in res)
    assert res[3].mode & 0x80                                             # uniform random block: nothing compresses it
    bits = np.array([r.bits for r in res], dtype=np.int64)
    dec = np.zeros((B, bs), dtype=np.uint8)
    res2 = kz.decode_blocks(ctx, chain, ent, bs, out, ostride, bits, dec, bs)
    assert all(r.status == 0 and r.length == bs for r in res2)
    assert np.array_equal(dec, inp)
    for i in (0, 4):
        s, w, sf, pl = oracle.encode_block(chain, ent, inp[i])
        assert (res[i].bits, res[i].skipFlags, res[i].length) == (w, sf, pl)
        assert out[i, :(w + 7) // 8].tobytes() == s


@pytest.mark.parametrize("bits", [32, 64])
def test_block_checksums_match_oracle(ctx, bits):
    """-x32 / -x64: XXHash32 (standard) and the reference's XXHash64 variant of the original block in the block
    header (CompressedOutputStream.java:749-755,887-891); corrupted payload must be reported as ERR_CRC_CHECK or
    a decode error (CompressedInputStream.java:1349-1363)."""
    data = datagen.stream(5, 32768).tobytes() + b"0123456789abcdefXYZ" + bytes(37)
    ref = oracle.compress("BWT+RANK+ZRLT", "ANS0", 32768, data, jobs=2, checksum=bits)
    cos = kz.CompressedOutputStream(ctx, "BWT+RANK+ZRLT", "ANS0", 32768, checksum=bits)
    cos.write(data)
    cos.close()
    assert cos.output == ref
    assert kz.CompressedInputStream(ctx, ref).read() == data
    raw = oracle.compress("NONE", "NONE", 32768, data, jobs=1, checksum=bits)     # payload = plain bytes: flip one
    bad = bytearray(raw)
    bad[len(bad) // 2] ^= 0x01
    with pytest.raises(kz.KanziError) as e:
        kz.CompressedInputStream(ctx, bytes(bad)).read(len(data))
    assert e.value.code == 19


# ---- mirrors of the reference's own stream tests (T = java/src/test/java/io/github/flanglet/kanzi/test) ----
def _mix32(c, h, v):
    c ^= (h * (~v & 0xFFFFFFFF)) & 0xFFFFFFFF
    c = ((c << 13) | (c >> 19)) & 0xFFFFFFFF
    return (c * 5 + 0x52DCE729) & 0xFFFFFFFF


def _block_header_checksum(mode, skip_flags, length, encoded_block_length):      # T/TestCompressedStream.java:488-498
    h = 0x1E35A7BD
    c = (h * 0x01030507) & 0xFFFFFFFF
    for v in (mode, skip_flags, length, encoded_block_length >> 32, encoded_block_length & 0xFFFFFFFF):
        c = _mix32(c, h, v)
    return ((c >> 23) ^ (c >> 3)) & 0xFF


def _write_bits(buf, off, value, count):
    for k in range(count):
        bit = (value >> (count - 1 - k)) & 1
        byte, sh = (off + k) >> 3, 7 - ((off + k) & 7)
        buf[byte] = (buf[byte] & ~(1 << sh)) | (bit << sh)


def _small_copy_stream(ctx):
    cos = kz.CompressedOutputStream(ctx, "NONE", "NONE", 1024)
    cos.write(bytes([1, 2, 3, 4, 5, 6, 7, 8]))
    cos.close()
    enc = cos.output
    assert enc == oracle.compress("NONE", "NONE", 1024, bytes([1, 2, 3, 4, 5, 6, 7, 8]), jobs=1)
    idx = kz.knz_index(enc)
    boff, w = idx["blocks"][0]
    lr = 3 if w < 8 else (w >> 3).bit_length() - 1 + 4
    mode = kz.extract_bits(enc, boff, 8)[0]
    assert mode & 0x80                                               # small blocks are copied
    data_size = 1 + ((mode >> 5) & 3)
    pre = int.from_bytes(kz.extract_bits(enc, boff + 8, 8 * data_size), "big")
    return bytearray(enc), boff, w, lr, mode, data_size, pre


def test_ref_bulk_read_end_of_stream(ctx):
    """T/TestCompressedStream.java:98-120: 3 bytes, NONE/NONE, block size 1024."""
    cos = kz.CompressedOutputStream(ctx, "NONE", "NONE", 1024)
    cos.write(bytes([1, 2, 3]))
    cos.close()
    assert cos.output == oracle.compress("NONE", "NONE", 1024, bytes([1, 2, 3]), jobs=1)
    assert kz.CompressedInputStream(ctx, cos.output).read() == bytes([1, 2, 3])


def test_ref_block_header_checksum_checked_before_payload_read(ctx):
    """T/TestCompressedStream.java:177-229: flip one bit of the block header checksum and cut the stream right
    behind it: the decoder must report ERR_CRC_CHECK (19), not a short read."""
    enc, boff, w, lr, mode, data_size, pre = _small_copy_stream(ctx)
    ck_off = boff + 8 + 8 * data_size
    enc[ck_off >> 3] ^= 1 << (7 - (ck_off & 7))
    cut = bytes(enc[:(ck_off + 15) >> 3])
    with pytest.raises(kz.KanziError) as e:
        kz.CompressedInputStream(ctx, cut).read(16)
    assert e.value.code == 19


def test_ref_encoded_block_length_bound_checked_before_payload_read(ctx):
    """T/TestCompressedStream.java:231-291: encoded block length + 8 with a matching header checksum, stream cut
    behind the header: ERR_BLOCK_SIZE (2)."""
    enc, boff, w, lr, mode, data_size, pre = _small_copy_stream(ctx)
    assert w + 8 < (1 << lr)
    _write_bits(enc, boff - lr, w + 8, lr)
    ck_off = boff + 8 + 8 * data_size
    _write_bits(enc, ck_off, _block_header_checksum(mode, 0, pre, w + 8), 8)
    cut = bytes(enc[:(ck_off + 15) >> 3])
    with pytest.raises(kz.KanziError) as e:
        kz.CompressedInputStream(ctx, cut).read(16)
    assert e.value.code == 2


def test_ref_correctness_matrix(ctx):
    """T/TestCompressedStream.java:57-96 testCorrectness: sizes 65536 << (test % 7), values from
    java.util.Random(Long.MAX_VALUE).nextInt(4*test+1); compress1 = NONE & HUFFMAN, compress2 = LZX & FPAQ with a
    random checksum kind; block size (length / (1 + nextInt(3))) & -16.  Every stream must equal the oracle's
    (where the oracle is fast enough) and decode back to the input."""
    rng = oracle.JavaRandom((1 << 63) - 1)
    for test in (1, 2, 3, 6, 7, 13):
        length = 65536 << (test % 7)
        values = bytes(rng.next_int(4 * test + 1) for _ in range(min(length, 1 << 18)))
        values = (values * (length // len(values) + 1))[:length]
        for chain, ent, chk in (("NONE", "HUFFMAN", 0), ("LZX", "FPAQ", rng.next_int(3) * 32)):
            bs = (length // (1 + rng.next_int(3))) & -16
            cos = kz.CompressedOutputStream(ctx, chain, ent, bs, checksum=chk)
            cos.write(values)
            cos.close()
            if length <= (1 << 20):
                assert cos.output == oracle.compress(chain, ent, bs, values, jobs=4, checksum=chk), (test, chain, bs, chk)
            assert kz.CompressedInputStream(ctx, cos.output).read() == values, (test, chain, bs, chk)


def _fuzz_input(rng, n):
    kind = rng.integers(0, 8)
    if kind == 0:
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == 1:
        return rng.integers(0, int(rng.integers(1, 5)), n, dtype=np.uint8)                 # tiny alphabet
    if kind == 2:
        return np.repeat(rng.integers(0, 256, max(1, n // 37 + 1), dtype=np.uint8), 37)[:n]   # long runs
    if kind == 3:
        period = int(rng.integers(1, 300))
        return np.resize(rng.integers(0, 256, period, dtype=np.uint8), n)                  # periodic: deep suffix sort rounds
    if kind == 4:
        a = rng.integers(0, 256, n, dtype=np.uint8)
        a[rng.random(n) < 0.9] = 0                                                         # sparse
        return a
    if kind == 5:
        return np.zeros(n, dtype=np.uint8)
    if kind == 6:
        words = [bytes(rng.integers(97, 123, int(rng.integers(2, 9)), dtype=np.uint8)) for _ in range(40)]
        out = b" ".join(words[int(i)] for i in rng.integers(0, 40, n // 4 + 1))
        return np.frombuffer(out[:n].ljust(n, b"."), dtype=np.uint8)
    return (np.arange(n) * int(rng.integers(1, 7)) % 251).astype(np.uint8)                 # ramps


def test_fuzz_streams_match_oracle(ctx):
    """Randomised parity: lengths, contents, chains, entropy coders, block sizes and checksum kinds drawn from a fixed
    seed; every HIP stream must equal the oracle's and decode back (through both decoders)."""
    rng = np.random.default_rng(20260928)
    chains = ["BWT+RANK+ZRLT", "BWT+MTFT+ZRLT", "BWT+SRT+ZRLT", "BWT", "RANK", "ZRLT", "SRT", "LZ", "LZX", "RANK+ZRLT", "BWT+RANK", "LZ+ZRLT", "NONE"]
    ents = ["ANS0", "HUFFMAN", "FPAQ", "NONE"]
    for case in range(160):
        n = int(rng.choice([0, 1, 15, 16, 17, 255, 1023, 1024, 4096, int(rng.integers(1, 70000)), int(rng.integers(1, 200000))]))
        data = _fuzz_input(rng, n).tobytes()
        chain, ent = chains[int(rng.integers(0, len(chains)))], ents[int(rng.integers(0, len(ents)))]
        bs = int(rng.choice([1024, 4096, 16384, 65536, 1 << 20]))
        chk = int(rng.choice([0, 0, 32, 64]))
        ref = oracle.compress(chain, ent, bs, data, jobs=4, checksum=chk)
        cos = kz.CompressedOutputStream(ctx, chain, ent, bs, checksum=chk)
        cos.write(data)
        cos.close()
        assert cos.output == ref, (case, n, chain, ent, bs, chk)
        assert kz.CompressedInputStream(ctx, ref).read(max(n, 1)) == data, (case, n, chain, ent, bs, chk)
        assert oracle.decompress(cos.output, n, jobs=2) == data


def _codec(ctx, name):
    if name == "RANK":
        return kz.SBRT(ctx, 2)
    if name == "MTFT":
        return kz.SBRT(ctx, 1)
    if name == "LZ":
        return kz.LZCodec(ctx, kz.LZ_TYPE)
    if name == "LZX":
        return kz.LZCodec(ctx, kz.LZX_TYPE)
    if name in ("PACK", "DNA"):
        return kz.AliasCodec(ctx, onlyDNA=(name == "DNA"))
    return {"BWT": kz.BWTBlockCodec, "ZRLT": kz.ZRLT, "SRT": kz.SRT, "MM": kz.FSDCodec}[name](ctx)


def _mm_inputs():
    """(label, bytes, context dataType before the call)"""
    rng = np.random.default_rng(11)
    cases = []
    for kind in range(5):
        for n in (1024, 1500, 5000, 65536, 300001):
            cases.append(("mm%d/%d" % (kind, n), refinputs.multimedia_like(kind, n, seed=kind + n), 0))
    cases.append(("random", bytes(rng.integers(0, 256, 50000, dtype=np.uint8)), 0))
    cases.append(("text", (b"the quick brown fox jumps over the lazy dog " * 2000)[:60000], 0))
    cases.append(("dna", bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 80000)]), 0))
    cases.append(("digits", bytes(np.frombuffer(b"0123456789,.", dtype=np.uint8)[rng.integers(0, 12, 30000)]), 0))
    cases.append(("base64", bytes(np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/", dtype=np.uint8)[rng.integers(0, 64, 30000)]), 0))
    cases.append(("two symbols", bytes(np.frombuffer(b"ab", dtype=np.uint8)[rng.integers(0, 2, 30000)]), 0))
    cases.append(("short", refinputs.multimedia_like(0, 1023), 0))
    cases.append(("png magic", b"\x89PNG" + refinputs.multimedia_like(0, 60000)[4:], 0))
    cases.append(("riff magic", b"RIFF" + refinputs.multimedia_like(0, 60000)[4:], kz.DATA_TYPES["MULTIMEDIA"]))
    cases.append(("bmp magic", b"BM" + refinputs.multimedia_like(1, 60000)[2:], kz.DATA_TYPES["MULTIMEDIA"]))
    cases.append(("pgm magic", b"P5\n" + refinputs.multimedia_like(2, 60000)[3:], 0))
    cases.append(("tagged exe", refinputs.multimedia_like(0, 60000), kz.DATA_TYPES["EXE"]))
    cases.append(("tagged bin", refinputs.multimedia_like(0, 60000), kz.DATA_TYPES["BIN"]))
    cases.append(("escapes everywhere", bytes(rng.integers(0, 2, 40000, dtype=np.uint8) * 255), 0))
    return cases


def test_mm_forward_inverse_and_data_type_match_oracle(ctx):
    """FSDCodec (transform MM): verdict, bytes and the "dataType" context entry it leaves behind, forward and inverse,
    against the oracle (FSDCodec.java:60-318; Global.detectSimpleType :556-605; Magic.getType)."""
    for label, data, dt0 in _mm_inputs():
        ok_o, out_o, dt_o = oracle.transform_forward("MM", data, data_type=dt0)
        ctx.set_data_type(dt0)
        codec = kz.FSDCodec(ctx)
        cap = codec.getMaxEncodedLength(len(data))
        assert cap == len(data) + max(64, len(data) >> 4)
        src = kz.SliceByteArray(np.frombuffer(data, dtype=np.uint8).copy(), len(data), 0)
        dst = kz.SliceByteArray(np.zeros(cap, dtype=np.uint8), cap, 0)
        ok_p = codec.forward(src, dst)
        assert bool(ok_p) == bool(ok_o), label
        assert ctx.get_data_type() == dt_o, label
        if ok_o:
            assert bytes(dst.array[:dst.index]) == out_o, label
            back = kz.SliceByteArray(np.zeros(len(data) + 64, dtype=np.uint8), len(data) + 64, 0)
            assert kz.FSDCodec(ctx).inverse(kz.SliceByteArray(np.frombuffer(out_o, dtype=np.uint8).copy(), len(out_o), 0), back)
            assert bytes(back.array[:back.index]) == data, label
    ctx.set_data_type(0)


@pytest.mark.parametrize("name", ["PACK", "DNA"])
def test_alias_codec_matches_oracle(ctx, name):
    """AliasCodec (PACK / DNA): verdict, bytes and the "dataType" entry, forward and inverse, on every branch (one
    symbol, 2-bit and 4-bit packing with every count remainder, digram aliases with and without a trailing byte, declined
    inputs) and under the context tags that rule it out (AliasCodec.java:76-470)."""
    cases = [(label, data, 0) for label, data in refinputs.alias_inputs()]
    text = dict(refinputs.alias_inputs())["text+0"]
    for tag in ("MULTIMEDIA", "UTF8", "EXE", "BIN", "TEXT", "DNA"):
        cases.append(("tagged " + tag, text[:60000], kz.DATA_TYPES[tag]))
    applied = 0
    for label, data, dt0 in cases:
        ok_o, out_o, dt_o = oracle.transform_forward(name, data, data_type=dt0)
        ctx.set_data_type(dt0)
        codec = _codec(ctx, name)
        cap = codec.getMaxEncodedLength(len(data))
        assert cap == len(data) + 1024
        dst = kz.SliceByteArray(np.zeros(cap, dtype=np.uint8), cap, 0)
        ok_p = codec.forward(kz.SliceByteArray(np.frombuffer(data, dtype=np.uint8).copy(), len(data), 0), dst)
        assert bool(ok_p) == bool(ok_o), (name, label)
        assert ctx.get_data_type() == dt_o, (name, label)
        if ok_o:
            applied += 1
            assert bytes(dst.array[:dst.index]) == out_o, (name, label)
            back = kz.SliceByteArray(np.zeros(len(data) + 64, dtype=np.uint8), len(data) + 64, 0)
            assert _codec(ctx, name).inverse(kz.SliceByteArray(np.frombuffer(out_o, dtype=np.uint8).copy(), len(out_o), 0), back)
            assert bytes(back.array[:back.index]) == data, (name, label)
    assert applied >= (4 if name == "DNA" else 12)
    ctx.set_data_type(0)


@pytest.mark.parametrize("chain,ent", [("PACK+LZX", "HUFFMAN"), ("DNA+LZ", "HUFFMAN"), ("PACK+MM+LZX", "HUFFMAN"), ("PACK+BWT+RANK+ZRLT", "ANS0")])
def test_alias_streams_match_oracle(ctx, chain, ent):
    """Streams with PACK / DNA in front (DNA+LZ & HUFFMAN is the reference's level 2, PACK+MM+LZX & HUFFMAN the tail
    of level 3): the data type PACK detects changes what LZ does with the same block (DNA -> minMatch 6, SMALL_ALPHABET
    -> LZ steps aside), and all of it must equal the oracle's stream."""
    inputs = dict(refinputs.alias_inputs())
    bs = 65536
    parts = [inputs["acgt+0"][:bs], inputs["text+0"][:bs], inputs["two symbols+1"][:30001] + inputs["digits+0"][:bs - 30001],
             refinputs.multimedia_like(1, bs), inputs["random"][:40000] + inputs["one symbol"][:5000] + inputs["5 symbols"][:bs - 45000],
             inputs["markov+0"][:bs], inputs["16 symbols+1"][:20001]]
    data = b"".join(parts)
    ref = oracle.compress(chain, ent, bs, data, jobs=2, checksum=32)
    cos = kz.CompressedOutputStream(ctx, chain, ent, bs, checksum=32)
    cos.write(data)
    cos.close()
    assert cos.output == ref, (chain, ent)
    assert kz.CompressedInputStream(ctx, ref).read(len(data)) == data


def test_lz_honours_the_data_type_entry(ctx):
    """LZCodec.forward reads the context's dataType: DNA raises minMatch to 6, SMALL_ALPHABET declines
    (LZCodec.java:343-352)."""
    rng = np.random.default_rng(4)
    words = [bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(rng.integers(5, 40)))]) for _ in range(300)]
    data = b"".join(words[int(i)] for i in rng.integers(0, 300, 6000))[:120000]
    for name in ("LZ", "LZX"):
        for dt in ("UNDEFINED", "DNA", "SMALL_ALPHABET", "MULTIMEDIA"):
            ok_o, out_o, _ = oracle.transform_forward(name, data, data_type=oracle.DT[dt])
            ctx.set_data_type(dt)
            codec = _codec(ctx, name)
            cap = codec.getMaxEncodedLength(len(data))
            dst = kz.SliceByteArray(np.zeros(cap, dtype=np.uint8), cap, 0)
            ok_p = codec.forward(kz.SliceByteArray(np.frombuffer(data, dtype=np.uint8).copy(), len(data), 0), dst)
            assert bool(ok_p) == bool(ok_o), (name, dt)
            if ok_o:
                assert bytes(dst.array[:dst.index]) == out_o, (name, dt)
                assert (out_o[12] >> 1) & 7 == (4 if dt == "DNA" else 2)
    ctx.set_data_type(0)


@pytest.mark.parametrize("chain,ent", [("MM+LZX", "HUFFMAN"), ("MM", "ANS0"), ("MM+BWT+RANK+ZRLT", "ANS0")])
def test_mm_streams_match_oracle(ctx, chain, ent):
    """Whole streams with MM in the chain (MM+LZX & HUFFMAN is the tail of the reference's level 3): blocks where MM
    applies, declines, and is ruled out by the writer's Magic tag; bit-identical to the oracle's stream."""
    rng = np.random.default_rng(8)
    bs = 65536
    parts = [refinputs.multimedia_like(0, bs), refinputs.multimedia_like(3, bs), bytes(rng.integers(0, 256, bs, dtype=np.uint8)),
             (b"plain text block " * 5000)[:bs], b"\x7FELF" + refinputs.multimedia_like(1, bs)[4:], b"RIFF" + refinputs.multimedia_like(4, bs)[4:],
             b"PK\x03\x04" + refinputs.multimedia_like(2, bs)[4:], refinputs.multimedia_like(1, 30001)]
    data = b"".join(parts)
    for chk in (0, 32):
        ref = oracle.compress(chain, ent, bs, data, jobs=2, checksum=chk)
        cos = kz.CompressedOutputStream(ctx, chain, ent, bs, checksum=chk)
        cos.write(data)
        cos.close()
        assert cos.output == ref, (chain, ent, chk)
        assert kz.CompressedInputStream(ctx, ref).read(len(data)) == data


@pytest.mark.parametrize("name", ["BWT", "SRT", "ZRLT", "RANK", "MTFT", "LZ", "LZX", "MM", "PACK"])
def test_inverse_transforms_follow_the_reference_on_corrupted_input(ctx, name):
    """Malformed input to an inverse transform: the verdict (applied / failed) AND, when it applies, every output byte
    must be what the reference's code path yields (the oracle restates it, Java int wrap-around included): BWT's 8
    literal walkers (BWT.java:295-368), SRT's stale rank table (SRT.java:204-250), ZRLT's wrapped run lengths
    (ZRLT.java:166-228), LZ's bounded reads (LZCodec.java:626-747)."""
    rng = np.random.default_rng(123)
    n = 20000
    cap = n + max(512, n >> 4)
    compared = 0
    for src_kind in range(8):
        pre = datagen.block(src_kind, n).tobytes() if name != "MM" else refinputs.multimedia_like(src_kind % 5, n, seed=src_kind)
        if name == "PACK":
            pre = refinputs.alias_inputs()[(0, 1, 5, 9, 11, 13, 15, 17)[src_kind]][1][:n]
        if name in ("SRT", "RANK", "MTFT", "ZRLT"):
            pre = oracle.transform_forward("BWT", pre)[1]
            if name == "ZRLT":
                pre = oracle.transform_forward("RANK", pre)[1]
        ok, good = oracle.transform_forward(name, pre)
        if not ok:
            continue
        for trial in range(24):
            bad = refinputs.corrupt(rng, good, trial % 8)
            ok_o, o = oracle.transform_inverse(name, bad, cap)
            src = kz.SliceByteArray(np.frombuffer(bad, dtype=np.uint8).copy(), len(bad), 0)
            dst = kz.SliceByteArray(np.zeros(cap, dtype=np.uint8), cap, 0)
            ok_p = _codec(ctx, name).inverse(src, dst)
            assert bool(ok_p) == bool(ok_o), (name, src_kind, trial)
            if ok_o:
                assert bytes(dst.array[:dst.index]) == o, (name, src_kind, trial)
            compared += 1
    assert compared >= 72


@pytest.mark.parametrize("ent", ["ANS0", "HUFFMAN", "FPAQ"])
def test_entropy_decoders_follow_the_reference_on_corrupted_input(ctx, ent):
    """Malformed entropy payloads: same verdict and same bytes as the reference's decoder, including its quirk of
    returning `count` after a chunk whose size does not add up (ANSRangeDecoder.java:229-231; the unwritten tail is
    zero on both sides) and FPAQ's over-read test (FPAQDecoder.java:231-232)."""
    dec = {"ANS0": kz.ANSRangeDecoder, "HUFFMAN": kz.HuffmanDecoder, "FPAQ": kz.FPAQDecoder}[ent]
    rng = np.random.default_rng(321)
    for src_kind in (3, 1, 6, 0):
        data = datagen.block(src_kind, 40000).tobytes()
        good, nbits = oracle.entropy_encode(ent, data)
        for trial in range(32):
            bad = refinputs.corrupt(rng, good, trial % 8)
            nb = min(nbits, len(bad) * 8)
            r, o, _ = oracle.entropy_decode(ent, bad, nb, len(data))
            buf = np.zeros(len(data), dtype=np.uint8)
            ok_p = dec(ctx, bad, nb).decode(buf, 0, len(data)) == len(data)
            assert ok_p == (r == len(data)), (ent, src_kind, trial)
            if ok_p:
                assert bytes(buf) == o, (ent, src_kind, trial)


@pytest.mark.gpu
@pytest.mark.parametrize("ent", ["ANS0", "HUFFMAN", "FPAQ", "NONE"])
def test_entropy_decoders_with_wrong_count_or_cut_bits(ctx, ent):
    """A damaged block header asks the decoder for a symbol count the payload was not written for, or the payload is shorter than
    the decoder needs: the raw tails (Huffman chunks below 32 symbols, ANS blocks up to 32) are bulk reads that throw past the
    block's bits (HuffmanDecoder.java:364-366, ANSRangeDecoder.java:193-196), and a zeroed tail sends the Exp-Golomb length
    decoder through Java's int shift and byte cast (ExpGolombDecoder.java:41-58).  Verdict and bytes are the oracle's
    (tools/entropy_count_fuzz.py is the long form: 234 000 cases)."""
    dec = {"ANS0": kz.ANSRangeDecoder, "HUFFMAN": kz.HuffmanDecoder, "FPAQ": kz.FPAQDecoder, "NONE": kz.NullEntropyDecoder}[ent]
    rng = np.random.default_rng(9)
    compared = 0
    for n in (27, 33, 180, 16384, 16386, 32770):
        for kind in (0, 1):
            data = datagen.block(n + kind, n, kind).tobytes()
            good, nbits = oracle.entropy_encode(ent, data)
            streams = [good]
            if len(good) > 16:
                z = bytearray(good); z[len(z) // 8:] = bytes(len(z) - len(z) // 8); streams.append(bytes(z))     # zeroed tail
                streams.append(bytes(refinputs.corrupt(rng, good, 0)))
            for st in streams:
                nbmax = min(nbits, len(st) * 8)
                for count in sorted(set(max(1, c) for c in (n - 1, n, n + 1, 32, 33, n + 16384))):
                    for nb in sorted(set(max(0, b) for b in (nbmax, nbmax - 1, nbmax - 8, nbmax // 2))):
                        r, o, _ = oracle.entropy_decode(ent, st, nb, count)
                        buf = np.zeros(count, dtype=np.uint8)
                        ok_p = dec(ctx, st, nb).decode(buf, 0, count) == count
                        assert ok_p == (r == count), (ent, n, kind, count, nb, nbits)
                        if ok_p:
                            assert bytes(buf) == o, (ent, n, kind, count, nb)
                        compared += 1
    assert compared > 400


@pytest.mark.parametrize("chain,ent", [("BWT+RANK+ZRLT", "ANS0"), ("BWT+SRT+ZRLT", "FPAQ"), ("PACK+MM+LZX", "HUFFMAN")])
def test_batched_decode_isolates_corrupted_blocks(ctx, chain, ent):
    """kz_decode_blocks on a batch where a few blocks are corrupted: every block gets the status (or the bytes) the
    oracle gives that block alone, and the clean neighbours decode as if nothing happened."""
    rng = np.random.default_rng(17)
    bs, B = 65536, 64
    inp = np.zeros((B, bs), dtype=np.uint8)
    for b in range(B):
        inp[b] = datagen.block(b, bs) if b % 3 else np.frombuffer(refinputs.multimedia_like(b % 5, bs, seed=b), dtype=np.uint8)
    ostride = kz.max_block_stream_bytes(bs)
    out = np.zeros((B, ostride), dtype=np.uint8)
    res = kz.encode_blocks(ctx, chain, ent, inp, bs, np.full(B, bs, dtype=np.int32), out, ostride)
    bits = np.array([r.bits for r in res], dtype=np.int64)
    for rnd in range(3):
        bad = out.copy()
        hit = set(int(x) for x in rng.integers(0, B, 8))
        for b in hit:
            nby = int((bits[b] + 7) // 8)
            row = (refinputs.corrupt(rng, bytes(bad[b, :nby]), int(rng.integers(0, 8))) + bytes(nby))[:nby]
            bad[b, :nby] = np.frombuffer(row, dtype=np.uint8)
        dec = np.zeros((B, bs), dtype=np.uint8)
        r2 = kz.decode_blocks(ctx, chain, ent, bs, bad, ostride, bits, dec, bs)
        for b in range(B):
            nby = int((bits[b] + 7) // 8)
            ro, oo = oracle.decode_block(chain, ent, bs, bytes(bad[b, :nby]), int(bits[b]), bs)
            if ro >= 0:
                assert r2[b].status == 0 and r2[b].length == ro and bytes(dec[b, :ro]) == oo, (chain, rnd, b, b in hit)
            else:
                assert r2[b].status == ro, (chain, rnd, b, b in hit, ro, r2[b].status)
            if b not in hit:
                assert ro == bs and bytes(dec[b]) == bytes(inp[b])


def test_huffman_oversubscribed_code_lengths_fail_cleanly(ctx):
    """A Huffman header whose code lengths over-subscribe the code space (found by tools/explore_stage_garbage.py; the
    fixture is that corrupted payload): the reference dies on the decoding-table index (HuffmanDecoder.java:183-186) and
    the block fails; the decoder here must report failure too -- it once wrote past its table instead."""
    bad = open(os.path.join(os.path.dirname(__file__), "golden", "huffman_oversubscribed_lengths.bin"), "rb").read()
    r, _, _ = oracle.entropy_decode("HUFFMAN", bad, 190574, 40000)
    assert r != 40000
    buf = np.zeros(40000, dtype=np.uint8)
    assert kz.HuffmanDecoder(ctx, bad, 190574).decode(buf, 0, 40000) != 40000
    data = datagen.block(0, 40000).tobytes()                                   # and the context still works
    good, nbits = oracle.entropy_encode("HUFFMAN", data)
    assert kz.HuffmanDecoder(ctx, good, nbits).decode(buf, 0, 40000) == 40000 and bytes(buf) == data


def test_stream_header_faults_report_reference_codes(ctx):
    """Stream-header faults surface with the code and in the order of CompressedInputStream.readHeader
    (CompressedInputStream.java:363-478, Error.java:24-43), the same as the oracle reports."""
    data = datagen.stream(5, 20000).tobytes()
    good = oracle.compress("BWT+RANK+ZRLT", "ANS0", 4096, data, checksum=32)
    assert kz.CompressedInputStream(ctx, good).read(len(data)) == data
    for what, bad, code in refinputs.header_faults(good):
        with pytest.raises(kz.KanziError) as e:
            kz.CompressedInputStream(ctx, bad).read(len(data))
        assert e.value.code == code, what
        with pytest.raises(oracle.OracleError) as eo:
            oracle.decompress(bad, len(data))
        assert eo.value.code == code, what
    assert kz.CompressedInputStream(ctx, good).read(len(data)) == data


def test_skip_blocks_option_matches_oracle(ctx):
    """The writer's "skipBlocks" option (CompressedOutputStream.java:769-788): blocks that start with the magic number
    of a compressed format, or whose order-0 entropy reaches 0.95 * 8 bits, become copy blocks -- large ones included;
    the others go through the chain as usual.  Stream identical to the oracle's, and smaller work for the decoder."""
    rng = np.random.default_rng(31)
    bs = 65536
    text = (b"blocks that compress are left alone by the option. " * 2000)[:bs]
    nearly = bytes(rng.integers(0, 200, bs, dtype=np.uint8))                    # 7.64 bits: just above the threshold
    below = bytes(rng.integers(0, 150, bs, dtype=np.uint8))                     # 7.23 bits: below it
    parts = [text, bytes(rng.integers(0, 256, bs, dtype=np.uint8)), b"\x1F\x8B\x08\x00" + text[4:], nearly, below,
             b"PK\x03\x04" + bytes(rng.integers(0, 4, bs - 4, dtype=np.uint8)), refinputs.multimedia_like(0, bs), text[:7000]]
    data = b"".join(parts)
    for chain, ent in (("BWT+RANK+ZRLT", "ANS0"), ("PACK+MM+LZX", "HUFFMAN"), ("LZ", "FPAQ")):
        for chk in (0, 64):
            ref = oracle.compress(chain, ent, bs, data, jobs=2, checksum=chk, skip_blocks=True)
            plain = oracle.compress(chain, ent, bs, data, jobs=2, checksum=chk)
            assert ref != plain
            cos = kz.CompressedOutputStream(ctx, chain, ent, bs, checksum=chk, skipBlocks=True)
            cos.write(data)
            cos.close()
            assert cos.output == ref, (chain, ent, chk)
            assert kz.CompressedInputStream(ctx, ref).read(len(data)) == data
            assert oracle.decompress(ref, len(data)) == data
    z = oracle.compress("BWT+RANK+ZRLT", "ANS0", bs, data, skip_blocks=True)
    modes = [kz.extract_bits(z, off, 8)[0] for off, nb in kz.knz_index(z)["blocks"]]
    copied = [i for i, m in enumerate(modes) if (m & 0x80) and not (m & 0x10)]   # copy block, not a "transformed copy"
    assert copied == [1, 2, 3, 5, 6]                # random, gzip magic, 7.64 bits, zip magic, 16-bit samples (noisy low bytes)


def test_fuzz_streams_with_pack_dna_mm(ctx):
    """The same randomised parity for the chains that carry the "dataType" entry from stage to stage: PACK / DNA / MM in
    front of LZ, LZX, BWT chains; inputs also drawn from the multimedia-like and alphabet-limited generators."""
    rng = np.random.default_rng(20260929)
    chains = ["PACK", "DNA", "MM", "PACK+MM+LZX", "DNA+LZ", "MM+LZX", "PACK+LZ", "PACK+BWT+RANK+ZRLT", "PACK+ZRLT", "MM+PACK", "DNA+MM+LZX"]
    ents = ["ANS0", "HUFFMAN", "FPAQ", "NONE"]
    alias = [d for _, d in refinputs.alias_inputs()]
    for case in range(120):
        n = int(rng.choice([0, 1, 16, 1023, 1024, 1025, 4096, int(rng.integers(1, 70000)), int(rng.integers(1, 200000))]))
        pick = int(rng.integers(0, 3))
        if pick == 0:
            data = _fuzz_input(rng, n).tobytes()
        elif pick == 1:
            data = refinputs.multimedia_like(int(rng.integers(0, 5)), n, seed=case) if n else b""
        else:
            src = alias[int(rng.integers(0, len(alias)))]
            data = (src * (n // len(src) + 1))[:n]
        chain, ent = chains[int(rng.integers(0, len(chains)))], ents[int(rng.integers(0, len(ents)))]
        bs = int(rng.choice([1024, 4096, 16384, 65536, 1 << 20]))
        chk = int(rng.choice([0, 0, 32, 64]))
        ref = oracle.compress(chain, ent, bs, data, jobs=4, checksum=chk)
        cos = kz.CompressedOutputStream(ctx, chain, ent, bs, checksum=chk)
        cos.write(data)
        cos.close()
        assert cos.output == ref, (case, n, chain, ent, bs, chk)
        assert kz.CompressedInputStream(ctx, ref).read(max(n, 1)) == data, (case, n, chain, ent, bs, chk)


@pytest.mark.parametrize("chain,ent", [("BWT+RANK+ZRLT", "ANS0"), ("LZ", "HUFFMAN"), ("BWT+SRT+ZRLT", "FPAQ"), ("LZX", "NONE")])
def test_corrupted_streams_never_hang_or_crash(ctx, chain, ent):
    """Bit flips, truncations and garbage payloads: the decoder must return (an error code or some bytes) -- no hang,
    no fault -- and a context must stay usable afterwards.  Block checksums (-x64) must flag every payload
    corruption that still decodes."""
    rng = np.random.default_rng(77)
    data = datagen.stream(3, 32768).tobytes()
    good = oracle.compress(chain, ent, 32768, data, jobs=2, checksum=64)
    hdr = 24                                                   # leave the stream header alone: those checks run on the host
    for trial in range(40):
        bad = bytearray(good)
        kind = trial % 4
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))):
                pos = int(rng.integers(hdr, len(bad)))
                bad[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            bad = bad[:int(rng.integers(hdr, len(bad)))]
        elif kind == 2:
            a = int(rng.integers(hdr, len(bad) - 64))
            bad[a:a + 64] = bytes(rng.integers(0, 256, 64, dtype=np.uint8))
        else:
            a = int(rng.integers(hdr, len(bad) - 8))
            del bad[a:a + int(rng.integers(1, 8))]
        try:
            out = kz.CompressedInputStream(ctx, bytes(bad)).read(len(data))
            assert out == data or len(out) <= len(data)       # undetected only if nothing that matters changed
            if bytes(bad) != good:
                assert out == data or len(out) < len(data), "payload corruption slipped through the block checksum"
            got = ("ok", out)
        except kz.KanziError as e:
            assert e.code > 0
            got = ("err", e.code)
        try:                                                   # and the same outcome as the reference's reader
            want = ("ok", oracle.decompress(bytes(bad), len(data)))
        except oracle.OracleError as e:
            want = ("err", e.code)
        assert got == want, (chain, ent, trial)
    assert kz.CompressedInputStream(ctx, good).read(len(data)) == data


def test_large_blocks_match_oracle(ctx):
    """Blocks above the 4 MiB default: 8 MiB blocks (the reference's -l 6 default, BlockCompressor.java:143-145) and a
    ragged 12 MiB block, against the oracle."""
    rng = np.random.default_rng(5)
    words = [bytes(rng.integers(97, 123, int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(500)]
    text = b" ".join(words[int(i)] for i in rng.integers(0, 500, 2_600_000))
    data = text[:8 * 1024 * 1024 + 12345] + bytes(rng.integers(0, 256, 300000, dtype=np.uint8)) + bytes(500000)
    for bs, chain, ent in ((8 * 1024 * 1024, "BWT+RANK+ZRLT", "ANS0"), (12 * 1024 * 1024, "BWT+SRT+ZRLT", "HUFFMAN")):
        ref = oracle.compress(chain, ent, bs, data, jobs=2)
        cos = kz.CompressedOutputStream(ctx, chain, ent, bs)
        cos.write(data)
        cos.close()
        assert cos.output == ref, (bs, chain)
        assert kz.CompressedInputStream(ctx, ref).read() == data


def test_blocks_above_16_mib_match_oracle(ctx):
    """Blocks of 2^24 bytes and more (round 5; the reference goes to 1 GiB and switches its inverse BWT to inverseBiPSIv2 above
    8 MiB, BWT.java:231-234,384-544, with the same output): 8-byte links in the inverse BWT, the plain by-position list in the
    RANK / MTFT inverse.  A 32 MiB block + ragged tail through the level-5 core chain, a 20 MiB block size through MTFT."""
    rng = np.random.default_rng(6)
    words = [bytes(rng.integers(97, 123, int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(800)]
    text = b" ".join(words[int(i)] for i in rng.integers(0, 800, 5_000_000))
    data = text[:30 * 1024 * 1024] + bytes(rng.integers(0, 256, 1_500_000, dtype=np.uint8)) + bytes(700000) + datagen.block(2, 3_000_000).tobytes() + text[:1234567]
    assert len(data) > 32 * 1024 * 1024 + 1000000
    for bs, chain, ent in ((32 * 1024 * 1024, "BWT+RANK+ZRLT", "ANS0"), (20 * 1024 * 1024, "BWT+MTFT+ZRLT", "HUFFMAN")):
        ref = oracle.compress(chain, ent, bs, data, jobs=2)
        cos = kz.CompressedOutputStream(ctx, chain, ent, bs)
        cos.write(data)
        cos.close()
        assert cos.output == ref, (bs, chain)
        assert kz.CompressedInputStream(ctx, ref).read() == data


def test_levels_1_to_3_tail_at_4mib_blocks(ctx):
    """The reference's level 1 (LZX & NONE), level 2 (DNA+LZ & HUFFMAN) and the tail of level 3 (PACK+MM+LZX &
    HUFFMAN) at the default 4 MiB block size, on a stream whose blocks take every route: text (digram aliases), DNA
    (2-bit packing, LZ with minMatch 6), 16-bit samples and RGB pixels (MM), random bytes, a short tail."""
    rng = np.random.default_rng(12)
    bs = 4 * 1024 * 1024
    words = [bytes(rng.integers(97, 123, int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(500)]
    text = b" ".join(words[int(i)] for i in rng.integers(0, 500, 900_000))[:bs]
    dna = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, bs)])
    data = text + dna + refinputs.multimedia_like(0, bs) + refinputs.multimedia_like(1, bs) + bytes(rng.integers(0, 256, bs, dtype=np.uint8)) + text[:123457]
    for chain, ent in (("LZX", "NONE"), ("DNA+LZ", "HUFFMAN"), ("PACK+MM+LZX", "HUFFMAN")):
        ref = oracle.compress(chain, ent, bs, data, jobs=3, checksum=32)
        cos = kz.CompressedOutputStream(ctx, chain, ent, bs, checksum=32)
        cos.write(data)
        cos.close()
        assert cos.output == ref, chain
        assert kz.CompressedInputStream(ctx, ref).read() == data


@pytest.mark.parametrize("force", ["lanes", "waves"])
def test_fpaq_both_arrangements(ctx, force, monkeypatch):
    """FPAQ has two kernel arrangements (one wave per block for batches up to 8 blocks per CU, one lane per block above):
    both must produce the oracle's bits."""
    monkeypatch.setenv("KZ_FPAQ_FORCE", force)
    data = datagen.stream(6, 65536).tobytes() + bytes(300) + bytes(range(256)) * 3
    for chain, bs in (("BWT+SRT+ZRLT", 65536), ("NONE", 16384), ("LZ", 1 << 20)):
        ref = oracle.compress(chain, "FPAQ", bs, data, jobs=2)
        cos = kz.CompressedOutputStream(ctx, chain, "FPAQ", bs)
        cos.write(data)
        cos.close()
        assert cos.output == ref, (force, chain)
        assert kz.CompressedInputStream(ctx, ref).read() == data
    big = (datagen.block(1, 4 * 1024 * 1024 + 4096).tobytes())          # > 4 MiB of FPAQ input: two coder chunks
    ref = oracle.compress("NONE", "FPAQ", 8 * 1024 * 1024, big, jobs=1)
    cos = kz.CompressedOutputStream(ctx, "NONE", "FPAQ", 8 * 1024 * 1024)
    cos.write(big)
    cos.close()
    assert cos.output == ref, force
    assert kz.CompressedInputStream(ctx, ref).read() == big


def test_device_resident_buffers_match_host_buffers(ctx):
    """memKind = KZ_MEM_DEVICE (what bench.py times): same streams as with host buffers, and the decoder restores the input."""
    torch = pytest.importorskip("torch")
    bs, B = 65536, 24
    inp = np.stack([datagen.block(i, bs) for i in range(B)])
    lens = np.full(B, bs, dtype=np.int32)
    lens[5], lens[7] = 12, 0                                         # a copy block and an empty one
    ostride = kz.max_block_stream_bytes(bs)
    out_h = np.zeros((B, ostride), dtype=np.uint8)
    res_h = kz.encode_blocks(ctx, "BWT+RANK+ZRLT", "ANS0", inp, bs, lens, out_h, ostride)
    d_in = torch.from_numpy(inp).cuda()
    d_out = torch.zeros((B, ostride), dtype=torch.uint8, device="cuda")
    res_d = kz.encode_blocks(ctx, "BWT+RANK+ZRLT", "ANS0", d_in.data_ptr(), bs, lens, d_out.data_ptr(), ostride, kz.MEM_DEVICE)
    out_d = d_out.cpu().numpy()
    for i in range(B):
        assert (res_d[i].bits, res_d[i].length, res_d[i].status) == (res_h[i].bits, res_h[i].length, res_h[i].status)
        nby = (res_h[i].bits + 7) // 8
        assert out_d[i, :nby].tobytes() == out_h[i, :nby].tobytes()
    bits = np.array([r.bits for r in res_d], dtype=np.int64)
    d_dec = torch.zeros((B, bs), dtype=torch.uint8, device="cuda")
    res2 = kz.decode_blocks(ctx, "BWT+RANK+ZRLT", "ANS0", bs, d_out.data_ptr(), ostride, bits, d_dec.data_ptr(), bs, kz.MEM_DEVICE)
    dec = d_dec.cpu().numpy()
    for i in range(B):
        assert res2[i].status == 0 and res2[i].length == lens[i]
        assert dec[i, :lens[i]].tobytes() == inp[i, :lens[i]].tobytes()


def test_small_blocks_that_grow_by_more_than_a_quarter(ctx):
    """1 KiB blocks of random bytes through SRT: 256 frequencies in front of every block (SRT.java MAX_HEADER_SIZE) make
    the stream 28 % larger than the input.  The writer keeps going like the reference's; `kz_compress_bound` is the
    destination size that always suffices, and a smaller destination is reported as ERR_WRITE_FILE, not overrun."""
    rng = np.random.default_rng(9103)
    data = bytes(rng.integers(0, 256, 96 * 1024 + 77, dtype=np.uint8))
    for chain, ent in (("MM+BWT+SRT+ZRLT", "HUFFMAN"), ("SRT", "ANS0")):
        ref = oracle.compress(chain, ent, 1024, data, jobs=2, checksum=64)
        assert len(ref) > len(data) + len(data) // 4
        assert len(ref) <= ctx.lib.kz_compress_bound(len(data), 1024)
        cos = kz.CompressedOutputStream(ctx, chain, ent, 1024, checksum=64)
        cos.write(data)
        cos.close()
        assert cos.output == ref
        assert kz.CompressedInputStream(ctx, ref).read(len(data)) == data
    src = np.frombuffer(data, dtype=np.uint8)
    small = np.empty(len(data) + len(data) // 8, dtype=np.uint8)
    ctx.set_checksum(64)
    try:
        rc = ctx.lib.kz_compress(ctx.h, kz.transform_type("SRT"), kz.ENTROPY_IDS["ANS0"], 1024, src.ctypes.data, len(src), small.ctypes.data, len(small))
    finally:
        ctx.set_checksum(0)
    assert rc == -12


LEVEL_CHAINS = [kz.level_chain(3), kz.level_chain(5), kz.level_chain(6)]


@pytest.mark.parametrize("chain,ent", LEVEL_CHAINS)
def test_level_exact_streams_match_oracle(ctx, chain, ent):
    """The reference's levels 3, 5 and 6 with their TEXT+UTF head (host stages in front of the GPU chain): whole .knz streams equal
    the oracle's on English / CRLF / XML text, UTF-8, binary and synthetic blocks, and decode back."""
    c = textgen.cases()
    data = (c["english"][:150000] + c["utf8"] + c["random"][:40000] + c["english_crlf"][:70000] + c["xml"][:60000] + c["english_escapes"][:50000]
            + datagen.stream(3, 30000).tobytes() + c["gif_magic_text"] + c["spaces_then_text"] + c["utf8_bom"])
    for bs in (32768, 262144):
        cos = kz.CompressedOutputStream(ctx, chain, ent, bs)
        cos.write(data)
        cos.close()
        ref = oracle.compress(chain, ent, bs, data, jobs=4)
        assert cos.output == ref, (chain, ent, bs, len(cos.output), len(ref))
        assert kz.CompressedInputStream(ctx, ref).read() == data
        # the skip flags differ from block to block: TEXT takes the prose, UTF the UTF-8, neither the binary blocks
        if bs == 32768:
            modes = [kz.extract_bits(ref, off, 16) for off, nb in kz.knz_index(ref)["blocks"]]
            assert len(set(modes)) > 2
    # checksummed, device-resident blocks through the batched calls
    torch = pytest.importorskip("torch")
    bs = 65536
    nb = len(data) // bs
    blocks = np.frombuffer(data[:nb * bs], dtype=np.uint8).reshape(nb, bs).copy()
    lens = np.full(nb, bs, dtype=np.int32)
    ostride = kz.max_block_stream_bytes(bs)
    ctx.set_block_size(bs)
    try:
        out_h = np.zeros((nb, ostride), dtype=np.uint8)
        res_h = kz.encode_blocks(ctx, chain, ent, blocks, bs, lens, out_h, ostride)
        d_in = torch.from_numpy(blocks).cuda()
        d_out = torch.zeros((nb, ostride), dtype=torch.uint8, device="cuda")
        res_d = kz.encode_blocks(ctx, chain, ent, d_in.data_ptr(), bs, lens, d_out.data_ptr(), ostride, kz.MEM_DEVICE)
        out_d = d_out.cpu().numpy()
        for i in range(nb):
            s, w, sf, pl = oracle.encode_block(chain, ent, blocks[i], block_size=bs)
            assert res_h[i].status == 0 and res_d[i].status == 0
            assert (res_h[i].bits, res_h[i].skipFlags, res_h[i].length) == (w, sf, pl) == (res_d[i].bits, res_d[i].skipFlags, res_d[i].length), i
            assert out_h[i, :(w + 7) // 8].tobytes() == s and out_d[i, :(w + 7) // 8].tobytes() == s, i
        bits = np.array([r.bits for r in res_d], dtype=np.int64)
        d_dec = torch.zeros((nb, bs), dtype=torch.uint8, device="cuda")
        res2 = kz.decode_blocks(ctx, chain, ent, bs, d_out.data_ptr(), ostride, bits, d_dec.data_ptr(), bs, kz.MEM_DEVICE)
        assert all(r.status == 0 and r.length == bs for r in res2)
        assert np.array_equal(d_dec.cpu().numpy(), blocks)
    finally:
        ctx.set_block_size(4 * 1024 * 1024)


def test_text_and_utf_through_the_transform_mirror(ctx):
    """kz_transform_forward / _inverse for TEXT and UTF (the ByteTransform mirror with a context): same answers as the
    context-free host entry points and the oracle, dataType entry included."""
    c = textgen.cases()
    for ent in ("ANS0", "FPAQ"):
        t = kz.TextCodec(ctx, ent, 65536)
        oracle.set_transform_ctx(ent, 65536)
        for name in ("english", "utf8", "random", "xml"):
            ctx.set_data_type(0)
            ok, enc = _fwd(ctx, kz.TEXT_TYPE, c[name])
            ok_o, enc_o, dt_o = oracle.transform_forward("TEXT", c[name], data_type=0)
            assert ok == ok_o and ctx.get_data_type() == dt_o, (ent, name)
            if ok:
                assert enc == enc_o
                assert _inv(ctx, kz.TEXT_TYPE, enc, len(c[name]) + 4096) == (True, c[name])
            ok, enc = _fwd(ctx, kz.UTF_TYPE, c[name])
            ok_o, enc_o, dt_o2 = oracle.transform_forward("UTF", c[name], data_type=dt_o)
            assert ok == ok_o and ctx.get_data_type() == dt_o2, (ent, name)
            if ok:
                assert enc == enc_o and _inv(ctx, kz.UTF_TYPE, enc, len(c[name]) + 4096) == (True, c[name])
        del t
    ctx.set_data_type(0)
    ctx.set_entropy("NONE")
    ctx.set_block_size(4 * 1024 * 1024)
    # a host stage behind a GPU stage is not a chain the reference's levels use: refused, not mis-coded
    with pytest.raises(kz.KanziError) as e:
        cos = kz.CompressedOutputStream(ctx, "BWT+TEXT", "ANS0", 65536)
        cos.write(c["english"][:100000])
        cos.close()
    assert e.value.code == 3


def test_level5_at_4mib_blocks(ctx):
    """-l 5 at its own block size: 3 text-like 4 MiB blocks + one synthetic, HIP .knz == oracle's, round trip."""
    bs = 4 * 1024 * 1024
    data = textgen.english(bs, 31) + textgen.many_words(bs, 32) + textgen.utf8(bs, 33) + datagen.block(2, bs).tobytes()[:bs // 2]
    chain, ent = kz.level_chain(5)
    cos = kz.CompressedOutputStream(ctx, chain, ent, bs)
    cos.write(data)
    cos.close()
    assert cos.output == oracle.compress(chain, ent, bs, data, jobs=4)
    assert kz.CompressedInputStream(ctx, cos.output).read() == data


def test_async_batches_on_two_contexts(ctx):
    """kz_submit_* / kz_wait: one host thread keeps two contexts busy (encode of batch k+1 under the decode of batch k); results equal
    the synchronous calls'."""
    bs, B = 65536, 16
    ctx2 = kz.Context(0)
    try:
        batches = [np.stack([datagen.block(k * B + i, bs) for i in range(B)]) for k in range(3)]
        lens = np.full(B, bs, dtype=np.int32)
        ostride = kz.max_block_stream_bytes(bs)
        sync = []
        for inp in batches:
            out = np.zeros((B, ostride), dtype=np.uint8)
            res = kz.encode_blocks(ctx, "BWT+RANK+ZRLT", "ANS0", inp, bs, lens, out, ostride)
            sync.append((out, [(r.bits, r.length, r.skipFlags) for r in res]))
        outs = [np.zeros((B, ostride), dtype=np.uint8) for _ in batches]
        decs = [np.zeros((B, bs), dtype=np.uint8) for _ in batches]
        enc_job = kz.submit_encode_blocks(ctx, "BWT+RANK+ZRLT", "ANS0", batches[0], bs, lens, outs[0], ostride)
        dec_jobs = []
        for k in range(len(batches)):
            res = enc_job.wait()
            assert [(r.bits, r.length, r.skipFlags) for r in res] == sync[k][1]
            for i in range(B):
                nby = (res[i].bits + 7) // 8
                assert outs[k][i, :nby].tobytes() == sync[k][0][i, :nby].tobytes()
            if k + 1 < len(batches):
                enc_job = kz.submit_encode_blocks(ctx, "BWT+RANK+ZRLT", "ANS0", batches[k + 1], bs, lens, outs[k + 1], ostride)
            bits = np.array([r.bits for r in res], dtype=np.int64)
            dec_jobs.append(kz.submit_decode_blocks(ctx2, "BWT+RANK+ZRLT", "ANS0", bs, outs[k], ostride, bits, decs[k], bs))
        for k, j in enumerate(dec_jobs):
            res2 = j.wait()
            assert all(r.status == 0 and r.length == bs for r in res2)
            assert np.array_equal(decs[k], batches[k])
        assert ctx.lib.kz_wait(ctx.h, 10 ** 6) == -18                     # unknown job: ERR_INVALID_PARAM
    finally:
        ctx2.close()


def test_overlapped_rank_bwt_inverse_schedule(ctx, monkeypatch):
    """Large batches decode the expensive blocks' RANK inverse on a side stream while the other blocks go on to the BWT
    inverse (kz_api.hip: overlapped_rank_bwt_inverse).  Forced here on a small batch: same bytes, lengths and statuses as the
    stage-by-stage schedule, corrupted blocks and raw / copy blocks included."""
    bs, B = 65536, 40
    inp = np.stack([datagen.block(i, bs) for i in range(B)])
    lens = np.full(B, bs, dtype=np.int32)
    lens[3], lens[11], lens[17] = 9, 0, 30000
    ostride = kz.max_block_stream_bytes(bs)
    for chain in ("BWT+RANK+ZRLT", "BWT+MTFT+ZRLT"):
        out = np.zeros((B, ostride), dtype=np.uint8)
        res = kz.encode_blocks(ctx, chain, "ANS0", inp, bs, lens, out, ostride)
        bits = np.array([r.bits for r in res], dtype=np.int64)
        bad = out.copy()
        bad[5, 40:60] ^= 0x5A                       # damage inside one block's payload
        bad[22, 3] ^= 0x01                          # and inside another block's header
        results = {}
        for mode, env in (("staged", "1000000"), ("overlapped", "8")):
            monkeypatch.setenv("KZ_FUSE_MIN_BLOCKS", env)
            for name, streams in (("good", out), ("bad", bad)):
                dec = np.zeros((B, bs), dtype=np.uint8)
                r2 = kz.decode_blocks(ctx, chain, "ANS0", bs, streams, ostride, bits, dec, bs)
                results[(mode, name)] = ([(r.status, r.length) for r in r2], [dec[i, :max(r2[i].length, 0)].tobytes() for i in range(B)])
        assert results[("staged", "good")] == results[("overlapped", "good")]
        assert results[("staged", "bad")] == results[("overlapped", "bad")]
        st = results[("overlapped", "good")][0]
        assert all(s == 0 and l == lens[i] for i, (s, l) in enumerate(st))
        assert all(results[("overlapped", "good")][1][i] == inp[i, :lens[i]].tobytes() for i in range(B))


def test_expensive_blocks_first_decode_schedule(ctx, monkeypatch):
    """BWT+RANK+ZRLT / BWT+MTFT+ZRLT batches without empty or damaged-header blocks take the "expensive blocks first" decoder
    schedule (kz_api.hip): the expensive group runs entropy decoding and ZRLT inverse first and starts its RANK inverse on the
    side stream while the other group is still being entropy decoded.  Same results as the staged schedule, transformed-copy
    blocks (uniform data), blocks with ZRLT skipped and a damaged payload included."""
    bs, B = 131072, 40
    inp = np.stack([datagen.block(i, bs) for i in range(B)])
    lens = np.full(B, bs, dtype=np.int32)
    lens[7], lens[18] = 50000, 4097
    ostride = kz.max_block_stream_bytes(bs)
    for chain, ent in (("BWT+RANK+ZRLT", "ANS0"), ("BWT+MTFT+ZRLT", "HUFFMAN")):
        out = np.zeros((B, ostride), dtype=np.uint8)
        res = kz.encode_blocks(ctx, chain, ent, inp, bs, lens, out, ostride)
        assert any(out[i, 0] & 0x90 == 0x90 for i in range(B))            # transformed-copy blocks are in the batch
        bits = np.array([r.bits for r in res], dtype=np.int64)
        bad = out.copy()
        bad[5, 400:420] ^= 0x5A                     # inside one block's payload: the header still parses
        results = {}
        # "first": four streams (every class's RANK inverse on a side stream, planned order); "first3": the three-stream form
        # taken when the process has too few hardware queues
        for mode, fuse, nosf, wide in (("staged", "1000000", None, None), ("overlapped", "8", "1", "1"), ("overlapped3", "8", "1", "0"),
                                       ("first", "8", None, "1"), ("first3", "8", None, "0")):
            monkeypatch.setenv("KZ_FUSE_MIN_BLOCKS", fuse)
            if nosf:
                monkeypatch.setenv("KZ_NO_SFIRST", nosf)
            else:
                monkeypatch.delenv("KZ_NO_SFIRST", raising=False)
            if wide:
                monkeypatch.setenv("KZ_WIDE_QUEUES", wide)
            else:
                monkeypatch.delenv("KZ_WIDE_QUEUES", raising=False)
            for name, streams in (("good", out), ("bad", bad)):
                dec = np.zeros((B, bs), dtype=np.uint8)
                r2 = kz.decode_blocks(ctx, chain, ent, bs, streams, ostride, bits, dec, bs)
                results[(mode, name)] = ([(r.status, r.length) for r in r2], [dec[i, :max(r2[i].length, 0)].tobytes() for i in range(B)])
        for name in ("good", "bad"):
            for mode in ("overlapped", "overlapped3", "first", "first3"):
                assert results[("staged", name)] == results[(mode, name)], (chain, name, mode)
        st = results[("first", "good")][0]
        assert all(s == 0 and l == lens[i] for i, (s, l) in enumerate(st))
        assert all(results[("first", "good")][1][i] == inp[i, :lens[i]].tobytes() for i in range(B))



def test_text_block_size_is_fixed_at_call_time(ctx):
    """TEXT sizes its hash map by the stream's block size.  A chain with TEXT is refused until the context has been told
    that size; blocks coded at two block sizes decode with their own size; a queued job keeps the size of its submit time."""
    c = textgen.cases()
    data = (c["english"] + c["many_words"])[:4 * 131072]
    fresh = kz.Context(0)
    try:
        blocks = np.frombuffer(data, dtype=np.uint8).reshape(4, 131072).copy()
        lens = np.full(4, 131072, dtype=np.int32)
        ostride = kz.max_block_stream_bytes(131072)
        out = np.zeros((4, ostride), dtype=np.uint8)
        with pytest.raises(kz.KanziError) as e:
            kz.encode_blocks(fresh, "TEXT+BWT+RANK+ZRLT", "ANS0", blocks, 131072, lens, out, ostride)
        assert e.value.code == 1                                          # ERR_MISSING_PARAM: "blockSize" was never set
        assert fresh.lib.kz_submit_encode_blocks(fresh.h, kz.transform_type("TEXT+BWT"), 5, blocks.ctypes.data, 131072, lens.ctypes.data, 4,
                                                 out.ctypes.data, ostride, 0, 0) == -1
        coded = {}
        for stream_bs in (131072, 1 << 20):                              # two streams with different block sizes, same block bytes
            fresh.set_block_size(stream_bs)
            res = kz.encode_blocks(fresh, "TEXT+BWT+RANK+ZRLT", "ANS0", blocks, 131072, lens, out, ostride)
            for i in range(4):
                s, w, sf, pl = oracle.encode_block("TEXT+BWT+RANK+ZRLT", "ANS0", blocks[i], block_size=stream_bs)
                assert (res[i].bits, res[i].skipFlags, res[i].length) == (w, sf, pl)
                assert out[i, :(w + 7) // 8].tobytes() == s
            coded[stream_bs] = (out.copy(), np.array([r.bits for r in res], dtype=np.int64))
        for stream_bs, (o, bits) in coded.items():
            dec = np.zeros((4, stream_bs), dtype=np.uint8)
            res2 = kz.decode_blocks(fresh, "TEXT+BWT+RANK+ZRLT", "ANS0", stream_bs, o, ostride, bits, dec, stream_bs)
            assert all(r.status == 0 and r.length == 131072 for r in res2)
            assert np.array_equal(dec[:, :131072], blocks)
        # a queued job keeps the block size of its submit time
        fresh.set_block_size(131072)
        out2 = np.zeros((4, ostride), dtype=np.uint8)
        job = kz.submit_encode_blocks(fresh, "TEXT+BWT+RANK+ZRLT", "ANS0", blocks, 131072, lens, out2, ostride)
        res = job.wait()
        fresh.set_block_size(1 << 20)
        for i in range(4):
            nby = (res[i].bits + 7) // 8
            assert res[i].bits == coded[131072][1][i] and out2[i, :nby].tobytes() == coded[131072][0][i, :nby].tobytes()
        # a job's result is handed out once
        assert fresh.lib.kz_wait(fresh.h, job.job) == -18
        assert fresh.lib.kz_poll(fresh.h, job.job) == -18
    finally:
        fresh.close()
    with pytest.raises(kz.KanziError) as e:
        ctx.set_entropy(9)                                                # TPAQX: TEXT's extra hash bit is not modelled
    assert e.value.code == 3


_FULL_SIZE_CACHE = {}


@pytest.mark.parametrize("chain,ent,nb", [("BWT+RANK+ZRLT", "ANS0", 45), ("LZ", "ANS0", 40), ("BWT+SRT+ZRLT", "FPAQ", 40), ("LZX", "HUFFMAN", 40)])
def test_full_size_batch_of_45_blocks_matches_oracle(ctx, chain, ent, nb):
    """the decoder's cost-class schedule (batches >= 32 blocks: four streams, "expensive first") and the suffix sort's
    bucket path at the full 4 MiB block size, compared with the ORACLE and not just round-tripped: 45 device-resident blocks (9 of
    every synthetic class, the last one ragged) through kz_encode_blocks -> every block stream equals the oracle's .knz payload;
    kz_decode_blocks of the batch restores the input.  The other BASELINE chains likewise at 40 blocks (configs[1] LZ & ANS0,
    configs[4] BWT+SRT+ZRLT & FPAQ with its cost-aware placement of the serial coders, level 3's LZX & HUFFMAN tail)."""
    torch = pytest.importorskip("torch")
    bs = 4 * 1024 * 1024
    if "blocks" not in _FULL_SIZE_CACHE:                                  # (generated once for the four chains)
        _FULL_SIZE_CACHE["blocks"] = np.stack([datagen.block(100 + i, bs) for i in range(45)])
    host = _FULL_SIZE_CACHE["blocks"][:nb]
    lens = np.full(nb, bs, dtype=np.int32)
    lens[-1] = 2242560                                                    # silesia.tar's tail
    ostride = kz.max_block_stream_bytes(bs)
    d_in = torch.from_numpy(host).cuda()
    d_out = torch.zeros((nb, ostride), dtype=torch.uint8, device="cuda")
    res = kz.encode_blocks(ctx, chain, ent, d_in.data_ptr(), bs, lens, d_out.data_ptr(), ostride, kz.MEM_DEVICE)
    assert all(r.status == 0 for r in res)
    out = d_out.cpu().numpy()
    bits = [int(r.bits) for r in res]
    data = host.reshape(-1)[:(nb - 1) * bs + int(lens[-1])]
    knz = kz.knz_assemble(chain, ent, bs, len(data), [out[i, :(bits[i] + 7) // 8].tobytes() for i in range(nb)], bits)
    ref = oracle.compress(chain, ent, bs, data, jobs=min(16, os.cpu_count() or 1))
    assert knz == ref, (len(knz), len(ref))
    d_dec = torch.zeros((nb, bs), dtype=torch.uint8, device="cuda")
    res2 = kz.decode_blocks(ctx, chain, ent, bs, d_out.data_ptr(), ostride, np.array(bits, dtype=np.int64), d_dec.data_ptr(), bs, kz.MEM_DEVICE)
    assert [r.length for r in res2] == list(lens) and all(r.status == 0 for r in res2)
    dec = d_dec.cpu().numpy()
    assert np.array_equal(dec[:-1], host[:-1]) and np.array_equal(dec[-1, :lens[-1]], host[-1, :lens[-1]])


@pytest.mark.parametrize("level", [5, 6, 3])
def test_host_stage_pipeline_gives_the_same_blocks(ctx, monkeypatch, level):
    """Chains led by TEXT / UTF on large batches run the host stages of chunk k+1 under the GPU work of chunk k (encode) and the host
    inverse of chunk k under the GPU work of chunk k+1 (decode).  Forced here with 8-block chunks on 44 blocks (ragged last chunk),
    host and device memory: every block stream equals the oracle's, decoding restores the input."""
    torch = pytest.importorskip("torch")
    monkeypatch.setenv("KZ_HOST_CHUNK", "8")
    monkeypatch.setenv("KZ_HOST_CHUNK_DEC", "8")
    chain, ent = kz.level_chain(level)
    c = textgen.cases()
    bs, nb = 65536, 44
    pool = c["english"] + c["utf8"] + c["xml"] + datagen.stream(6, bs).tobytes() + c["english_crlf"] + c["many_words"] + c["utf8_bom"] + c["random"]
    data = (pool * (nb * bs // len(pool) + 1))[:nb * bs]
    blocks = np.frombuffer(data, dtype=np.uint8).reshape(nb, bs).copy()
    blocks[7, :10] = 0                                                    # a block TEXT declines next to ones it takes
    lens = np.full(nb, bs, dtype=np.int32)
    lens[-1] = 12345
    lens[3] = 9                                                           # a copy block (<= 15 bytes) in the middle of a chunk
    ostride = kz.max_block_stream_bytes(bs)
    ctx.set_block_size(bs)
    try:
        ref = [oracle.encode_block(chain, ent, blocks[i, :lens[i]], block_size=bs) for i in range(nb)]
        out_h = np.zeros((nb, ostride), dtype=np.uint8)
        res_h = kz.encode_blocks(ctx, chain, ent, blocks, bs, lens, out_h, ostride)
        d_in = torch.from_numpy(blocks).cuda()
        d_out = torch.zeros((nb, ostride), dtype=torch.uint8, device="cuda")
        res_d = kz.encode_blocks(ctx, chain, ent, d_in.data_ptr(), bs, lens, d_out.data_ptr(), ostride, kz.MEM_DEVICE)
        out_d = d_out.cpu().numpy()
        flags = set()
        for i in range(nb):
            s, w, sf, pl = ref[i]
            assert res_h[i].status == 0 and res_d[i].status == 0
            assert (res_h[i].bits, res_h[i].skipFlags, res_h[i].length) == (w, sf, pl) == (res_d[i].bits, res_d[i].skipFlags, res_d[i].length), i
            assert out_h[i, :(w + 7) // 8].tobytes() == s and out_d[i, :(w + 7) // 8].tobytes() == s, i
            flags.add(sf)
        assert len(flags) >= 3                                            # TEXT taken / UTF taken / neither
        bits = np.array([r.bits for r in res_d], dtype=np.int64)
        dec_h = np.zeros((nb, bs), dtype=np.uint8)
        res2 = kz.decode_blocks(ctx, chain, ent, bs, out_h, ostride, bits, dec_h, bs)
        d_dec = torch.zeros((nb, bs), dtype=torch.uint8, device="cuda")
        res3 = kz.decode_blocks(ctx, chain, ent, bs, d_out.data_ptr(), ostride, bits, d_dec.data_ptr(), bs, kz.MEM_DEVICE)
        dec_d = d_dec.cpu().numpy()
        for i in range(nb):
            assert res2[i].status == 0 and res3[i].status == 0 and res2[i].length == lens[i] == res3[i].length, i
            assert np.array_equal(dec_h[i, :lens[i]], blocks[i, :lens[i]]) and np.array_equal(dec_d[i, :lens[i]], blocks[i, :lens[i]]), i
        # the opt-in staged form of the host inverse (pinned slots, one gather / scatter kernel per sub-chunk) gives the same blocks
        monkeypatch.setenv("KZ_HOST_INV_STAGED", "1")
        d_dec.zero_()
        res5 = kz.decode_blocks(ctx, chain, ent, bs, d_out.data_ptr(), ostride, bits, d_dec.data_ptr(), bs, kz.MEM_DEVICE)
        monkeypatch.delenv("KZ_HOST_INV_STAGED")
        dec_s = d_dec.cpu().numpy()
        for i in range(nb):
            assert res5[i].status == 0 and res5[i].length == lens[i] and np.array_equal(dec_s[i, :lens[i]], blocks[i, :lens[i]]), i
        # a damaged TEXT block in the middle of a chunk fails alone (ERR_PROCESS_BLOCK), its neighbours decode
        if level == 5:
            bad = out_h.copy()
            victim = next(i for i in range(8, nb) if not (res_h[i].skipFlags & 0x80))
            bad[victim, 40:60] ^= 0x5A
            res4 = kz.decode_blocks(ctx, chain, ent, bs, bad, ostride, bits, dec_h, bs)
            assert all(res4[i].status == 0 for i in range(nb) if i != victim)
    finally:
        ctx.set_block_size(4 * 1024 * 1024)


@pytest.mark.parametrize("chain,ent,chk", [("BWT+RANK+ZRLT", "ANS0", 0), ("TEXT+UTF+BWT+RANK+ZRLT", "ANS0", 32), ("LZX", "NONE", 64), ("BWT+SRT+ZRLT", "FPAQ", 0)])
def test_pipelined_stream_writer_and_staged_reader(ctx, monkeypatch, chain, ent, chk):
    """kz_compress on inputs of several chunks runs a three-stage pipeline (stage chunk k+1 | code chunk k | bring chunk k-1 back and emit
    it in order), kz_decompress moves whole batches through pinned staging: forced here with 4-block chunks over 37 blocks + a
    ragged tail.  The stream equals the oracle's and the one-batch-at-a-time form's; both readers restore the input."""
    c = textgen.cases()
    bs = 65536
    data = (c["english"] + datagen.stream(20, bs).tobytes() + c["utf8"] + c["xml"] + c["random"])[:37 * bs + 4321]
    ctx.set_checksum(chk)
    try:
        ref = oracle.compress(chain, ent, bs, data, jobs=4, checksum=chk)
        monkeypatch.setenv("KZ_STREAM_CHUNK", "4")
        cos = kz.CompressedOutputStream(ctx, chain, ent, bs, checksum=chk)
        cos.write(data)
        cos.close()
        assert cos.output == ref, (len(cos.output), len(ref))
        assert kz.CompressedInputStream(ctx, ref).read() == data
        # the writer's "skipBlocks" option decides copy blocks on the device BEFORE the host stages: the pipeline then leaves the host
        # stages to the batched call (no pre-staging) and the stream still equals the oracle's
        skip_ref = oracle.compress(chain, ent, bs, data, jobs=4, checksum=chk, skip_blocks=True)
        cos3 = kz.CompressedOutputStream(ctx, chain, ent, bs, checksum=chk, skipBlocks=True)
        cos3.write(data)
        cos3.close()
        assert cos3.output == skip_ref and kz.CompressedInputStream(ctx, skip_ref).read() == data
        monkeypatch.setenv("KZ_STREAM_SERIAL", "1")
        cos2 = kz.CompressedOutputStream(ctx, chain, ent, bs, checksum=chk)
        cos2.write(data)
        cos2.close()
        assert cos2.output == ref
        assert kz.CompressedInputStream(ctx, ref).read() == data
        monkeypatch.delenv("KZ_STREAM_SERIAL")
        # a destination that is too small: ERR_WRITE_FILE from the pipeline's emitter, and the context still works afterwards
        lib = ctx.lib
        src = np.frombuffer(data, dtype=np.uint8)
        small = np.zeros(len(ref) // 2, dtype=np.uint8)
        ctx.set_block_size(bs)
        assert lib.kz_compress(ctx.h, kz.transform_type(chain), kz.ENTROPY_IDS[ent], bs, src.ctypes.data, len(src), small.ctypes.data, len(small)) == -12
        back = np.zeros(len(data), dtype=np.uint8)
        knz = np.frombuffer(ref, dtype=np.uint8)
        assert lib.kz_decompress(ctx.h, knz.ctypes.data, len(knz), back.ctypes.data, len(back)) == len(data) and back.tobytes() == data
        # and a destination one block short for the reader: ERR_WRITE_FILE as before
        short = np.zeros(len(data) - bs, dtype=np.uint8)
        assert lib.kz_decompress(ctx.h, knz.ctypes.data, len(knz), short.ctypes.data, len(short)) == -12
    finally:
        ctx.set_checksum(0)
        ctx.set_block_size(4 * 1024 * 1024)


def test_prestaged_batches_larger_than_the_arena_are_split(ctx, tmp_path):
    """kz_compress always pre-stages the host stages of a TEXT / UTF chain per chunk; a chunk larger than the arena budget
    allows (little free HBM, KZ_ARENA_BUDGET_GB, a large KZ_STREAM_CHUNK) used to fail with ERR_INVALID_PARAM where the plain
    path splits.  The budget is read once per process, so the small-budget run is a subprocess: 72 blocks of 1 MiB, chunks of
    64, a 1 GiB budget; its stream must equal this process's."""
    import subprocess
    import sys
    bs = 1 << 20
    blocks = [bytes(textgen.bulk_text(bs, 2000 + i, ("english", "xml", "utf8")[i % 3])) for i in range(6)]
    data = b"".join(blocks[i % 6] for i in range(72)) + b"the tail of the stream"
    src = tmp_path / "in.bin"
    src.write_bytes(data)
    chain, ent = kz.level_chain(5)
    cos = kz.CompressedOutputStream(ctx, chain, ent, bs)
    cos.write(data)
    cos.close()
    prog = ("import os, sys\nsys.path.insert(0, %r)\nimport torch\nimport kanzi_amd as kz\nc = kz.Context(0)\n"
            "d = open(%r, 'rb').read()\ncos = kz.CompressedOutputStream(c, %r, %r, %d)\ncos.write(d)\ncos.close()\n"
            "open(%r, 'wb').write(cos.output)\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(src), chain, ent, bs, str(tmp_path / "out.knz"))
    env = dict(os.environ, KZ_ARENA_BUDGET_GB="1", KZ_STREAM_CHUNK="64", GPU_MAX_HW_QUEUES="8")
    r = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert (tmp_path / "out.knz").read_bytes() == cos.output
    assert kz.CompressedInputStream(ctx, cos.output).read() == data


def _trie_inputs():
    """Inputs of 64 KiB and more (the trie rounds of kz_bwt_fwd.hip apply from 64 KiB up): every synthetic class around the size
    threshold, long runs, periods, tiny alphabets, a few huge groups, text that ends in zeros (padding vs real zero bytes)."""
    rng = np.random.default_rng(5)

    def repeats(n, nwords, wlen):
        words = rng.integers(0, 256, (nwords, wlen), dtype=np.uint8)
        return words[rng.integers(0, nwords, n // wlen + 1)].reshape(-1)[:n].tobytes()

    ins = []
    for n in (65536, 65537, 100000):
        for c in range(5):
            ins.append(datagen.block(c, n, c).tobytes())
    ins += [bytes(300000), bytes(65536), b"ab" * 200000, (b"abcdefghijklmnopqrstuvwxy" * 20000)[:499979],
            rng.integers(0, 3, 300000, dtype=np.uint8).tobytes(), rng.integers(0, 2, 200000, dtype=np.uint8).tobytes(),
            repeats(700000, 3, 33), repeats(1 << 19, 12, 40), repeats(300000, 900, 16),
            rng.integers(0, 256, 200000, dtype=np.uint8).tobytes() + bytes(70000), bytes(99999) + b"\x01",
            b"\xff" * 150000 + rng.integers(0, 256, 50000, dtype=np.uint8).tobytes()]
    # lazy ranks (k_tr_sort): blocks without an expanded node whose round 0 still leaves live suffixes (the second pass stores the ranks
    # after all): noise twice in a row (every suffix of the first copy live), noise with one planted repeat (a handful live), and
    # noise whose length is not a multiple of 8 (the 8 primary indexes are the only ranks a finished block stores)
    noise = rng.integers(0, 256, 300001, dtype=np.uint8).tobytes()
    ins += [noise[:150000] * 2, noise[:150000] + noise[1000:1100] + noise[150000:250000], noise, noise[:262147] + noise[:7]]
    return ins


@pytest.mark.parametrize("switch", ["default", "KZ_BWT_DMAX=7", "KZ_BWT_TRIEWIN=0", "KZ_BWT_TRIE=0", "KZ_BWT_RETIRE=0", "KZ_BWT_LAZYRANK=0"])
def test_bwt_forward_trie_rounds_match_oracle(ctx, monkeypatch, switch):
    """Round 0 as a trie round (count by byte, move once, finish buckets in LDS) and the key trie round over the window of oversized
    buckets in the doubling rounds, against the oracle's induced-sorting BWT (the BWT is unique): default depth 6, depth 7, the
    LSD window instead of the key round, the whole old path, and the later rounds without retiring finished blocks."""
    if switch != "default":
        k, v = switch.split("=")
        monkeypatch.setenv(k, v)
    for d in _trie_inputs():
        ok_o, enc_o = oracle.transform_forward("BWT", d)
        ok_g, enc_g = _fwd(ctx, kz.BWT_TYPE, d)
        assert ok_g == ok_o and enc_g == enc_o, (switch, len(d))
    if switch in ("default", "KZ_BWT_RETIRE=0"):
        # one ragged batch: blocks of 64 KiB .. 4 MiB and two short ones in the same call (the tables follow the longest block); the
        # uniform and geometric blocks are done after round 0 / 1 and leave the later rounds' grids (rows = blocks still live: the
        # list has holes), the text-like and sparse ones stay to the end; KZ_BWT_RETIRE=0 = every block in every round, as before
        blocks = ([datagen.block(3, 1 << 20, 3).tobytes(), datagen.block(0, 4 << 20, 0).tobytes(), datagen.block(1, 4 << 20, 1).tobytes(),
                   datagen.block(3, 4 << 20, 3).tobytes(), datagen.block(4, 4 << 20, 4).tobytes()]
                  + [_trie_inputs()[k] for k in (3, 15, 21, -4, -3)] + [b"short block", b"", datagen.block(8, 300000, 3).tobytes(), datagen.block(2, 1 << 21, 2).tobytes()])
        bs = 4 << 20
        B = len(blocks)
        inp = np.zeros((B, bs), dtype=np.uint8)
        lens = np.array([len(d) for d in blocks], dtype=np.int32)
        for i, d in enumerate(blocks):
            inp[i, :len(d)] = np.frombuffer(d, dtype=np.uint8)
        ostride = kz.max_block_stream_bytes(bs)
        out = np.zeros((B, ostride), dtype=np.uint8)
        res = kz.encode_blocks(ctx, "BWT", "NONE", inp, bs, lens, out, ostride)
        for i, d in enumerate(blocks):
            if len(d) == 0:
                continue
            so, w, sf, pl = oracle.encode_block("BWT", "NONE", d)
            assert res[i].status == 0 and (res[i].bits, res[i].skipFlags, res[i].length) == (w, sf, pl), len(d)
            assert out[i, :(w + 7) // 8].tobytes() == so, len(d)


# ---- round 5: the differential fuzz tools of round 4 as a driver-run test ----
@pytest.mark.gpu
@pytest.mark.parametrize("tool,seconds,seed", [("bwt_fuzz", 7, 5001), ("zrlt_fuzz", 6, 5002), ("tightcap_fuzz", 6, 5003), ("entropy_count_fuzz", 6, 5004), ("text_fwd_gpu_fuzz", 8, 5005), ("utf_fwd_gpu_fuzz", 10, 5006)])
def test_differential_fuzz_tools_bounded(tool, seconds, seed):
    """tools/bwt_fuzz.py (forward BWT on structured and degenerate strings, ragged batches), tools/zrlt_fuzz.py (row / wave / tile
    seams of the ZRLT kernels), tools/tightcap_fuzz.py (every inverse transform with the buffer cut to the byte) and
    tools/entropy_count_fuzz.py (decoders asked for the wrong count or given cut bits), tools/text_fwd_gpu_fuzz.py (the device TEXT
    forward on generated words around its rules, small blocks), tools/utf_fwd_gpu_fuzz.py (the device UTF forward on UTF-8 of
    two- and three-unit code points with mutations the pair statistics do not see): a fixed seed and a few seconds each, so the
    first cases of every campaign run wherever the GPU suite runs.  Each tool compares HIP with the oracle case by case and exits
    non-zero on the first campaign with a difference."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)          # the tools keep failing inputs there
    r = subprocess.run([sys.executable, os.path.join(root, "tools", tool + ".py"), str(seconds), str(seed)], capture_output=True, text=True, timeout=300)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, tail


@pytest.mark.gpu
@pytest.mark.parametrize("at_round", ["0", "1"])
def test_bwt_forward_falls_back_when_the_trie_tables_overflow(ctx, monkeypatch, at_round):
    """an overflow of the trie tables (round 0 or a key round of the window; the table bounds say it cannot happen) must
    not fail the batch nor leave mis-ranked suffixes behind: the stage is redone on the LSD rounds.  The overflow is simulated
    (KZ_BWT_TEST_TRIE_OVERFLOW=<round>)."""
    monkeypatch.setenv("KZ_BWT_TEST_TRIE_OVERFLOW", at_round)
    for i, c in enumerate((0, 2, 4, 1)):
        d = datagen.block(i, 300000, c).tobytes()
        ok_o, enc_o = oracle.transform_forward("BWT", d)
        ok_g, enc_g = _fwd(ctx, kz.BWT_TYPE, d)
        assert ok_g == ok_o and enc_g == enc_o, (at_round, c)


# ---- round 5: the TEXT inverse on the device (kz_text_gpu.hip, opt-in KZ_TEXT_GPU=1) ----
@pytest.mark.gpu
@pytest.mark.parametrize("form", ["", "1", "2", "3"])
@pytest.mark.parametrize("chain,ent", [("TEXT", "NONE"), ("TEXT", "FPAQ"), ("TEXT+UTF+BWT+RANK+ZRLT", "ANS0"), ("TEXT+UTF+BWT+SRT+ZRLT", "FPAQ")])
def test_text_inverse_on_the_device(ctx, chain, ent, form, monkeypatch, capfd):
    """Streams written by the oracle (TextCodec2 for NONE / ANS0, TextCodec1 for FPAQ; English, CRLF, XML, escape bytes, a dictionary
    that doubles and wraps, UTF-8, binary and short blocks) are decoded with the TEXT inverse running on the device: same bytes as the
    input (= what the oracle's and the host stage's decoders give), corrupted copies get the oracle's verdict (the device form hands
    whatever it cannot finish to the host stage), and the trace shows that the device did take blocks."""
    if form:
        monkeypatch.setenv("KZ_TEXT_GPU", form)
    else:
        monkeypatch.delenv("KZ_TEXT_GPU", raising=False)            # the default: the row form for TextCodec2 streams, the host stage for FPAQ's
        monkeypatch.setenv("KZ_TEXT_GPU_MIN", "1")                  # (... in batches of 512 blocks or more: any batch here)
    # 1: rows of 64 coded bytes, three waves in lockstep (TextCodec2 blocks), 2: the serial walk, 3: rows, one wave
    monkeypatch.setenv("KZ_TEXT_GPU_TRACE", "1")
    c = textgen.cases()
    data = (c["english"][:150000] + c["utf8"][:30000] + c["random"][:40000] + c["english_crlf"][:70000] + c["xml"][:60000] + c["english_escapes"][:50000]
            + c["many_words"][:300000] + datagen.stream(2, 30000).tobytes() + c["gif_magic_text"] + c["spaces_then_text"] + c["short"] + c["min"])
    rng = np.random.default_rng(17)
    for bs in (32768, 1 << 20):
        ref = oracle.compress(chain, ent, bs, data, jobs=4)
        assert kz.CompressedInputStream(ctx, ref).read() == data, (chain, ent, bs)
        for _ in range(6):                                        # corrupted copies: the verdict of the oracle's decoder
            bad = bytearray(ref)
            pos = int(rng.integers(40, len(bad) - 8))
            bad[pos] ^= 1 << int(rng.integers(0, 8))
            try:
                want = oracle.decompress(bytes(bad), len(data) + 4 * bs, jobs=2)
            except Exception:
                want = None
            try:
                got = kz.CompressedInputStream(ctx, bytes(bad)).read()
            except Exception:
                got = None
            assert (got is None) == (want is None) and (got is None or got == want), (chain, ent, bs, pos)
    err = capfd.readouterr().err
    took = [int(l.split()[2]) for l in err.splitlines() if l.startswith("[textgpu]")]
    fin = [int(l.split()[5]) for l in err.splitlines() if l.startswith("[textgpu]")]
    if form == "2" or ent != "FPAQ":                                # (the row form leaves TextCodec1 blocks to the host)
        assert took and sum(fin) > 0 and sum(fin) >= sum(took) // 2, err[-400:]


@pytest.mark.gpu
@pytest.mark.parametrize("chain,ent", [("UTF", "NONE"), ("TEXT+UTF", "NONE"), ("TEXT+UTF+BWT+RANK+ZRLT", "ANS0"), ("UTF+LZ", "HUFFMAN")])
def test_utf_inverse_on_the_device(ctx, chain, ent, monkeypatch, capfd):
    """Streams written by the oracle from UTF-8 text (Cyrillic with and without a BOM, more than 128 distinct code points so that
    two-byte aliases occur, English and binary blocks that UTF declines, short blocks) are decoded with the UTF inverse running on
    the device (the parity of the run of bytes >= 0x80 in front of a byte says whether it starts an alias: `k_utf_*`): same bytes as
    the input, corrupted copies get the oracle's verdict (a block the device form refuses goes to the host stage), and the trace
    shows that the device took and finished blocks."""
    monkeypatch.delenv("KZ_UTF_GPU", raising=False)
    monkeypatch.setenv("KZ_TEXT_GPU_MIN", "1")
    monkeypatch.setenv("KZ_TEXT_GPU_TRACE", "1")
    c = textgen.cases()
    rng = np.random.default_rng(23)
    cps = [0x400 + i for i in range(200)] + [0x4E00 + 7 * i for i in range(300)] + [0x1F600 + i for i in range(40)] + list(range(0x20, 0x7F))
    wide = "".join(chr(cps[int(i)]) for i in rng.integers(0, len(cps), 60000)).encode("utf-8")      # 3- and 4-byte code points, > 128 symbols
    data = (c["utf8"][:200000] + textgen.utf8(150000, 5, bom=True) + wide + c["english"][:60000] + c["random"][:30000] + c["utf8"][:3000] + c["short"])
    for bs in (65536, 1 << 20):
        ref = oracle.compress(chain, ent, bs, data, jobs=4)
        assert kz.CompressedInputStream(ctx, ref).read() == data, (chain, ent, bs)
        for _ in range(8):                                        # corrupted copies: the verdict of the oracle's decoder
            bad = bytearray(ref)
            pos = int(rng.integers(40, len(bad) - 8))
            bad[pos] ^= 1 << int(rng.integers(0, 8))
            try:
                want = oracle.decompress(bytes(bad), len(data) + 4 * bs, jobs=2)
            except Exception:
                want = None
            try:
                got = kz.CompressedInputStream(ctx, bytes(bad)).read()
            except Exception:
                got = None
            assert (got is None) == (want is None) and (got is None or got == want), (chain, ent, bs, pos)
    err = capfd.readouterr().err
    took = [int(l.split()[2]) for l in err.splitlines() if l.startswith("[utfgpu]")]
    fin = [int(l.split()[5]) for l in err.splitlines() if l.startswith("[utfgpu]")]
    assert took and sum(fin) > 0 and sum(fin) >= sum(took) // 2, err[-400:]


# ---- round 5: the TEXT forward on the device (kz_text_fwd_gpu.hip) ----
def _text_fwd_blocks(bs):
    c = textgen.cases()
    data = (c["english"][:400000] + c["utf8"][:50000] + c["random"][:40000] + c["english_crlf"][:150000] + c["xml"][:200000] + c["english_escapes"][:120000]
            + c["many_words"][:600000] + datagen.stream(2, 30000).tobytes() + c["gif_magic_text"] + c["spaces_then_text"] + c["short"] + c["min"]
            + b" " * 70 + c["english"][1000:90000] + b"the the  the The tHe the.The\r\nthe\rthe\n" * 3000 + c["english"][:5000].upper() + c["english"][5000:40000])
    blocks = [data[i:i + bs] for i in range(0, len(data), bs)]
    # blocks that are NOT text: what detectType says decides whether UTF looks at them (TextCodec.java:386-440, on the device: k_tf_pairs)
    rng = np.random.default_rng(99)
    n = min(bs, 200000)
    pick = lambda alphabet: bytes(np.frombuffer(alphabet, dtype=np.uint8)[rng.integers(0, len(alphabet), n)])
    bad_utf = bytearray(textgen.utf8(n, 3)); bad_utf[n // 2] = 0xC0                     # a byte no UTF-8 text holds
    cut_utf = bytearray(textgen.utf8(n, 4)); cut_utf[n // 3] = 0x41                     # a lead byte followed by a letter
    # UTF-8 without four-unit code points (the device UTF forward takes these: kz_utf_fwd_gpu.hip): Cyrillic + CJK, with a byte order
    # mark, cut in front (continuation bytes first) and behind (a code point reaching into the four tail bytes), few and many symbols
    cps3 = [0x400 + i for i in range(220)] + [0x4E00 + 5 * i for i in range(900)] + list(range(0x20, 0x7F)) * 3 + [0xA0, 0x7FF, 0x800, 0xFFFD]
    u3 = lambda m, k: "".join(chr(cps3[int(i)]) for i in rng.integers(0, m, k)).encode("utf-8")
    blocks += [u3(len(cps3), n)[:n], b"\xef\xbb\xbf" + u3(200, n)[:n - 3], u3(len(cps3), n)[1:n + 1], u3(300, n)[2:n - 1], u3(40, n)[:n],
               (u3(len(cps3), n // 2) + "\u0436".encode("utf-8") * (n // 4))[:n]]
    blocks += [pick(b"acgtn"), pick(b"0123456789+-*/=,.:; "), pick(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/"),
               pick(b"abc"), bytes(textgen.utf8(n, 5, bom=True)), bytes(bad_utf), bytes(cut_utf), bytes(textgen.utf8(n, 6))[1:],
               pick(bytes(range(0x80, 0xC0)) + b"  etaoin"), bytes(rng.integers(0, 256, n, dtype=np.uint8))]
    return blocks


@pytest.mark.gpu
@pytest.mark.parametrize("chain,ent", [("TEXT", "NONE"), ("TEXT+UTF", "HUFFMAN"), ("TEXT+UTF+BWT+RANK+ZRLT", "ANS0"), ("TEXT", "FPAQ")])
def test_text_forward_on_the_device(ctx, chain, ent, monkeypatch, capfd):
    """Blocks of English, CRLF text, XML, escape bytes, many invented words (the word list doubles), runs of spaces, repeated short
    words around the three-letter rule, upper-case text (the flipped-case lookups), UTF-8, binary, Magic-tagged and short blocks go
    through kz_encode_blocks with the TEXT forward running on the device (k_tf_*: TextCodec2 streams; FPAQ's TextCodec1 stays on the
    host): block streams, skip flags and lengths equal the oracle's, and the trace shows that the device took and finished blocks
    (what it declines -- not text, too close to the block's length -- goes through the host stage)."""
    monkeypatch.setenv("KZ_TEXT_FWD_GPU", "1")
    monkeypatch.setenv("KZ_TEXT_GPU_TRACE", "1")
    ctx = kz.Context(0)                                               # (its own "blockSize" entry)
    for bs in (32768, 1 << 20):
        ctx.set_block_size(bs)
        blocks = _text_fwd_blocks(bs)
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
            assert res[i].status == 0 and (res[i].bits, res[i].skipFlags, res[i].length) == (w, sf, pl), (chain, ent, bs, i, len(d))
            assert out[i, :(w + 7) // 8].tobytes() == so, (chain, ent, bs, i)
    ctx.close()
    err = capfd.readouterr().err
    took = [int(l.split()[2]) for l in err.splitlines() if l.startswith("[textfwd] took")]
    fin = [int(l.split()[5]) for l in err.splitlines() if l.startswith("[textfwd] took")]
    if ent != "FPAQ":
        assert took and sum(fin) >= sum(took) // 2, err[-400:]
    else:
        assert not took
    if "UTF" in chain and ent != "FPAQ":                              # round 6: the UTF forward of the blocks TEXT declined as UTF-8 ran on the device
        u = [l.split() for l in err.splitlines() if l.startswith("[utffwd] took")]
        assert u and sum(int(x[2]) for x in u) >= 6 and sum(int(x[5].rstrip(",")) for x in u) >= 4, err[-400:]


@pytest.mark.gpu
@pytest.mark.parametrize("chain,ent", [("TEXT+UTF+BWT+RANK+ZRLT", "ANS0"), ("TEXT", "HUFFMAN")])
def test_text_forward_on_the_device_in_kz_compress(chain, ent, monkeypatch, capfd):
    """kz_compress on a host buffer of 700 blocks: its pipeline's chunks of 256 blocks run the TEXT stage on the device (they are not
    pre-staged on the host), the last chunk of 188 blocks takes the host stages as before; the .knz is the oracle's, byte for byte."""
    monkeypatch.setenv("KZ_STREAM_CHUNK", "256")
    monkeypatch.setenv("KZ_TEXT_GPU_TRACE", "1")
    monkeypatch.delenv("KZ_TEXT_FWD_GPU", raising=False)
    bs = 32768
    c = textgen.cases()
    base = (c["english"][:900000] + c["utf8"][:120000] + c["xml"][:300000] + c["random"][:70000] + c["english_crlf"][:400000] + c["many_words"][:500000])
    data = (base * 11)[:700 * bs - 777]
    ctx = kz.Context(0)
    out = kz.CompressedOutputStream(ctx, chain, ent, bs)
    out.write(data)
    out.close()
    got = bytes(out.output)
    ctx.close()
    assert got == oracle.compress(chain, ent, bs, data, jobs=8)
    err = capfd.readouterr().err
    assert sum(int(l.split()[5]) for l in err.splitlines() if l.startswith("[textfwd] took")) > 200, err[-300:]


# ---- round 6: the device TEXT / UTF kernels at BASELINE's block size, and the staging race of kz_compress ----
_BIG = {}


def _big_blocks():
    if "b" not in _BIG:
        _BIG["b"] = textgen.big_blocks(4 << 20)
    return _BIG["b"]


@pytest.mark.gpu
@pytest.mark.parametrize("chain,ent", [("TEXT+UTF+BWT+RANK+ZRLT", "ANS0"), ("TEXT+UTF", "NONE")])
def test_text_forward_on_the_device_at_4mib_blocks(chain, ent, monkeypatch, capfd):
    """Nine blocks of 4 MiB (BASELINE's block size: TextCodec2's map starts at 2^17 entries there, TextCodec.java:1071-1081) --
    English LF / CRLF, XML, UTF-8, invented words that grow the word list past 2^17 entries, invented words that wrap it at 2^19,
    escape bytes inside text, binary, a ragged last block -- go through kz_encode_blocks with the TEXT forward forced onto the
    device: block streams, bit counts, skip flags and lengths equal oracle.encode_block(..., block_size=4 MiB), and the trace shows
    that the device took and finished blocks."""
    monkeypatch.setenv("KZ_TEXT_FWD_GPU", "1")
    monkeypatch.setenv("KZ_TEXT_FWD_GPU_MIN", "1")
    monkeypatch.setenv("KZ_TEXT_GPU_TRACE", "1")
    bs = 4 << 20
    blocks = _big_blocks()
    B = len(blocks)
    ctx = kz.Context(0)
    ctx.set_block_size(bs)
    inp = np.zeros((B, bs), dtype=np.uint8)
    lens = np.array([len(d) for d in blocks], dtype=np.int32)
    for i, d in enumerate(blocks):
        inp[i, :len(d)] = np.frombuffer(d, dtype=np.uint8)
    ostride = kz.max_block_stream_bytes(bs)
    out = np.zeros((B, ostride), dtype=np.uint8)
    res = kz.encode_blocks(ctx, chain, ent, inp, bs, lens, out, ostride)
    ctx.close()
    for i, d in enumerate(blocks):
        so, w, sf, pl = oracle.encode_block(chain, ent, d, block_size=bs)
        assert res[i].status == 0 and (res[i].bits, res[i].skipFlags, res[i].length) == (w, sf, pl), (chain, ent, i, len(d))
        assert out[i, :(w + 7) // 8].tobytes() == so, (chain, ent, i)
    err = capfd.readouterr().err
    took = [int(l.split()[2]) for l in err.splitlines() if l.startswith("[textfwd] took")]
    fin = [int(l.split()[5]) for l in err.splitlines() if l.startswith("[textfwd] took")]
    assert took and sum(took) >= 6 and sum(fin) >= 5, err[-400:]


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["", "2", "3"])
def test_text_and_utf_inverse_on_the_device_at_4mib_blocks(form, monkeypatch, capfd):
    """The oracle's level-5 .knz of the nine 4 MiB blocks is decoded with the TEXT inverse (default row form / serial walk / one-wave
    rows) and the UTF inverse running on the device: the input comes back; corrupted copies get the oracle's verdict and bytes; the
    trace shows that the device took and finished blocks."""
    if form:
        monkeypatch.setenv("KZ_TEXT_GPU", form)
    else:
        monkeypatch.delenv("KZ_TEXT_GPU", raising=False)
    monkeypatch.setenv("KZ_TEXT_GPU_MIN", "1")
    monkeypatch.delenv("KZ_UTF_GPU", raising=False)
    monkeypatch.setenv("KZ_TEXT_GPU_TRACE", "1")
    bs = 4 << 20
    data = b"".join(_big_blocks())
    chain, ent = "TEXT+UTF+BWT+RANK+ZRLT", "ANS0"
    if "knz" not in _BIG:
        _BIG["knz"] = oracle.compress(chain, ent, bs, data, jobs=8)
    ref = _BIG["knz"]
    ctx = kz.Context(0)
    assert kz.CompressedInputStream(ctx, ref).read() == data
    rng = np.random.default_rng(600 + len(form))
    for _ in range(2):
        bad = bytearray(ref)
        pos = int(rng.integers(40, len(bad) - 8))
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            want = oracle.decompress(bytes(bad), len(data) + 4 * bs, jobs=8)
        except Exception:
            want = None
        try:
            got = kz.CompressedInputStream(ctx, bytes(bad)).read()
        except Exception:
            got = None
        assert (got is None) == (want is None) and (got is None or got == want), (form, pos)
    ctx.close()
    err = capfd.readouterr().err
    took = [int(l.split()[2]) for l in err.splitlines() if l.startswith("[textgpu]")]
    fin = [int(l.split()[5]) for l in err.splitlines() if l.startswith("[textgpu]")]
    assert took and sum(fin) >= 5, err[-400:]
    utook = [int(l.split()[2]) for l in err.splitlines() if l.startswith("[utfgpu]")]
    ufin = [int(l.split()[5]) for l in err.splitlines() if l.startswith("[utfgpu]")]
    assert utook and sum(ufin) >= 1, err[-400:]


@pytest.mark.gpu
@pytest.mark.parametrize("tool", ["text_fwd_gpu_fuzz", "text_gpu_fuzz"])
def test_device_text_fuzz_with_a_4mib_case(tool):
    import subprocess
    import sys
    """the two device-TEXT differential fuzzers in their full-size mode (block sizes up to 4 MiB) for a bounded time, fixed seed"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", tool + ".py"), "40", "606", "big4"], capture_output=True, text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0 and " 0 mismatches" in r.stdout and "4MiB cases: 0" not in r.stdout, tail


@pytest.mark.gpu
def test_kz_compress_device_text_chunks_do_not_share_staging_with_the_upload_thread(monkeypatch):
    """kz_compress's pipeline pre-stages chunk k+1 on its upload thread (pinned hsOut[k & 1]) while the main thread
    encodes chunk k; a device-TEXT chunk used to run its host passes in the same pinned buffers.  Three chunks of 256 / 256 / 61
    blocks whose text blocks are non-ASCII UTF-8 (TEXT declines them on the device, UTF applies on the host: the host passes write
    real output), interleaved with English: the .knz must be the oracle's on every one of five runs."""
    monkeypatch.setenv("KZ_STREAM_CHUNK", "256")
    monkeypatch.delenv("KZ_TEXT_FWD_GPU", raising=False)
    bs = 32768
    rng = np.random.default_rng(77)
    cps = [0x400 + i for i in range(200)] + [0x4E00 + 7 * i for i in range(300)] + [32] * 60
    wide = "".join(chr(cps[int(i)]) for i in rng.integers(0, len(cps), 150000)).encode("utf-8")
    c = textgen.cases()
    base = wide[:bs * 5] + c["english"][:bs * 2] + textgen.utf8(bs * 3, 5) + c["xml"][:bs]
    data = (base * 60)[:573 * bs - 333]
    chain, ent = "TEXT+UTF+BWT+RANK+ZRLT", "ANS0"
    want = oracle.compress(chain, ent, bs, data, jobs=8)
    ctx = kz.Context(0)
    for run in range(5):
        out = kz.CompressedOutputStream(ctx, chain, ent, bs)
        out.write(data)
        out.close()
        assert bytes(out.output) == want, run
    ctx.close()
