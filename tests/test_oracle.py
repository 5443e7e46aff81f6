"""CPU tests: the oracle against the hand-derived known answers (the reference holds
no byte-exact goldens -> PARITY UNPINNED beyond these), round-trip identity on the reference's own test
inputs, and the committed self-generated fixtures."""
import ctypes
import json
import os

import numpy as np
import pytest

import datagen
import oracle
import refinputs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_bwt_mississippi_javadoc_known_answer():
    # K/transform/BWT.java:45-50: BWT("mississippi") = "ipssmpissii", primary index 5
    L = oracle.lib()
    src = b"mississippi"
    out = ctypes.create_string_buffer(len(src))
    pr = (ctypes.c_int32 * 8)()
    assert L.kzo_bwt_forward_raw(src, len(src), out, pr) == 1
    assert out.raw == b"ipssmpissii"
    assert pr[0] == 5
    ok, enc = oracle.transform_forward("BWT", src)
    assert ok and enc == bytes([0x00, 0x04]) + b"ipssmpissii"      # mode 0, primary-1 = 4 (BWTBlockCodec.java:110-126)
    ok, back = oracle.transform_inverse("BWT", enc, 64)
    assert ok and back == src


def test_zrlt_hand_vectors():
    ok, out = oracle.transform_forward("ZRLT", bytes([0, 0, 0, 0, 0, 5, 0xFE, 9]))
    assert ok and out == bytes([0x01, 0x00, 0x06, 0xFF, 0x00, 0x0A])
    ok, back = oracle.transform_inverse("ZRLT", out, 64)
    assert ok and back == bytes([0, 0, 0, 0, 0, 5, 0xFE, 9])
    ok, _ = oracle.transform_forward("ZRLT", bytes([0, 0, 0, 5, 0xFE, 0]))    # ZRLT.java:94 strict check trips
    assert not ok


def test_sbrt_mtf_hand_vector():
    ok, out = oracle.transform_forward("MTFT", b"abca")
    assert ok and list(out) == [97, 98, 99, 2]


def test_varint_and_alphabet_hand_vectors():
    L = oracle.lib()
    s = oracle._Obs()
    L.kzo_obs_init(ctypes.byref(s), 64)
    L.kzo_write_varint(ctypes.byref(s), 300)
    assert ctypes.string_at(s.buf, 2) == bytes([0xAC, 0x02]) and s.nbits == 16
    L.kzo_obs_free(ctypes.byref(s))
    L.kzo_obs_init(ctypes.byref(s), 64)
    alpha = (ctypes.c_int * 256)(2, 3)
    L.kzo_encode_alphabet(ctypes.byref(s), alpha, 2)
    # bit 1 (partial), 00000 (lastMask), 0x0C  -> 1 00000 00001100
    assert s.nbits == 14 and ctypes.string_at(s.buf, 2) == bytes([0b10000000, 0b00110000])
    L.kzo_obs_free(ctypes.byref(s))


def test_suffix_array_matches_naive():
    L = oracle.lib()
    rng = np.random.default_rng(3)
    for _ in range(150):
        n = int(rng.integers(2, 300))
        k = int(rng.choice([1, 2, 3, 4, 256]))
        b = bytes(rng.integers(0, k, n, dtype=np.uint8))
        sa = np.zeros(n, dtype=np.int32)
        L.kzo_suffix_array(b, sa.ctypes.data, n)
        assert list(sa) == sorted(range(n), key=lambda i: b[i:])


@pytest.mark.parametrize("name", ["RANK", "MTFT", "ZRLT", "BWT", "SRT", "LZ", "LZX"])
def test_transform_roundtrip_reference_inputs(name):
    # T/test/TestTransforms.java:172-337: forward then inverse equals input; "false" = no compression is accepted
    for data in refinputs.transform_inputs() + refinputs.bwt_inputs():
        ok, enc = oracle.transform_forward(name, data)
        if not ok:
            continue
        ok2, back = oracle.transform_inverse(name, enc, len(data) + 1024)
        assert ok2 and back == data


def test_entropy_roundtrip_reference_inputs():
    # T/test/TestEntropyCodec.java:203-290
    for ent in ("ANS0", "HUFFMAN", "FPAQ", "NONE"):
        for data in refinputs.entropy_inputs() + refinputs.edge_inputs():
            bits, nb = oracle.entropy_encode(ent, data)
            r, back, used = oracle.entropy_decode(ent, bits, nb, len(data))
            assert r == len(data) and back == data and (used == nb or len(data) == 0)


@pytest.mark.parametrize("chain,ent", [("BWT+RANK+ZRLT", "ANS0"), ("BWT+MTFT+ZRLT", "ANS0"), ("ZRLT", "NONE"), ("NONE", "ANS0"), ("BWT+RANK+ZRLT", "HUFFMAN"),
                                       ("BWT+SRT+ZRLT", "FPAQ"), ("LZ", "HUFFMAN"), ("LZ", "ANS0"), ("LZX", "NONE")])
def test_stream_roundtrip(chain, ent):
    rng = np.random.default_rng(5)
    data = bytes(np.minimum(rng.geometric(0.05, 200000) - 1, 255).astype(np.uint8)) + b"abc" * 1000 + bytes(3000)
    knz = oracle.compress(chain, ent, 65536, data, jobs=4)
    assert knz[:4] == b"KANZ"
    assert oracle.decompress(knz, len(data), jobs=4) == data
    assert oracle.compress(chain, ent, 65536, data, jobs=1) == knz          # job count never changes the bytes


def test_golden_fixtures():
    """Self-generated fixtures (NOT reference-generated: no JVM here). They freeze the oracle so that an
    accidental change is caught; promote to true goldens when `java -jar kanzi.jar` output is available."""
    with open(os.path.join(GOLD, "manifest.json")) as f:
        man = json.load(f)
    assert man["provenance"].startswith("self-generated")
    for e in man["entries"]:
        inp = open(os.path.join(GOLD, e["input"]), "rb").read()
        exp = open(os.path.join(GOLD, e["output"]), "rb").read()
        got = oracle.compress(e["chain"], e["entropy"], e["blockSize"], inp, jobs=2)
        assert got == exp, e
        assert oracle.decompress(exp, len(inp), jobs=2) == inp


def test_xxhash32_is_standard_xxh32_and_xxhash64_quirks():
    """External pin: K/util/hash/XXHash32.java is the standard XXH32 (checked against the `xxhash` package).
    XXHash64 deviates from XXH64 (32-bit rotate amounts in the lane merge, sign-extended 4-byte tail): it equals
    the standard only where neither quirk triggers (< 32 bytes and no 4-byte tail with the top bit set)."""
    xxhash = pytest.importorskip("xxhash")
    rng = np.random.default_rng(9)
    for n in (0, 1, 3, 4, 15, 16, 17, 31, 32, 33, 100, 1000, 65536):
        d = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert oracle.xxhash32(d) == xxhash.xxh32(d, seed=0x4B414E5A).intdigest()
    for n in (0, 1, 3, 8, 9, 16, 24):
        d = bytes(rng.integers(0, 128, n, dtype=np.uint8))
        assert oracle.xxhash64(d) == xxhash.xxh64(d, seed=0x4B414E5A).intdigest()
    d = bytes(rng.integers(0, 256, 1000, dtype=np.uint8))
    assert oracle.xxhash64(d) != xxhash.xxh64(d, seed=0x4B414E5A).intdigest()


def test_checksummed_stream_roundtrip():
    data = bytes(np.random.default_rng(2).integers(0, 7, 100000, dtype=np.uint8))
    for bits in (32, 64):
        knz = oracle.compress("BWT+RANK+ZRLT", "ANS0", 32768, data, jobs=2, checksum=bits)
        assert oracle.decompress(knz, len(data), jobs=2) == data


def test_stream_header_faults_report_reference_codes():
    """Each stream-header fault is reported with the code, and in the order, of CompressedInputStream.readHeader
    (CompressedInputStream.java:363-478; codes from Error.java:24-43)."""
    data = bytes(range(256)) * 20
    good = oracle.compress(["BWT", "RANK", "ZRLT"], "ANS0", 4096, data)
    assert oracle.decompress(good, len(data)) == data
    for what, bad, code in refinputs.header_faults(good):
        with pytest.raises(oracle.OracleError) as e:
            oracle.decompress(bad, len(data))
        assert e.value.code == code, what


def test_block_faults_report_first_failing_block():
    """Block-level faults: truncation inside a block is ERR_READ_FILE (11) once the whole blocks before it decoded; a
    flipped header byte is ERR_CRC_CHECK (19); an oversized encoded length is ERR_BLOCK_SIZE (2)
    (CompressedInputStream.java:1027-1028,1088-1091,1151-1165)."""
    data = bytes(range(256)) * 40
    good = oracle.compress(["BWT", "RANK", "ZRLT"], "ANS0", 4096, data)
    with pytest.raises(oracle.OracleError) as e:
        oracle.decompress(good[:len(good) - 40], len(data))
    assert e.value.code == 11
    bad = bytearray(good)
    bad[25] ^= 0x40                                            # inside the first block's header bytes
    with pytest.raises(oracle.OracleError) as e:
        oracle.decompress(bytes(bad), len(data))
    assert e.value.code in (19, 2, 11, 13)


def test_oracle_survives_corrupted_input():
    """The oracle is the judge of the corrupted-input parity tests: it must stay inside its buffers on any input (a
    Java ArrayIndexOutOfBounds is restated as a failure return) and still round-trip the clean buffer afterwards."""
    import datagen
    rng = np.random.default_rng(9)
    n = 12000
    cap = n + max(512, n >> 4)
    alias = [d for _, d in refinputs.alias_inputs()]
    for name in ["BWT", "SRT", "ZRLT", "RANK", "MTFT", "LZ", "LZX", "MM", "PACK"]:
        for src_kind in (0, 2, 3):
            pre = datagen.block(src_kind, n).tobytes() if name != "MM" else refinputs.multimedia_like(src_kind, n)
            if name == "PACK":
                pre = alias[(1, 9, 17)[src_kind if src_kind < 3 else 2]][:n]
            if name in ("SRT", "RANK", "MTFT", "ZRLT"):
                pre = oracle.transform_forward("BWT", pre)[1]
            ok, good = oracle.transform_forward(name, pre)
            if not ok:
                continue
            for trial in range(16):
                oracle.transform_inverse(name, refinputs.corrupt(rng, good, trial % 8), cap)
            ok, back = oracle.transform_inverse(name, good, cap)
            assert ok and back == pre
    for ent in ["ANS0", "HUFFMAN", "FPAQ"]:
        data = datagen.block(3, 30000).tobytes()
        good, nbits = oracle.entropy_encode(ent, data)
        for trial in range(24):
            bad = refinputs.corrupt(rng, good, trial % 8)
            oracle.entropy_decode(ent, bad, min(nbits, len(bad) * 8), len(data))
        r, back, _ = oracle.entropy_decode(ent, good, nbits, len(data))
        assert r == len(data) and back == data
    data = datagen.stream(4, 16384).tobytes()
    good = oracle.compress("BWT+RANK+ZRLT", "ANS0", 16384, data, checksum=32)
    for trial in range(32):
        bad = bytearray(refinputs.corrupt(rng, good[24:], trial % 8))
        try:
            oracle.decompress(good[:24] + bytes(bad), len(data))
        except oracle.OracleError as e:
            assert e.code in (2, 11, 13, 19)
    assert oracle.decompress(good, len(data)) == data


def test_log_table_and_entropy_known_answers():
    """Global.LOG2_4096 is generated (round(4096*log2 x)), not transcribed: pin it on entries read off
    Global.java:104-127, and log2_1024 / computeFirstOrderEntropy1024 on values worked by hand from :222-235, :440-456."""
    L = oracle.lib()
    known = {0: 0, 1: 0, 2: 4096, 3: 6492, 5: 9511, 7: 11499, 10: 13607, 17: 16742, 100: 27213, 129: 28718, 200: 31309,
             255: 32745, 256: 32768}
    for x, v in known.items():
        assert L.kzo_log2_4096(x) == v
    assert L.kzo_log2_1024(1) == 0 and L.kzo_log2_1024(2) == 1024 and L.kzo_log2_1024(3) == (6492 + 2) >> 2
    assert L.kzo_log2_1024(4096) == 12 * 1024                      # exact power of two
    assert L.kzo_log2_1024(1000) == 2 * 1024 + ((L.kzo_log2_4096(1000 >> 2) + 2) >> 2)
    h = (ctypes.c_int * 256)()
    h[0] = h[1] = 512                                               # two equiprobable symbols: 1 bit; the scale is 1024 per 8 bits
    assert L.kzo_entropy1024(1024, h) == 128                        # 2 * ((512 * (10240 - 9216)) >> 3) / 1024
    for i in range(256):
        h[i] = 4                                                    # uniform: 8 bits
    assert L.kzo_entropy1024(1024, h) == 1024


def test_magic_and_block_data_type():
    """Magic.getType quirks: JPEG returns the key itself (only ...E0 counts as compressed), 3-byte and 2-byte magics,
    PNM needs a whitespace third byte; the writer's tag (CompressedOutputStream.java:795-804)."""
    L = oracle.lib()
    L.kzo_magic_type.restype = ctypes.c_int32
    t = lambda b: L.kzo_magic_type(ctypes.c_char_p(bytes(b) + b"\0" * 4))
    assert t(b"\xFF\xD8\xFF\xE0") == ctypes.c_int32(0xFFD8FFE0).value and t(b"\xFF\xD8\xFF\xE1") == ctypes.c_int32(0xFFD8FFE1).value
    assert t(b"BZh9") == 0x425A68 and t(b"ID3\x03") == 0x494433 and t(b"\x1F\x8B\x08\x00") == 0x1F8B
    assert t(b"BM\x00\x00") == 0x424D and t(b"P5\n1") == 0x5035 and t(b"P5x1") == 0 and t(b"abcd") == 0
    dt = lambda b: L.kzo_block_data_type(ctypes.c_char_p(bytes(b) + b"\0" * 4), len(b))
    assert dt(b"\xFF\xD8\xFF\xE0....") == oracle.DT["BIN"] and dt(b"\xFF\xD8\xFF\xE1....") == oracle.DT["UNDEFINED"]
    assert dt(b"RIFF....") == oracle.DT["MULTIMEDIA"] and dt(b"\x7FELF....") == oracle.DT["EXE"] and dt(b"MZ..") == oracle.DT["EXE"]
    assert dt(b"PK\x03\x04") == oracle.DT["BIN"] and dt(b"BM") == oracle.DT["UNDEFINED"]


@pytest.mark.parametrize("kind", [0, 1, 2, 3, 4])
def test_mm_roundtrip_and_choices(kind):
    """FSDCodec on multimedia-like inputs: applied, header (mode, step) as the sampling rules dictate, exact round
    trip, context entry set to MULTIMEDIA; declined (with the detected simple type) where coding does not pay."""
    want = {0: (1, 2), 1: (0, 3), 2: (0, 1), 3: (0, 4), 4: (1, 4)}[kind]
    for n in (1024, 5000, 65536, 300001):
        data = refinputs.multimedia_like(kind, n)
        ok, out, dt = oracle.transform_forward("MM", data, data_type=oracle.DT["UNDEFINED"])
        assert ok and dt == oracle.DT["MULTIMEDIA"], (kind, n)
        if n >= 5000:
            assert (out[0], out[1]) == want, (kind, n, out[0], out[1])
        ok2, back = oracle.transform_inverse("MM", out, len(data) + 64)
        assert ok2 and back == data
    rnd = bytes(np.random.default_rng(0).integers(0, 256, 50000, dtype=np.uint8))
    assert oracle.transform_forward("MM", rnd, data_type=0)[0::2] == (False, oracle.DT["BIN"])
    txt = (b"the quick brown fox jumps over the lazy dog " * 2000)[:60000]
    assert oracle.transform_forward("MM", txt, data_type=0)[0::2] == (False, oracle.DT["UNDEFINED"])
    dna = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(3).integers(0, 4, 80000)])
    assert oracle.transform_forward("MM", dna, data_type=0)[0::2] == (False, oracle.DT["DNA"])
    png = b"\x89PNG" + refinputs.multimedia_like(0, 60000)[4:]
    assert oracle.transform_forward("MM", png, data_type=0)[0::2] == (False, 0)            # magic not a candidate
    assert oracle.transform_forward("MM", refinputs.multimedia_like(0, 60000), data_type=oracle.DT["EXE"])[0] is False
    assert oracle.transform_forward("MM", refinputs.multimedia_like(0, 1023), data_type=0)[0] is False


def test_mm_in_a_stream():
    """MM+LZX & HUFFMAN (the tail of the reference's level 3 chain) as a whole stream, mixed applicable / declined blocks."""
    data = refinputs.multimedia_like(0, 200000) + refinputs.multimedia_like(3, 150000) + bytes(np.random.default_rng(5).integers(0, 256, 70000, dtype=np.uint8)) + (b"plain text block " * 5000)
    z = oracle.compress("MM+LZX", "HUFFMAN", 65536, data, checksum=32)
    assert oracle.decompress(z, len(data)) == data
    assert len(z) < len(oracle.compress("LZX", "HUFFMAN", 65536, data, checksum=32))


@pytest.mark.parametrize("name", ["PACK", "DNA"])
def test_alias_codec_roundtrip_and_branches(name):
    """AliasCodec (PACK; DNA = the same codec restricted to DNA-looking blocks, TransformFactory.java:341-343): every
    branch round-trips, the header says which branch ran, the context entry gets Global.detectSimpleType's verdict."""
    seen = set()
    for label, data in refinputs.alias_inputs():
        ok, out, dt = oracle.transform_forward(name, data, data_type=oracle.DT["UNDEFINED"])
        if not ok:
            seen.add("declined")
            continue
        assert len(out) < len(data), label                          # applied only when it shrinks (:276)
        n0 = out[0]
        seen.add("one" if n0 == 255 else "2bit" if n0 >= 252 else "4bit" if n0 >= 240 else "digram")
        ok2, back = oracle.transform_inverse(name, out, len(data) + 64)
        assert ok2 and back == data, label
        if name == "DNA":
            assert dt == oracle.DT["DNA"], label
    assert seen >= ({"declined", "2bit"} if name == "DNA" else {"declined", "one", "2bit", "4bit", "digram"})
    txt = dict(refinputs.alias_inputs())["text+0"]
    for tag in ("MULTIMEDIA", "UTF8", "EXE", "BIN"):                # :103-109
        assert oracle.transform_forward("PACK", txt, data_type=oracle.DT[tag])[0] is False
    assert oracle.transform_forward("DNA", txt, data_type=oracle.DT["TEXT"])[0] is False      # :111-113
    assert oracle.transform_forward("PACK", txt, data_type=oracle.DT["TEXT"])[0] is True


def test_alias_codec_known_answer():
    """Hand-worked from AliasCodec.java:143-199: 1024 bytes of "ab" repeated -> header 254 (two symbols present),
    'a','b', count & 3 = 0, then 0b00010001 for every four bytes."""
    ok, out, _ = oracle.transform_forward("PACK", b"ab" * 512, data_type=0)
    assert ok and out == bytes([254, ord("a"), ord("b"), 0]) + bytes([0x11]) * 256
    ok, out, _ = oracle.transform_forward("PACK", b"q" * 1500, data_type=0)
    assert ok and out == bytes([255, ord("q")]) + (1500).to_bytes(4, "little")


def test_fpaq_known_answers_from_a_python_model_of_the_reference():
    """FPAQ on blocks of a few bytes against tests/katmodels.fpaq_encode, a pure-Python model written from
    K/entropy/FPAQEncoder.java:84-97,128-238 (not from oracle/kzo_fpaq_srt.c).  First step by hand, block 00 00: low = 0,
    high = 2^56 - 1, p[1] = 32768: split = ((2^56 - 1) >>> 8) * 32768 >>> 8 = 0x7FFFFFFFFFFF80, the bit is 0 so
    low = split + 1 = 0x7FFFFFFFFFFF81 and p[1] = 32768 - 512 = 32256; the top 32 bits of low and high still differ, so nothing is
    flushed.  After the 16 bits nothing has been flushed: the chunk is empty (varint 00) and dispose() writes low | 0xFFFFFF in
    56 bits."""
    import katmodels
    assert katmodels.fpaq_encode(b"\x00\x00").hex() == "00fffede31ffffff"
    for data in (b"\x00\x00", b"AB", b"\xff\x00\x80\x7f", bytes(range(40)), b"a" * 100, bytes([0x0F, 0xF0] * 9)):
        model = katmodels.fpaq_encode(data)
        enc, nbits = oracle.entropy_encode("FPAQ", data)
        assert nbits == 8 * len(model) and enc[:len(model)] == model, data[:8]


def test_ans0_known_answers_from_a_python_model_of_the_reference():
    """ANS0 against tests/katmodels.ans0_encode, a pure-Python model written from K/entropy/ANSRangeEncoder.java:263-407,473-496 and
    K/entropy/EntropyUtils.java:38-75,141-250 (not from oracle/kzo_ans.c): a 33-byte chunk (the smallest that is not stored, :267-270),
    chunks with a partial and with the full alphabet, a one-symbol chunk (header only, :292-295), a two-chunk block, skewed
    frequencies that take normalizeFrequencies' fast and slow paths."""
    import katmodels
    rng = np.random.default_rng(2)
    cases = [bytes(range(33)), b"abracadabra" * 3, bytes(rng.integers(0, 256, 100, dtype=np.uint8)), b"a" * 40, b"ab" * 20 + b"c",
             bytes(np.minimum(rng.geometric(0.2, 700) - 1, 255).astype(np.uint8)), bytes(rng.integers(0, 4, 20000, dtype=np.uint8)),
             bytes(rng.integers(0, 256, 5000, dtype=np.uint8)), b"x" * 10,
             bytes(np.minimum(rng.geometric(0.01, 3000) - 1, 255).astype(np.uint8)), bytes([0] * 3000 + [1, 2, 3] * 11)]
    for data in cases:
        model, nbits = katmodels.ans0_encode(data)
        enc, obits = oracle.entropy_encode("ANS0", data)
        assert obits == nbits and enc[:len(model)] == model, (len(data), nbits, obits)
    # the 33-byte ramp, first bits by hand: logRange - 8 = 4 in 3 bits (100), partial alphabet (1), last mask byte 33 >> 3 = 4 in 5 bits
    # (00100), then the masks FF FF FF FF 01
    model, _ = katmodels.ans0_encode(bytes(range(33)))
    assert model[:6].hex() == "927fffffff80" and (model[6] & 0x80) == 0x80      # those 49 bits; the frequency header follows


def test_huffman_known_answers_from_a_python_model_of_the_reference():
    """Huffman against tests/katmodels.huffman_encode (written from K/entropy/HuffmanEncoder.java, its code lengths from a
    textbook construction instead of the in-place passes): inputs whose optimal lengths are unique -- dyadic weights, a flat
    alphabet of 256, 16 and 2 symbols, a one-symbol chunk (header only), a block below 32 bytes (stored), a tail behind the
    four fragments -- plus the first bits of one case by hand."""
    import katmodels
    dyadic = bytes([97] * 32 + [98] * 16 + [99] * 8 + [100] * 4 + [101] * 2 + [102, 103])
    rng = np.random.default_rng(8)
    perm = bytes(rng.permutation(np.frombuffer(dyadic, dtype=np.uint8)))
    cases = [dyadic, perm, perm + b"ab", bytes(range(256)) * 3, bytes(range(16)) * 9 + bytes(range(3)), b"ab" * 40, b"q" * 77, b"short block",
             bytes(rng.permutation(np.repeat(np.arange(8, dtype=np.uint8), [64, 64, 32, 32, 16, 16, 16, 16])))]
    for data in cases:
        model, nbits = katmodels.huffman_encode(data)
        enc, obits = oracle.entropy_encode("HUFFMAN", data)
        assert obits == nbits and enc[:len(model)] == model, (len(data), nbits, obits)
    # by hand, the dyadic block: partial alphabet (1), last mask byte 103 >> 3 = 12 (01100), masks 0 x 12 then 0xFE (symbols 97..103 =
    # bits 1..7 of byte 12): 1 01100 | 96 zero bits | 11111110; lengths 1,2,3,4,5,6,6 as signed Exp-Golomb deltas from 2:
    # -1 = 0101, +1 = 0100 five times, 0 = 1; 16-byte fragments: "a" x 16 = 16 bits, "a" x 16 = 16 bits, "b" x 16 = 32 bits, c x 8, d x 4,
    # e x 2, f, g = 24 + 16 + 10 + 12 = 62 bits -> varints 10 10 20 3E
    model, nbits = katmodels.huffman_encode(dyadic)
    want = "1" + "01100" + "0" * 96 + "11111110" + "0101" + "0100" * 5 + "1" + "".join(format(v, "08b") for v in (16, 16, 32, 62))
    want += "0" * 32 + "10" * 16 + "110" * 8 + "1110" * 4 + "11110" * 2 + "111110" + "111111"
    assert nbits == len(want)
    assert model == int(want + "0" * (-len(want) % 8), 2).to_bytes((len(want) + 7) // 8, "big")


def test_lz_frames_are_read_back_by_a_python_model_of_the_reference_decoder():
    """LZ / LZX forward output of the oracle, decoded by tests/katmodels.lz_decode (written from K/transform/LZCodec.java
    inverseV6 :617-756, not from oracle/kzo_lz.c), and one frame taken apart by hand."""
    import katmodels
    rng = np.random.default_rng(3)
    cases = [b"abc" * 420, datagen.block(0, 50000).tobytes(), datagen.block(2, 70000).tobytes(), bytes(rng.integers(0, 4, 30000, dtype=np.uint8)),
             bytes(1000) + b"xyz" * 500, b"0123456789" * 8 + b"ABCD", datagen.block(4, 300000).tobytes(), bytes(rng.integers(0, 256, 2000, dtype=np.uint8)) * 40]
    for name in ("LZ", "LZX"):
        for d in cases:
            ok, enc = oracle.transform_forward(name, d)
            if ok:
                assert katmodels.lz_decode(enc, len(d)) == d, (name, len(d))
    # "abc" x 420 through LZ, by the frame rules: token stream at 40 = 13 + 27 literal bytes, 2 token bytes, 1 match-index byte,
    # flag 04 = 64 KiB window, minMatch 4.  Token 6f: 3 literals "abc", then a match with a 1-byte distance (03) and length
    # 7 + 4 + readLength(fe 03 c9 = 254 + 0x03c9) = 1234; token e0: 7 + readLength(10) = 23 literals, the tail the encoder never
    # matches; 3 + 1234 + 23 = 1260
    ok, enc = oracle.transform_forward("LZ", b"abc" * 420)
    assert ok and enc.hex() == "28000000" "02000000" "01000000" "04" + "616263" + "10" + (b"bca" * 8)[:23].hex() + "6fe0" + "03" + "fe03c9"



# ---- round 3: the stages whose output is a CHOICE, pinned by pure-Python models written from the Java (tests/katmodels.py) ----
import functools


@functools.lru_cache(maxsize=1)
def _model_inputs_cached():
    return tuple(_model_inputs_uncached())


def _model_inputs():
    return list(_model_inputs_cached())


def _model_inputs_uncached():
    """the reference's own transform generators (T/test/TestTransforms.java:183-254, capped at 40 000 bytes for the Python loops)
    + 64 KiB of every synthetic class + text-like cases"""
    import textgen
    out = [("ref%d" % i, bytes(a[:40000])) for i, a in enumerate(refinputs.transform_inputs())]
    out += [("class%d" % c, datagen.block(c, 65536, c).tobytes()) for c in range(5)]
    tc = textgen.cases()
    out += [(k, tc[k][:50000]) for k in ("english", "xml", "utf8", "dna", "digits")]
    return out


def _sbrt_oracle(mode, data):
    a = np.frombuffer(bytes(data), dtype=np.uint8)
    out = np.zeros(max(len(a), 1), dtype=np.uint8)
    fn = oracle.lib().kzo_sbrt_forward
    fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    assert fn(mode, a.ctypes.data if len(a) else out.ctypes.data, len(a), out.ctypes.data)
    return out[:len(a)].tobytes()


def test_lz_lzx_parse_known_answers_from_a_python_model(built):
    """LZXCodec.forward (LZCodec.java:299-597) for LZ and LZX: hash, the two repeat distances, lazy +1 / +2 probes, the skip
    acceleration srcInc >> 6, backward extension, MAX_MATCH clamp, the fixed token buffer.  Any valid parse round-trips, so only a
    second implementation pins the decisions."""
    import katmodels
    differ = 0
    for name, d in _model_inputs():
        for tname, extra in (("LZ", False), ("LZX", True)):
            for dt in ("UNDEFINED", "DNA"):
                if dt == "DNA" and not name.startswith(("dna", "class0", "ref20")):
                    continue
                try:
                    ok_m, enc_m = katmodels.lz_forward(d, extra, dt)
                    threw_m = False
                except katmodels.JavaException:
                    threw_m = True
                try:
                    ok_o, enc_o, _ = oracle.transform_forward(tname, d, data_type=oracle.DT[dt])
                    threw_o = False
                except oracle.TransformThrows:
                    threw_o = True
                assert threw_m == threw_o, (name, tname, dt)
                if threw_m or len(d) == 0:
                    continue
                assert ok_m == ok_o, (name, tname, dt, ok_m, ok_o)
                # a declined block's bytes are not part of the contract (Sequence copies the input through): compare applied ones
                if ok_m:
                    assert enc_m == enc_o, (name, tname, dt, len(enc_m), len(enc_o))
                    differ += 1
    assert differ > 40
    assert katmodels.lz_forward(b"ab" * 600, False, "SMALL_ALPHABET") == (False, b"")


def test_srt_forward_known_answers_from_a_python_model(built):
    """SRT.forward (SRT.java:73-168), preprocess (shell sort :266-302) and encodeHeader (:312-325)"""
    import katmodels
    for name, d in _model_inputs():
        if not d:
            continue
        ok, enc = oracle.transform_forward("SRT", d)
        assert ok and enc == katmodels.srt_forward(d), name
    # worked by hand: "abracadabra": first appearances a b r c d = the initial list; frequencies a5 b2 r2 c1 d1 -> bucket order
    # a b r c d (freq desc, symbol asc) at offsets 0 5 7 9 10.  Move-to-front ranks in text order: a0 b1 r2 a2 c3 a1 d4 a1 b4 r4 a2
    # (lists: abrcd, bar.., rba.., arb.., carbd, acrbd, dacrb, adcrb, badcr, rbadc), gathered per symbol
    enc = katmodels.srt_forward(b"abracadabra")
    freqs = [0] * 256
    for c in b"abracadabra":
        freqs[c] += 1
    assert enc[:256] == bytes(freqs)
    assert enc[256:] == bytes([0, 2, 1, 1, 2]) + bytes([1, 4]) + bytes([2, 4]) + bytes([3]) + bytes([4])


def test_sbrt_rank_and_timestamp_known_answers_from_a_python_model(built):
    """SBRT.forward (SBRT.java:87-151) in all three modes (MTF 1, RANK 2, TIMESTAMP 3)"""
    import katmodels
    for name, d in _model_inputs():
        for mode in (1, 2, 3):
            assert _sbrt_oracle(mode, d) == katmodels.sbrt_forward(d, mode), (name, mode)
    # RANK by hand on "abab" + "c": ranks of a, b = 97, 98; then a sits at rank 1 (b moved in front), b at 1, ...
    assert katmodels.sbrt_forward(b"ababc", 2) == bytes([97, 98, 1, 1, 99])
    assert katmodels.sbrt_forward(b"aab", 1) == bytes([97, 0, 98])


def test_alias_forward_known_answers_from_a_python_model(built):
    """AliasCodec.forward (AliasCodec.java:78-278): packing for <= 16 symbols, digram aliases picked in TreeSet order, the savings
    test, the dataType rules and what is stored back"""
    import katmodels
    rng = np.random.default_rng(11)
    extra = [("four", bytes(rng.choice(np.frombuffer(b"acgt", dtype=np.uint8), 4099))), ("one", b"z" * 3000),
             ("sixteen", bytes(rng.integers(64, 80, 5001, dtype=np.uint8))), ("nodigram", bytes(rng.integers(0, 200, 30000, dtype=np.uint8)))]
    applied = 0
    for name, d in _model_inputs() + extra + [(k, bytes(v)[:60000]) for k, v in refinputs.alias_inputs()]:
        for tname, only_dna in (("PACK", False), ("DNA", True)):
            for dt in ("UNDEFINED", "TEXT", "BIN", "DNA"):
                ok_m, enc_m, left_m = katmodels.alias_forward(d, dt, only_dna)
                ok_o, enc_o, left_o = oracle.transform_forward(tname, d, data_type=oracle.DT[dt])
                assert (ok_m, oracle.DT[left_m]) == (ok_o, left_o), (name, tname, dt, ok_m, ok_o, left_m, left_o)
                if ok_m and len(d):
                    assert enc_m == enc_o, (name, tname, dt)
                    applied += 1
    assert applied > 30


def test_utf_forward_known_answers_from_a_python_model(built):
    """UTFCodec.forward (UTFCodec.java:68-218) with validate (:313-434) and pack (:437-466)"""
    import katmodels
    import textgen
    tc = textgen.cases()
    cases = _model_inputs() + [(k, tc[k]) for k in ("utf8_bom", "utf8_cut", "english_escapes", "random")]
    applied = 0
    for name, d in cases:
        for dt in ("UNDEFINED", "UTF8", "TEXT"):
            ok_m, enc_m, left_m = katmodels.utf_forward(d, dt)
            ok_o, enc_o, left_o = oracle.transform_forward("UTF", d, data_type=oracle.DT[dt])
            assert (ok_m, oracle.DT[left_m]) == (ok_o, left_o), (name, dt, ok_m, ok_o, left_m, left_o)
            if ok_m and len(d):
                assert enc_m == enc_o, (name, dt)
                applied += 1
    assert applied >= 4


def test_mm_forward_known_answers_from_a_python_model(built):
    """FSDCodec.forward (FSDCodec.java:63-246): the sampled entropies (Global.computeFirstOrderEntropy1024, log2_1024), the step and
    the coding it picks, escapes, the final sanity check, Magic.getType and the dataType it leaves behind"""
    import katmodels
    cases = [("mm%d" % k, refinputs.multimedia_like(k, 60000 + 7 * k, seed=k + 1)) for k in range(5)]
    cases += [("mm%d_short" % k, refinputs.multimedia_like(k, 1024 + k, seed=9)) for k in range(5)]
    cases += [("bmp", b"BM" + refinputs.multimedia_like(1, 30000, seed=3)), ("riff", b"RIFF" + refinputs.multimedia_like(0, 30000, seed=4)),
              ("gif", b"GIF8" + refinputs.multimedia_like(2, 30000, seed=5)), ("ppm", b"P6\n" + refinputs.multimedia_like(1, 30000, seed=6))]
    cases += [(n, d) for n, d in _model_inputs() if n.startswith(("class", "english", "dna", "digits", "ref1", "ref2"))]
    applied = 0
    for name, d in cases:
        for dt in ("UNDEFINED", "MULTIMEDIA", "TEXT"):
            ok_m, enc_m, left_m = katmodels.fsd_forward(d, dt)
            ok_o, enc_o, left_o = oracle.transform_forward("MM", d, data_type=oracle.DT[dt])
            assert (ok_m, oracle.DT[left_m]) == (ok_o, left_o), (name, dt, ok_m, ok_o, left_m, left_o)
            if ok_m and len(d):
                assert enc_m == enc_o, (name, dt)
                applied += 1
    assert applied >= 8
    assert katmodels.log2_1024(1) == 0 and katmodels.log2_1024(3) == 1623 and katmodels.log2_1024(4096) == 12288


def _text_static_words():
    """DICT_EN_1024 (wire-format data: word indexes in coded blocks refer to it) as the generated header holds it"""
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "kzo_text_dict.h")).read()
    return b"".join(m.group(1).encode() for m in re.finditer(r'^\s*"([^"]*)"', src, re.M))


def test_text_codec_known_answers_from_a_python_model(built):
    """TextCodec1.forward / TextCodec2.forward (TextCodec.java:618-870, 1124-1410) with computeStats / detectType (:269-466), the
    static dictionary (:205-236), both word-index codings, the escapes, CR+LF mode, XML flag, the dictionary doubling up to 2^19
    entries and wrapping: a second implementation written from the Java (tests/katmodels.py), since a dictionary coder's output is
    a sequence of CHOICES (which words enter the dictionary, which slot they evict) that a round trip does not pin."""
    import katmodels
    import textgen
    words = _text_static_words()
    c = textgen.cases()
    inputs = [(k, bytes(v)[:160000]) for k, v in c.items()]
    inputs.append(("many_words_4m", textgen.many_words(4 << 20, 21)))            # > 2^19 dictionary entries: doubles, then wraps
    inputs.append(("class0", datagen.block(0, 70000, 0).tobytes()))
    inputs.append(("bulk_english", bytes(textgen.bulk_text(300000, 1001, "english"))))
    inputs.append(("bulk_xml", bytes(textgen.bulk_text(200000, 1002, "xml"))))
    applied = 0
    for name, data in inputs:
        for variant, ent in ((1, "FPAQ"), (2, "ANS0")):
            for bs in ((4 << 20,) if len(data) > 200000 else (65536, 4 << 20)):
                sd = katmodels.text_static_dictionary(words)
                ok_m, out_m, dt_m = katmodels.text_forward(data, variant, bs, sd)
                oracle.set_transform_ctx(ent, bs)
                ok_o, out_o, dt_o = oracle.transform_forward("TEXT", data, data_type=oracle.DT["UNDEFINED"])
                assert ok_m == ok_o, (name, variant, bs)
                assert oracle.DT[dt_m] == dt_o, (name, variant, bs, dt_m, dt_o)
                if ok_o:
                    assert out_m == out_o, (name, variant, bs, len(out_m), len(out_o))
                    applied += 1
    oracle.set_transform_ctx("NONE", 4 << 20)
    assert applied >= 30
    # the data type filter (:641-649): everything but UNDEFINED / TEXT / BIN declines without looking
    eng = bytes(c["english"])[:5000]
    sd = katmodels.text_static_dictionary(words)
    for tag in ("MULTIMEDIA", "EXE", "NUMERIC", "BASE64", "DNA", "UTF8", "SMALL_ALPHABET"):
        assert katmodels.text_forward(eng, 2, 65536, sd, tag)[0] is False
        assert oracle.transform_forward("TEXT", eng, data_type=oracle.DT[tag])[0] is False


def test_knz_streams_equal_a_python_model_of_the_container(built):
    """Levels 5 and 6 (TEXT+UTF+BWT+RANK+ZRLT & ANS0, TEXT+UTF+BWT+SRT+ZRLT & FPAQ) and the core chain: the oracle's whole .knz equals,
    byte for byte, the stream of tests/katmodels.knz_stream -- DefaultOutputBitStream.writeBits (:103-222), the stream header and
    its checksum (CompressedOutputStream.java:236-313), Sequence.forward's skip flags, the block header / mode byte / header
    checksum / raw "transformed copy" fallback (:861-985), the 5 + lw length prefix and the end marker (:1024-1035, :489-492), with
    every stage a Python model written from the Java (and a third suffix sorter for the BWT).  A misreading of the bit I/O or the
    framing would now have to be made twice, independently."""
    import katmodels
    import textgen
    words = _text_static_words()
    c = textgen.cases()
    data = (bytes(c["english"][:40000]) + bytes(c["utf8"][:30000]) + bytes(c["random"][:9000]) + bytes(c["english_crlf"][:20000])
            + datagen.block(2, 12000, 2).tobytes() + bytes(c["xml"][:20000]) + b"tail!")
    short = data[:2 * 16384 + 5]                                                   # the last block is a copy block (<= 15 bytes)
    for names, ent in ((["TEXT", "UTF", "BWT", "RANK", "ZRLT"], "ANS0"), (["TEXT", "UTF", "BWT", "SRT", "ZRLT"], "FPAQ"),
                       (["BWT", "RANK", "ZRLT"], "ANS0"), (["BWT", "MTFT", "ZRLT"], "NONE")):
        for d, bs in ((data, 16384), (data, 65536), (short, 16384)):
            want = oracle.compress("+".join(names), ent, bs, d, jobs=1)
            got = katmodels.knz_stream(d, names, ent, bs, words, len(d))
            assert got == want, (names, ent, bs, len(d), len(got), len(want))
    # the bit writer alone: unaligned byte-array writes behind odd bit counts, against a plain big-integer model
    rng = np.random.default_rng(3)
    os_ = katmodels.JavaOutputBitStream(64)
    ref = katmodels._Bits()
    for _ in range(300):
        k = int(rng.integers(1, 65))
        v = int(rng.integers(0, 1 << 63)) & ((1 << k) - 1)
        os_.write_bits(v, k)
        ref.write(v, k)
        if rng.random() < 0.3:
            nb = int(rng.integers(1, 700))
            raw = rng.integers(0, 256, (nb + 7) // 8, dtype=np.uint8).tobytes()
            os_.write_bytes(raw, 0, nb)
            ref.write(int.from_bytes(raw, "big") >> (8 * len(raw) - nb), nb)
    os_.close()
    assert os_.written() == ref.n and bytes(os_.sink) == ref.bytes()


def test_trie_round_model_reproduces_suffix_arrays():
    """The algorithm of the forward BWT's round 0 (kz_bwt_fwd.hip: count by byte level, classify children as expanded / terminal /
    small, merge small siblings into buckets, sort a bucket by the key bits behind its common prefix) followed by rank doubling from
    h = 6 gives the true suffix array although buckets are resolved to different depths: numpy model in tools/trie_sim.py."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("trie_sim", os.path.join(root, "tools", "trie_sim.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.check(trials=18)


# ---- round 5: the last single-sourced encoder surface ----
def _counts_to_chunk(counts, seed):
    """a chunk (<= 16 KiB) holding symbol v exactly counts[v] times, shuffled"""
    rng = np.random.default_rng(seed)
    return bytes(rng.permutation(np.repeat(np.arange(len(counts), dtype=np.uint8), counts)))


def _fib(n):
    f = [1, 1]
    while len(f) < n:
        f.append(f[-1] + f[-2])
    return f[:n]


def test_huffman_code_length_passes_from_a_python_model_of_the_reference():
    """HuffmanEncoder's in-place Moffat-Katajainen passes, `limitCodeLengths` with its linked lists AND its renormalising slow
    path, and the canonical codes (tests/katmodels.huffman_encode_exact, written from HuffmanEncoder.java:191-376, not from
    oracle/kzo_huffman.c) on inputs where ties between equal weights decide the lengths and where the optimal code is deeper
    than 12 bits: the C oracle must write the same bits."""
    import katmodels
    rng = np.random.default_rng(21)
    cases = []
    # ties everywhere: flat alphabets of every size class, two-level weights, all-ones plus one heavy symbol
    for k in (2, 3, 5, 17, 100, 255, 256):
        cases.append(("flat%d" % k, _counts_to_chunk([40] * k if 40 * k <= 16384 else [16384 // k] * k, k)))
    cases.append(("two_levels", _counts_to_chunk([8] * 100 + [64] * 100, 1)))
    cases.append(("ones_and_heavy", _counts_to_chunk([1] * 200 + [9000], 2)))
    cases.append(("pairs_of_ties", _counts_to_chunk([c for c in range(1, 90) for _ in (0, 1)], 3)))
    # deeper than 12 bits: Fibonacci weights (the worst case of Huffman depth), with and without ties, scaled
    for n in (14, 16, 19, 20):
        cases.append(("fib%d" % n, _counts_to_chunk(_fib(n), n)))
    cases.append(("fib19_twice", _counts_to_chunk(sorted(_fib(17) + _fib(17)), 4)))
    cases.append(("fib_and_ones", _counts_to_chunk(_fib(18) + [1] * 150, 5)))
    cases.append(("powers", _counts_to_chunk([1, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8190], 6)))
    cases.append(("geometric_tail", _counts_to_chunk([max(1, int(12000 * 0.55 ** i)) for i in range(40)] + [1] * 100, 7)))
    # several chunks, a tail below 32 bytes, a tail chunk with one symbol
    cases.append(("multi_chunk", datagen.block(1, 40000).tobytes() + b"\x07" * 50 + b"tail"))
    cases.append(("text_class", datagen.block(0, 50000).tobytes()))
    for i in range(40):                                               # random skewed histograms: whatever path they take
        k = int(rng.integers(2, 257))
        w = rng.random(k) ** float(rng.uniform(1, 14))
        counts = np.maximum(1, (w / w.sum() * 16000).astype(np.int64))
        while counts.sum() > 16384:
            counts[np.argmax(counts)] -= counts.sum() - 16384
        cases.append(("rand%d" % i, _counts_to_chunk(list(counts), 100 + i)))
    paths = {"limit": 0, "slow": 0, "flat8": 0}
    orig_limit, orig_norm = katmodels.huffman_limit_code_lengths, katmodels.normalize_frequencies_java

    def limit_spy(*a):
        paths["limit"] += 1
        return orig_limit(*a)

    def norm_spy(*a):
        paths["slow"] += 1
        return orig_norm(*a)

    katmodels.huffman_limit_code_lengths, katmodels.normalize_frequencies_java = limit_spy, norm_spy
    try:
        for name, data in cases:
            model, nbits = katmodels.huffman_encode_exact(data)
            enc, obits = oracle.entropy_encode("HUFFMAN", data)
            assert obits == nbits and enc[:len(model)] == model, (name, len(data), nbits, obits)
            r, back, _ = oracle.entropy_decode("HUFFMAN", enc, obits, len(data))
            assert r == len(data) and back == data, name
    finally:
        katmodels.huffman_limit_code_lengths, katmodels.normalize_frequencies_java = orig_limit, orig_norm
    # the inputs must have reached the limiter, and its slow path
    assert paths["limit"] >= 8 and paths["slow"] >= 1, paths
    # the tie-free model of round 1 and the exact one agree where the first is defined
    dyadic = bytes([97] * 32 + [98] * 16 + [99] * 8 + [100] * 4 + [101] * 2 + [102, 103])
    assert katmodels.huffman_encode_exact(dyadic) == katmodels.huffman_encode(dyadic)


def test_normalize_frequencies_from_a_python_model_of_the_reference():
    """EntropyUtils.normalizeFrequencies (tests/katmodels.normalize_frequencies_java, from EntropyUtils.java:141-250) against the
    oracle's kzo_normalize_freqs on tie-heavy, Fibonacci and random histograms at every scale the codecs use (2^8 .. 2^16):
    both rounding paths, the fast correction of the maximum and the slow spreading loop with its five rounds."""
    import ctypes
    import katmodels
    L = oracle.lib()
    L.kzo_normalize_freqs.restype = ctypes.c_int
    L.kzo_normalize_freqs.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int]
    rng = np.random.default_rng(5)
    hists = [[7] * 256, [1] * 255 + [100000], [3] * 100 + [0] * 156, _fib(24) + [0] * 232, [1] * 128 + [2] * 128, [0] * 255 + [5],
             [1, 0, 1] + [0] * 253, [16384] + [1] * 255, [100] * 3 + [0] * 253]
    for i in range(300):
        k = int(rng.integers(1, 257))
        h = np.zeros(256, dtype=np.int64)
        idx = rng.permutation(256)[:k]
        w = rng.random(k) ** float(rng.uniform(0.2, 12))
        h[idx] = np.maximum(1, (w / w.sum() * int(rng.integers(k, 70000))).astype(np.int64))
        if i % 3 == 0:
            h[idx] = (h[idx] // 8 + 1) * 8                              # ties
        hists.append([int(v) for v in h])
    slow = 0
    for h in hists:
        total = sum(h)
        for lg in (8, 10, 11, 12, 13, 16):
            scale = 1 << lg
            f = list(h)
            alpha = katmodels.normalize_frequencies_java(f, 256, total, scale)
            cf = (ctypes.c_int * 257)(*h)
            ca = (ctypes.c_int * 256)()
            n = L.kzo_normalize_freqs(cf, ca, total, scale)
            assert n == len(alpha) and list(ca[:n]) == alpha and list(cf[:256]) == f, (h[:8], lg)
            if total != scale and len(alpha) > 1:
                assert sum(f) == scale or min(v for v in f if v) <= 2, (lg, sum(f))
                slow += 1
    assert slow > 500


# ---- decoders against Python models written from the Java (VERDICT r4 "missing" 1: the decoders' behaviour on malformed input
# was the oracle's reading alone) ----
def _model_verdict(fn, *a):
    import katmodels
    try:
        ok, out = fn(*a)
    except katmodels.JavaException:
        return False, b""
    return ok, out


def _damaged(rng, enc, k):
    """k damaged copies of a coded block: flipped bytes, bytes forced to the codes' special values, cuts, insertions"""
    out = []
    for j in range(k):
        b = bytearray(enc)
        kind = j % 5
        if kind == 0 and len(b):
            for _ in range(1 + int(rng.integers(0, 4))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif kind == 1 and len(b):
            b[int(rng.integers(0, len(b)))] = (0, 1, 0xFF, 0x80, 0xFE)[int(rng.integers(0, 5))]
        elif kind == 2 and len(b) > 2:
            del b[int(rng.integers(1, len(b))):]
        elif kind == 3:
            at = int(rng.integers(0, len(b) + 1))
            b[at:at] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
        elif len(b) > 8:
            at = int(rng.integers(0, min(len(b), 600)))
            b[at] = int(rng.integers(0x80, 0x100))
        out.append(bytes(b))
    return out


def test_zrlt_sbrt_srt_inverse_from_python_models_of_the_reference_decoders(built):
    """ZRLT.inverse (ZRLT.java:146-231), SBRT.inverse (SBRT.java:154-214), SRT.inverse (SRT.java:178-263, decodeHeader :327-346):
    valid streams, outputs that fit exactly / miss by one, and damaged copies -- verdict and bytes"""
    import katmodels
    rng = np.random.default_rng(5)
    inputs = [(n, d[:6000]) for n, d in _model_inputs() if d]
    inputs += [("zeros", bytes(9000)), ("zero_tail", bytes(rng.integers(0, 3, 3000, dtype=np.uint8)) + bytes(700)),
               ("ff", bytes([0xFF, 0xFE, 0, 0, 0xFF]) * 400), ("one", b"q" * 2000)]
    checked = failed = 0
    for name, d in inputs:
        for mode, t in ((1, "MTFT"), (2, "RANK")):
            ok, enc = oracle.transform_forward(t, d)
            assert ok and katmodels.sbrt_inverse(enc, mode) == d == oracle.transform_inverse(t, enc, len(d))[1], (name, t)
            x = bytes(rng.integers(0, 256, 3000, dtype=np.uint8))
            assert katmodels.sbrt_inverse(x, mode) == oracle.transform_inverse(t, x, len(x))[1], (name, t)
        assert katmodels.sbrt_inverse(d[:3000], 3) == _sbrt_inverse_oracle(3, d[:3000]), name
        ok, enc = oracle.transform_forward("ZRLT", d)
        if ok:
            for cap in (len(d), len(d) + 1, len(d) - 1, len(d) + 100, max(len(d) // 2, 1)):
                got, want = _model_verdict(katmodels.zrlt_inverse, enc, cap), oracle.transform_inverse("ZRLT", enc, cap)
                assert got[0] == want[0] and (not got[0] or got[1] == want[1]), (name, cap, got[0], want[0])
            for bad in _damaged(rng, enc, 24):
                for cap in (len(d), len(d) + 64):
                    got, want = _model_verdict(katmodels.zrlt_inverse, bad, cap), oracle.transform_inverse("ZRLT", bad, cap)
                    assert got[0] == want[0] and (not got[0] or got[1] == want[1]), (name, cap, bad[:16])
                    checked += 1
                    failed += not got[0]
        ok, enc = oracle.transform_forward("SRT", d)
        assert ok
        for cap in (len(d), len(d) - 1, len(d) + 9):
            got, want = _model_verdict(katmodels.srt_inverse, enc, cap), oracle.transform_inverse("SRT", enc, cap)
            assert got[0] == want[0] and (not got[0] or got[1] == want[1] == d), (name, cap)
        for bad in _damaged(rng, enc, 24):
            got, want = _model_verdict(katmodels.srt_inverse, bad, len(d) + 64), oracle.transform_inverse("SRT", bad, len(d) + 64)
            assert got[0] == want[0] and (not got[0] or got[1] == want[1]), (name, got[0], want[0], bad[:8])
            checked += 1
            failed += not got[0]
    assert checked > 1000 and 0 < failed < checked


def _sbrt_inverse_oracle(mode, data):
    a = np.frombuffer(bytes(data), dtype=np.uint8)
    out = np.zeros(max(len(a), 1), dtype=np.uint8)
    fn = oracle.lib().kzo_sbrt_inverse
    fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    assert fn(mode, a.ctypes.data if len(a) else out.ctypes.data, len(a), out.ctypes.data)
    return out[:len(a)].tobytes()


def test_utf_inverse_from_a_python_model_of_the_reference_decoder(built):
    """UTFCodec.inverse (UTFCodec.java:224-306, unpackV1 :508-541): valid streams, tight outputs, damaged headers / tables / aliases"""
    import katmodels
    import textgen
    rng = np.random.default_rng(6)
    tc = textgen.cases()
    checked = failed = applied = 0
    for name in ("utf8", "utf8_bom", "utf8_cut"):
        d = tc[name][:30000]
        ok, enc, _ = oracle.transform_forward("UTF", d, data_type=oracle.DT["UNDEFINED"])
        if not ok:
            continue
        applied += 1
        for cap in (len(d), len(d) + 1, len(d) + 4, len(d) + 5, len(d) + 64, len(d) - 1, 3):
            got, want = _model_verdict(katmodels.utf_inverse, enc, cap), oracle.transform_inverse("UTF", enc, cap)
            assert got[0] == want[0] and (not got[0] or got[1] == want[1] == d), (name, cap, got[0], want[0])
        for bad in _damaged(rng, enc, 150):
            cap = len(d) + 64
            got, want = _model_verdict(katmodels.utf_inverse, bad, cap), oracle.transform_inverse("UTF", bad, cap)
            assert got[0] == want[0] and (not got[0] or got[1] == want[1]), (name, got[0], want[0], bad[:8])
            checked += 1
            failed += not got[0]
    assert applied >= 2 and checked > 250 and 0 < failed < checked


def test_text_inverse_from_a_python_model_of_the_reference_decoder(built):
    """TextCodec1.inverse / TextCodec2.inverse (TextCodec.java:876-1031, 1410-1603), the last decoder that only the oracle restated
    (VERDICT r5 item 6): a Python model written from the Java (tests/katmodels.text_inverse) against the oracle AND the library's
    host stage on valid blocks (English, CR+LF, XML, escape bytes, a word list that doubles, short blocks), on outputs that fit
    exactly / miss by a few bytes, and on more than a thousand damaged copies of each codec's blocks (flipped bytes, forced escape /
    index bytes, cuts, insertions: word indexes behind the dictionary, references to words not learned yet, escapes cut by the
    block's end, outputs that no longer fit): the same verdict and, where the block decodes, the same bytes."""
    import katmodels
    import textgen
    import kanzi_amd as kz
    words = _text_static_words()
    c = textgen.cases()
    rng = np.random.default_rng(66)
    inputs = [("english", bytes(c["english"])[:16000]), ("crlf", bytes(c["english_crlf"])[:12000]), ("xml", bytes(c["xml"])[:12000]),
              ("escapes", bytes(c["english_escapes"])[:12000]), ("many_words", bytes(c["many_words"])[:70000]), ("min", bytes(c["min"])),
              ("bulk", bytes(textgen.bulk_text(20000, 77, "english")))]
    checked = {1: 0, 2: 0}
    failed = {1: 0, 2: 0}
    valid = 0
    for name, d in inputs:
        for variant, ent in ((1, "FPAQ"), (2, "ANS0")):
            bs = 65536 if len(d) <= 65536 else 1 << 20
            oracle.set_transform_ctx(ent, bs)
            ok, enc, _ = oracle.transform_forward("TEXT", d, data_type=oracle.DT["UNDEFINED"])
            if not ok:
                continue
            valid += 1
            assert ((enc[0] & 0x10) != 0) == (variant == 2), (name, variant)

            def three(block, cap):
                sd = katmodels.text_static_dictionary(words)
                got = _model_verdict(lambda b, k: katmodels.text_inverse(b, bs, sd, k), block, cap)
                want = oracle.transform_inverse("TEXT", block, cap)
                host = kz.host_stage_inverse("TEXT", block, cap, block_size=bs)
                assert got[0] == want[0] == host[0], (name, variant, cap, got[0], want[0], host[0], bytes(block[:12]))
                if want[0]:
                    assert got[1] == want[1] == host[1], (name, variant, cap)
                return want[0]

            for cap in (len(d), len(d) + 1, len(d) + 2, len(d) + 40, len(d) + 4096, len(d) - 1, len(d) - 7, len(d) // 2):
                r = three(enc, cap)
                assert r == (cap > len(d)) or cap == len(d), (name, variant, cap)           # dstIdx + length >= dstEnd needs room behind the last word
            for bad in _damaged(rng, enc, 110 if len(enc) < 50000 else 60):
                r = three(bad, len(d) + 64)
                checked[variant] += 1
                failed[variant] += not r
            for _ in range(40):                                                                 # aimed: an index or escape byte where a token starts
                b = bytearray(enc)
                at = int(rng.integers(1, len(b)))
                b[at] = int(rng.choice([0x0F, 0x0E, 0x80, 0xC0, 0xF0, 0xFF, 0xEF, 0x8F, 0x0A, 0x0D])) if variant == 2 else int(rng.choice([0x0F, 0x0E, 0x80, 0xFF, 0xE0, 0x0A]))
                if rng.random() < 0.5 and at + 2 < len(b):
                    b[at + 1] = int(rng.integers(0, 256)); b[at + 2] = int(rng.integers(0, 256))
                r = three(bytes(b), len(d) + 64)
                checked[variant] += 1
                failed[variant] += not r
    oracle.set_transform_ctx("NONE", 4 << 20)
    assert valid >= 12
    for v in (1, 2):
        assert checked[v] >= 1000 and 0 < failed[v] < checked[v], (v, checked, failed)


def test_bwt_block_inverse_from_a_python_model_of_the_reference_decoder(built):
    """BWTBlockCodec.inverse (BWTBlockCodec.java:131-201) + BWT.inverse / inverseMergeTPSI (BWT.java:203-235, :289-381): valid
    blocks of every header shape, short outputs, damaged mode bytes and primary indexes (a wrong index in range = wrong bytes with
    a success verdict, in the model as in the oracle)"""
    import katmodels
    rng = np.random.default_rng(8)
    inputs = [(n, d[:5000]) for n, d in _model_inputs() if d][:8]
    inputs += [("n%d" % n, bytes(rng.integers(97, 101, n, dtype=np.uint8))) for n in (1, 2, 3, 4, 5, 9, 63, 64, 255, 256, 257, 263, 264, 1000, 20001)]
    checked = failed = wrong_but_ok = 0
    for name, d in inputs:
        ok, enc = oracle.transform_forward("BWT", d)
        if not ok:                                                               # (the forward declines the smallest blocks)
            continue
        for cap in (len(d), len(d) + 7, len(d) - 1):
            got, want = _model_verdict(katmodels.bwt_block_inverse, enc, cap), oracle.transform_inverse("BWT", enc, cap)
            assert got[0] == want[0] and (not got[0] or got[1] == want[1]), (name, cap, got[0], want[0])
            assert not got[0] or got[1] == d or len(d) < 6, (name, cap)          # (tiny blocks: the header is longer than the body, BWT.inverse's index test)
        if len(d) < 256:
            continue
        hdr = len(enc) - len(d)
        for j in range(60):
            b = bytearray(enc)
            if j % 3 == 0:
                b[0] = int(rng.integers(0, 256))
            elif j % 3 == 1:
                b[int(rng.integers(1, hdr))] = int(rng.integers(0, 256))
            else:
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
            bad = bytes(b)
            got, want = _model_verdict(katmodels.bwt_block_inverse, bad, len(d) + 64), oracle.transform_inverse("BWT", bad, len(d) + 64)
            assert got[0] == want[0] and (not got[0] or got[1] == want[1]), (name, j, got[0], want[0])
            checked += 1
            failed += not got[0]
            wrong_but_ok += got[0] and got[1] != d
    assert checked > 500 and failed > 0 and wrong_but_ok > 0


def test_lz_inverse_on_damaged_frames_from_the_python_model(built):
    """LZCodec.inverseV6 (LZCodec.java:627-756) on damaged LZ / LZX frames: verdict and bytes of katmodels.lz_decode (a Python
    IndexError = the Java's array fault).  Writing this test found an error in the MODEL (the walk ends at the first literal run that
    reaches tkIdx - 13, :647/:676, not the token stream itself); the oracle had it right."""
    import katmodels
    rng = np.random.default_rng(9)
    cases = [b"abc" * 420, datagen.block(0, 20000).tobytes(), datagen.block(2, 30000).tobytes(), bytes(rng.integers(0, 4, 20000, dtype=np.uint8)),
             bytes(1000) + b"xyz" * 500]
    checked = failed = 0
    for name in ("LZ", "LZX"):
        for d in cases:
            ok, enc = oracle.transform_forward(name, d)
            if not ok:
                continue
            for bad in _damaged(rng, enc, 80):
                cap = len(d) + 64
                try:
                    m = katmodels.lz_decode(bad, cap)
                except IndexError:
                    m = None
                o = oracle.transform_inverse(name, bad, cap)
                assert (m is not None) == o[0] and (m is None or m == o[1]), (name, len(d), bad[:13].hex())
                checked += 1
                failed += m is None
    assert checked >= 700 and 0 < failed < checked


def test_mm_inverse_from_a_python_model_of_the_reference_decoder(built):
    """FSDCodec.inverse (FSDCodec.java:249-313), both codings: valid blocks, exact-fit / one-short / one-spare outputs, damaged copies
    (mode and distance bytes, escapes at the very end, insertions that overrun the output)"""
    import katmodels
    rng = np.random.default_rng(10)
    applied = checked = failed = 0
    for kind in range(5):
        d = refinputs.multimedia_like(kind, 12000, seed=kind)
        ok, enc = oracle.transform_forward("MM", d)
        if not ok:
            continue
        applied += 1
        for cap in (len(d), len(d) + 1, len(d) - 1, len(d) + 64):
            got, want = _model_verdict(katmodels.fsd_inverse, enc, cap), oracle.transform_inverse("MM", enc, cap)
            assert got[0] == want[0] and (not got[0] or got[1] == want[1] == d), (kind, cap - len(d))
        for bad in _damaged(rng, enc, 100):
            for cap in (len(d) + 64, len(d)):
                got, want = _model_verdict(katmodels.fsd_inverse, bad, cap), oracle.transform_inverse("MM", bad, cap)
                assert got[0] == want[0] and (not got[0] or got[1] == want[1]), (kind, cap - len(d), bad[:3].hex())
                checked += 1
                failed += not got[0]
    assert applied >= 3 and checked >= 600 and 0 < failed < checked


def test_alias_inverse_from_a_python_model_of_the_reference_decoder(built):
    """AliasCodec.inverse (AliasCodec.java:281-418): the one-symbol, 2-bit, 4-bit and digram-alias forms of PACK / DNA blocks, tight
    outputs, damaged copies (a negative one-symbol size makes the Java call 'succeed' with a negative length: a failed block)"""
    import katmodels
    rng = np.random.default_rng(11)
    ins = [("four", bytes(rng.choice(np.frombuffer(b"acgt", dtype=np.uint8), 4099))), ("one", b"z" * 3000),
           ("sixteen", bytes(rng.integers(64, 80, 5001, dtype=np.uint8))), ("three", bytes(rng.integers(64, 67, 4002, dtype=np.uint8)))]
    ins += [(k, bytes(v)[:12000]) for k, v in refinputs.alias_inputs()]
    forms, checked, failed = set(), 0, 0
    for name, d in ins:
        for t in ("PACK", "DNA"):
            ok, enc = oracle.transform_forward(t, d)
            if not ok:
                continue
            forms.add("one" if enc[0] == 255 else "2bit" if enc[0] >= 252 else "4bit" if enc[0] >= 240 else "alias")
            for cap in (len(d), len(d) + 1, len(d) - 1, len(d) + 64):
                got, want = _model_verdict(katmodels.alias_inverse, enc, cap), oracle.transform_inverse(t, enc, cap)
                assert got[0] == want[0] and (not got[0] or got[1] == want[1] == d), (name, t, enc[0], cap - len(d))
            for bad in _damaged(rng, enc, 60):
                got, want = _model_verdict(katmodels.alias_inverse, bad, len(d) + 64), oracle.transform_inverse(t, bad, len(d) + 64)
                assert got[0] == want[0] and (not got[0] or got[1] == want[1]), (name, t, bad[:6].hex())
                checked += 1
                failed += not got[0]
    assert forms == {"one", "2bit", "4bit", "alias"} and checked > 1500 and 0 < failed < checked


def test_fpaq_decoder_from_a_python_model_of_the_reference_decoder(built):
    """FPAQDecoder.decode (FPAQDecoder.java:164-234, decodeBitV2 :294-314, read :322-334): return value, bytes and bits consumed on
    valid bit strings; verdict and bytes on damaged ones (most damaged FPAQ strings still decode -- to the same wrong bytes)"""
    import katmodels
    rng = np.random.default_rng(12)
    cases = [datagen.block(c, 6000, c).tobytes() for c in range(5)] + [b"a" * 3000, bytes(rng.integers(0, 256, 40, dtype=np.uint8))]

    def model(enc, nbits, n):
        try:
            return katmodels.fpaq_decode(enc, nbits, n)
        except katmodels.JavaException:
            return -1, b"", 0

    checked = failed = 0
    for d in cases:
        enc, nbits = oracle.entropy_encode("FPAQ", d)
        m, o = model(enc, nbits, len(d)), oracle.entropy_decode("FPAQ", enc, nbits, len(d))
        assert m[0] == o[0] == len(d) and m[1] == o[1] == d and m[2] == o[2] == nbits
        for bad in _damaged(rng, enc, 40):
            nb = min(nbits, len(bad) * 8) if len(bad) < len(enc) else nbits + (len(bad) - len(enc)) * 8
            m, o = model(bad, nb, len(d)), oracle.entropy_decode("FPAQ", bad, nb, len(d))
            assert (m[0] == len(d)) == (o[0] == len(d)) and (m[0] != len(d) or m[1] == o[1]), (len(d), m[0], o[0], bad[:4].hex())
            checked += 1
            failed += m[0] != len(d)
    assert checked >= 250 and 0 < failed < checked


def test_ans0_decoder_from_a_python_model_of_the_reference_decoder(built):
    """ANSRangeDecoder order 0 (ANSRangeDecoder.java: decode :158-206, decodeHeader :374-466, decodeChunkV2 :281-366) +
    EntropyUtils.decodeAlphabet (:86-118): return value, bytes and bits consumed on valid bit strings; the verdict on damaged ones
    (a chunk whose byte count does not come out ends the walk but decode STILL returns count: the oracle does the same), the bytes
    too whenever every chunk came out"""
    import katmodels
    rng = np.random.default_rng(13)
    cases = [datagen.block(c, 40000, c).tobytes() for c in range(5)] + [b"a" * 3000, bytes(rng.integers(0, 256, 30, dtype=np.uint8)),
             bytes(rng.integers(0, 256, 5000, dtype=np.uint8)), bytes(rng.integers(0, 3, 20001, dtype=np.uint8))]

    def model(enc, nbits, n):
        try:
            return katmodels.ans0_decode(enc, nbits, n)
        except katmodels.JavaException:
            return -1, b"", 0, True

    checked = failed = early = 0
    for d in cases:
        enc, nbits = oracle.entropy_encode("ANS0", d)
        m, o = model(enc, nbits, len(d)), oracle.entropy_decode("ANS0", enc, nbits, len(d))
        assert m[0] == o[0] == len(d) and m[1] == o[1] == d and m[2] == o[2] == nbits and m[3]
        for bad in _damaged(rng, enc, 30):
            nb = min(nbits, len(bad) * 8) if len(bad) < len(enc) else nbits + (len(bad) - len(enc)) * 8
            m, o = model(bad, nb, len(d)), oracle.entropy_decode("ANS0", bad, nb, len(d))
            ok_m = m[0] == len(d)
            assert ok_m == (o[0] == len(d)) and (not (ok_m and m[3]) or m[1] == o[1]), (len(d), m[0], o[0], m[3], bad[:4].hex())
            checked += 1
            failed += not ok_m
            early += ok_m and not m[3]
    assert checked >= 250 and failed > 0 and early > 0


def test_huffman_decoder_from_a_python_model_of_the_reference_decoder(built):
    """HuffmanDecoder (HuffmanDecoder.java: decodeV6 :353-383, readLengths :116-150, buildDecodingTables :153-172, decodeChunk with
    its closing test that each of the four fragments took exactly its stated bits): return value, bytes, bits consumed on valid bit
    strings; verdict and bytes on damaged ones.  This model found a difference: the oracle (and the HIP decoder) did not make
    decodeChunk's closing test and accepted damaged fragments the reference refuses -- fixed in both."""
    import katmodels
    rng = np.random.default_rng(14)
    cases = [datagen.block(c, 40000, c).tobytes() for c in range(5)] + [b"a" * 3000, bytes(rng.integers(0, 256, 30, dtype=np.uint8)),
             bytes(rng.integers(0, 256, 5000, dtype=np.uint8)), bytes(rng.integers(0, 3, 20001, dtype=np.uint8))]

    def model(enc, nbits, n):
        try:
            return katmodels.huffman_decode(enc, nbits, n)
        except katmodels.JavaException:
            return -1, b"", 0

    checked = failed = 0
    for d in cases:
        enc, nbits = oracle.entropy_encode("HUFFMAN", d)
        m, o = model(enc, nbits, len(d)), oracle.entropy_decode("HUFFMAN", enc, nbits, len(d))
        assert m[0] == o[0] == len(d) and m[1] == o[1] == d and m[2] == o[2] == nbits
        for bad in _damaged(rng, enc, 40):
            nb = min(nbits, len(bad) * 8) if len(bad) < len(enc) else nbits + (len(bad) - len(enc)) * 8
            m, o = model(bad, nb, len(d)), oracle.entropy_decode("HUFFMAN", bad, nb, len(d))
            ok_m = m[0] == len(d)
            assert ok_m == (o[0] == len(d)) and (not ok_m or m[1] == o[1]), (len(d), m[0], o[0], bad[:4].hex())
            checked += 1
            failed += not ok_m
    assert checked >= 350 and 0 < failed < checked
