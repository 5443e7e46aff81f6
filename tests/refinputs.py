"""The reference's own test inputs, re-generated (no expected outputs exist upstream: round-trip only).
  transforms : T/test/TestTransforms.java:183-254 (java.util.Random(Long.MAX_VALUE) LCG reproduced)
  entropy    : T/test/TestEntropyCodec.java:219-244
  bwt        : T/test/TestBWT.java:85-103
"""
import numpy as np

from oracle import JavaRandom

LONG_MAX = (1 << 63) - 1


def transform_inputs():
    rnd = JavaRandom(LONG_MAX)
    out = []
    for ii in range(0, 51):
        if ii == 0:
            arr = [0, 1, 2, 2, 2, 2, 7, 9, 9, 16, 16, 16, 1] + [3] * 19
        elif ii < 10:
            arr = [ii] * 80000
        elif ii in (10, 11):
            arr = [1] + [8] * 79999
        elif ii == 12:
            arr = [0, 0, 1, 1, 2, 2, 3, 3]
        elif ii == 13:
            arr = [0] * 512
            for i in range(256):
                arr[2 * i] = i
                arr[2 * i + 1] = i
            arr[1] = 255
        elif ii < 16:
            arr = []
            for _ in range(1 << (ii + 6)):
                v = rnd.next_int(100)
                arr.append(0 if v >= 33 else v)
        elif ii == 16:
            arr = [0] * 20 + [rnd.next_int(256) for _ in range(20, 512)]
        else:
            arr = [0] * 1024
            idx = 20
            while idx < 1024:
                ln = rnd.next_int(120)
                if ln % 3 == 0:
                    ln = 1
                val = rnd.next_int(256)
                end = min(idx + ln, 1024)
                for j in range(idx, end):
                    arr[j] = val
                idx += ln
        out.append(bytes(x & 0xFF for x in arr))
    return out


def entropy_inputs():
    rnd = JavaRandom(LONG_MAX)
    out = []
    for ii in range(1, 20):
        if ii == 3:
            v = [0, 0, 32, 15, -4, 16, 0, 16, 0, 7, -1, -4, -32, 0, 31, -1]
        elif ii == 2:
            v = [61, 77, 84, 71, 90, 54, 57, 38, 114, 111, 108, 101, 61, 112, 114, 101]
        elif ii == 1:
            v = [2] * 40
        elif ii == 4:
            v = [2 + (i & 1) for i in range(40)]
        elif ii == 5:
            v = [42]
        elif ii == 6:
            v = [42, 42]
        else:
            v = [64 + 4 * ii + rnd.next_int(8 * ii + 1) for _ in range(256)]
        out.append(bytes(x & 0xFF for x in v))
    return out


def bwt_inputs():
    rnd = JavaRandom(LONG_MAX)
    out = [b"mississippi", b"3.14159265358979323846264338327950288419716939937510", b"SIX.MIXED.PIXIES.SIFT.SIXTY.PIXIE.DUST.BOXES"]
    out.append(bytes(65 + rnd.next_int(4 * 1) for _ in range(128)))
    return out


def edge_inputs():
    """empty / ragged / threshold sizes: copy-block limit 15|16, ANS raw limit 32|33, BWT chunk switch 255|256|257."""
    rng = np.random.default_rng(12345)
    out = []
    for n in (1, 2, 3, 15, 16, 17, 31, 32, 33, 34, 63, 64, 65, 255, 256, 257, 1000, 4095, 4096, 4097, 16383, 16384, 16385, 16417, 70001):
        out.append(bytes(rng.integers(0, 4, n, dtype=np.uint8)))
        out.append(bytes(np.minimum(rng.geometric(0.3, n) - 1, 255).astype(np.uint8)))
    out.append(bytes([0xFF] * 300))
    out.append(bytes([0xFE, 0xFF] * 200))
    out.append(bytes(5000))
    out.append(bytes([1] + [0] * 4999))
    out.append(bytes([0] * 4999 + [1]))
    return out


def set_bits(buf, bitoff, nbits, value):
    """Overwrite nbits (MSB first, the bitstream's order) at bit offset bitoff of bytearray buf."""
    for i in range(nbits):
        bit = (value >> (nbits - 1 - i)) & 1
        pos = bitoff + i
        if bit:
            buf[pos >> 3] |= 0x80 >> (pos & 7)
        else:
            buf[pos >> 3] &= ~(0x80 >> (pos & 7)) & 0xFF


def header_faults(good):
    """Stream-header faults and the K/Error.java code CompressedInputStream.readHeader throws for each, in the order it
    checks them (CompressedInputStream.java:363-478).  Field offsets in bits: magic 0, version 32, checksum kind 36,
    entropy 38, transforms 43, block size 91, size mask 119."""
    cases = []

    def mk(off, nbits, value, code, what):
        b = bytearray(good)
        set_bits(b, off, nbits, value)
        cases.append((what, bytes(b), code))

    mk(0, 32, 0x4B414E5B, 15, "magic")                 # ERR_INVALID_FILE :367-368
    mk(32, 4, 8, 16, "version")                        # ERR_STREAM_VERSION :374-377
    mk(36, 2, 3, 15, "checksum kind 3")                # ERR_INVALID_FILE :390-392
    mk(38, 5, 3, 3, "entropy id 3 (obsolete PAQ)")     # ERR_INVALID_CODEC :399-406
    mk(38, 5, 31, 3, "entropy id 31")
    mk(43, 6, 4, 3, "transform id 4 (SNAPPY, removed)")  # ERR_INVALID_CODEC :408-415
    mk(43 + 42, 6, 63, 3, "transform id 63 in slot 8")
    mk(91, 28, 1, 2, "block size 16")                  # ERR_BLOCK_SIZE :419-422
    mk(91, 28, (1 << 26) + 1, 2, "block size > 1 GiB")
    mk(119, 2, 0, 19, "size mask cleared")             # ERR_CRC_CHECK :477-478
    mk(38, 5, 1, 19, "entropy id changed to a valid one")
    return cases


def corrupt(rng, good, kind):
    """One corruption of a well-formed buffer: 0 bit flips, 1 truncation, 2 a 64-byte garbage splice, 3 deleted bytes,
    4 bit flips in the first 300 bytes (headers), 5 all garbage, 6 zeroed tail (what a stopped entropy decode leaves),
    7 a long run of 0/1 bytes (ZRLT digits, wraps the Java int run length)."""
    import numpy as np
    bad = bytearray(good)
    if kind == 0:
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(0, len(bad)))
            bad[pos] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:
        bad = bad[:int(rng.integers(1, len(bad)))]
    elif kind == 2:
        a = int(rng.integers(0, max(1, len(bad) - 64)))
        bad[a:a + 64] = bytes(rng.integers(0, 256, 64, dtype=np.uint8))
    elif kind == 3:
        a = int(rng.integers(0, max(1, len(bad) - 8)))
        del bad[a:a + int(rng.integers(1, 8))]
    elif kind == 4:
        for _ in range(int(rng.integers(1, 3))):
            pos = int(rng.integers(0, min(len(bad), 300)))
            bad[pos] ^= 1 << int(rng.integers(0, 8))
    elif kind == 5:
        bad = bytearray(rng.integers(0, 256, len(bad), dtype=np.uint8).tobytes())
    elif kind == 6:
        a = int(rng.integers(0, len(bad)))
        bad[a:] = bytes(len(bad) - a)
    else:
        a = int(rng.integers(0, max(1, len(bad) - 80)))
        k = int(rng.integers(28, 70))
        bad[a:a + k] = bytes(rng.integers(0, 2, k, dtype=np.uint8))
    return bytes(bad)


def multimedia_like(kind, n, seed=1):
    """Inputs FSDCodec (MM) applies to: 0 = 16-bit little-endian PCM-like samples (XOR coding, step 2), 1 = smooth RGB
    pixels (delta coding, step 3), 2 = smooth 8-bit samples (delta, step 1), 3 = RGBA with sharp edges every 97 pixels
    (delta coding with escape tokens, step 4), 4 = two interleaved 16-bit channels (step 4)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    if kind == 0:
        t = np.arange(n // 2 + 1)
        x = (8000 * np.sin(t / 37.0) + 3000 * np.sin(t / 5.1) + rng.normal(0, 40, t.size)).astype(np.int16)
        return x.astype("<i2").tobytes()[:n]
    if kind == 1:
        w = n // 3 + 1
        t = np.arange(w)
        px = np.stack([128 + 100 * np.sin(t / 91.0), 128 + 90 * np.sin(t / 57.0 + 1), 100 + 80 * np.sin(t / 33.0 + 2)], 1)
        return np.clip(px + rng.normal(0, 1.5, (w, 3)), 0, 255).astype(np.uint8).tobytes()[:n]
    if kind == 2:
        t = np.arange(n)
        return np.clip(128 + 100 * np.sin(t / 45.0) + rng.normal(0, 2, n), 0, 255).astype(np.uint8).tobytes()
    if kind == 3:
        w = n // 4 + 1
        t = np.arange(w)
        base = np.where((t // 97) % 2 == 0, 30.0, 220.0)
        px = np.stack([base + 10 * np.sin(t / 9.0), 255 - base, base * 0.5 + 60, np.full(w, 255.0)], 1)
        return np.clip(px + rng.normal(0, 1.0, (w, 4)), 0, 255).astype(np.uint8).tobytes()[:n]
    t = np.arange(n // 4 + 1)
    a = (6000 * np.sin(t / 23.0) + rng.normal(0, 30, t.size)).astype(np.int16)
    b = (5000 * np.cos(t / 41.0) + rng.normal(0, 30, t.size)).astype(np.int16)
    return np.stack([a, b], 1).astype("<i2").tobytes()[:n]


def alias_inputs():
    """(label, bytes) inputs for AliasCodec (PACK / DNA): every branch -- one symbol, <= 4 symbols (2-bit packing, all
    four values of count & 3), <= 16 symbols (4-bit packing, odd and even lengths), digram aliasing on text-like data
    (with and without a trailing odd byte), and inputs it declines (no absent symbols, too few savings, too short)."""
    import numpy as np
    import datagen
    rng = np.random.default_rng(2024)
    pick = lambda alphabet, n: bytes(np.frombuffer(alphabet, dtype=np.uint8)[rng.integers(0, len(alphabet), n)])
    words = [pick(b"abcdefghijklmnopqrstuvwxyz", int(rng.integers(2, 9))) for _ in range(400)]
    text = b" ".join(words[int(i)] for i in rng.integers(0, 400, 40000))
    cases = [("one symbol", b"z" * 5000)]
    for k in range(4):
        cases.append(("acgt+%d" % k, pick(b"ACGT", 80000 + k)))
        cases.append(("two symbols+%d" % k, pick(b"ab", 30000 + k)))
    for k in range(2):
        cases.append(("digits+%d" % k, pick(b"0123456789,.", 30000 + k)))
        cases.append(("16 symbols+%d" % k, bytes(rng.integers(0, 16, 40000 + k, dtype=np.uint8))))
        cases.append(("17 symbols+%d" % k, bytes(rng.integers(0, 17, 40000 + k, dtype=np.uint8))))
        cases.append(("text+%d" % k, text[:150000 + k]))
        cases.append(("markov+%d" % k, datagen.block(0, 100000 + k).tobytes()))
    cases.append(("5 symbols", pick(b"ACGTN", 50001)))
    cases.append(("239 absent", bytes(rng.integers(0, 17, 2000, dtype=np.uint8)) + text[:3000]))
    cases.append(("random", bytes(rng.integers(0, 256, 40000, dtype=np.uint8))))
    cases.append(("records", datagen.block(2, 100000).tobytes()))
    cases.append(("mostly zeros", datagen.block(4, 100000).tobytes()))
    cases.append(("short", text[:1023]))
    cases.append(("min length", text[:1024]))
    cases.append(("few savings", bytes(rng.integers(0, 200, 60000, dtype=np.uint8))))
    return cases
