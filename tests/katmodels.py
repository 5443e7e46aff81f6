"""Tiny pure-Python models used for known answers: each is written directly from the cited reference lines (not from
oracle/*.c), with Java's 64-bit long arithmetic made explicit, and is only ever run on inputs of a few bytes."""

M64 = (1 << 64) - 1


def fpaq_encode(data):
    """K/entropy/FPAQEncoder.java: constructor :84-97 (low = 0, high = TOP, every probability PSCALE >> 1), encode :128-173,
    encodeBit :182-199, flush :208-213, dispose :232-238.  One chunk (len(data) <= 4 MiB).  Returns the bit string as bytes:
    varint(chunk bytes) | chunk bytes | 56 bits of low | 0xFFFFFF."""
    TOP, MASK_24_56, MASK_0_24, MASK_0_32, PSCALE = 0x00FFFFFFFFFFFFFF, 0x00FFFFFFFF000000, 0xFFFFFF, 0xFFFFFFFF, 65536
    low, high = 0, TOP
    probs = [[PSCALE >> 1] * 256 for _ in range(4)]
    p = probs[0]
    out = bytearray()

    def encode_bit(bit, idx):
        nonlocal low, high
        split = ((((high - low) & M64) >> 8) * p[idx] & M64) >> 8           # :185
        if bit == 0:
            low = (low + split + 1) & M64                                     # :189
            p[idx] -= p[idx] >> 6                                             # :190
        else:
            high = (low + split) & M64                                        # :192
            p[idx] -= (p[idx] - PSCALE + 64) >> 6                             # :193 (arithmetic shift of a negative int)
        while ((low ^ high) & MASK_24_56) == 0:                               # :197-198
            out.extend(((high >> 24) & 0xFFFFFFFF).to_bytes(4, "big"))        # flush :209-212
            low = (low << 32) & M64
            high = ((high << 32) & M64) | MASK_0_32

    for val in data:
        bits = val + 256
        encode_bit(val & 0x80, 1)
        for k in range(7, 0, -1):                                             # :150-156: contexts bits >> 7 .. bits >> 1
            encode_bit(val & (1 << (k - 1)), bits >> k)
        p = probs[val >> 6]                                                   # :157
    n = len(out)
    header = bytearray()                                                      # EntropyUtils.writeVarInt :259-276
    v = n
    while v >= 128:
        header.append(0x80 | (v & 0x7F))
        v >>= 7
    header.append(v)
    tail = ((low | MASK_0_24) & ((1 << 56) - 1)).to_bytes(7, "big")           # dispose :237
    return bytes(header) + bytes(out) + tail
