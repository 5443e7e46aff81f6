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


class _Bits:
    """MSB-first bit string (K/bitstream/DefaultOutputBitStream.java:103-123)"""

    def __init__(self):
        self.v, self.n = 0, 0

    def write(self, value, count):
        self.v = (self.v << count) | (value & ((1 << count) - 1))
        self.n += count

    def bytes(self):
        pad = (-self.n) % 8
        return (self.v << pad).to_bytes((self.n + 7) // 8, "big")


def _write_varint(bs, v):                                                     # EntropyUtils.writeVarInt :259-276
    while v >= 128:
        bs.write(0x80 | (v & 0x7F), 8)
        v >>= 7
    bs.write(v, 8)


def _normalize(freqs, total, scale):
    """EntropyUtils.normalizeFrequencies :141-250 on a 256-entry list (modified in place); returns the alphabet"""
    if total == 0:
        return []
    if total == scale:
        return [i for i in range(256) if freqs[i]]
    alphabet, sum_scaled, sum_freq, idx_max = [], 0, 0, 0
    for i in range(256):
        f = freqs[i]
        if f == 0:
            continue
        sf = f * scale
        scaled = 1 if sf <= total else (sf + (total >> 1)) // total
        alphabet.append(i)
        sum_scaled += scaled
        freqs[i] = scaled
        sum_freq += f
        if scaled > freqs[idx_max]:
            idx_max = i
        if sum_freq >= total:
            break
    if not alphabet:
        return []
    if len(alphabet) == 1:
        freqs[alphabet[0]] = scale
        return alphabet
    if sum_scaled == scale:
        return alphabet
    delta = sum_scaled - scale
    err_thr = freqs[idx_max] >> 4
    if abs(delta) <= err_thr:
        freqs[idx_max] -= delta
        return alphabet
    if delta < 0:
        delta += err_thr
        freqs[idx_max] += err_thr
    else:
        delta -= err_thr
        freqs[idx_max] -= err_thr
    inc = -1 if delta > 0 else 1
    delta = abs(delta)
    rnd = 0
    while True:
        rnd += 1
        if not (rnd < 6 and delta > 0):
            break
        adjustments = 0
        for idx in alphabet:
            if freqs[idx] <= 2:
                continue
            freqs[idx] += inc
            adjustments += 1
            delta -= 1
            if delta == 0:
                break
        if adjustments == 0:
            break
    freqs[idx_max] = max(freqs[idx_max] - delta, 1)
    return alphabet


def ans0_encode(data):
    """K/entropy/ANSRangeEncoder.java, order 0, logRange 12, 16 KiB chunks (:126-160): encode :263-305, rebuildStatistics /
    updateFrequencies :164-200, encodeHeader :211-252, EntropyUtils.encodeAlphabet :38-75, Symbol.reset :473-496, encodeChunk
    :337-407, encodeSymbol :315-328.  Returns (bytes, number of bits)."""
    bs = _Bits()
    count = len(data)
    if count <= 32:                                                           # :267-270: tiny blocks are stored
        for b in data:
            bs.write(b, 8)
        return bs.bytes(), bs.n
    lr, ANS_TOP = 12, 1 << 15
    for start in range(0, count, 16384):
        chunk = data[start:start + 16384]
        freqs = [0] * 256
        for b in chunk:
            freqs[b] += 1
        bs.write(lr - 8, 3)                                                   # :167
        alphabet = _normalize(freqs, len(chunk), 1 << lr)
        sym, cum = {}, 0
        if alphabet:
            for i in range(256):                                              # :177-186 (ascending symbol order)
                if freqs[i] == 0:
                    continue
                f = freqs[i]
                if f >= 1 << lr:
                    f = (1 << lr) - 1
                x_max = ((ANS_TOP >> lr) << 16) * f
                cmpl = (1 << lr) - f
                if f < 2:
                    inv_freq, inv_shift, bias = 0xFFFFFFFF, 32, cum + (1 << lr) - 1
                else:
                    shift = 0
                    while f > (1 << shift):
                        shift += 1
                    inv_freq = (((1 << (shift + 31)) + f - 1) // f) & 0xFFFFFFFF
                    inv_shift, bias = 32 + shift - 1, cum
                sym[i] = (x_max, bias, cmpl, inv_shift, inv_freq)
                cum += freqs[i]
        # encodeAlphabet :38-75
        n_alpha = len(alphabet)
        if n_alpha == 0:
            bs.write(0, 1); bs.write(1, 1)                                    # FULL_ALPHABET = 0, ALPHABET_0 = 1 (:31-34)
        elif n_alpha == 256:
            bs.write(0, 1); bs.write(0, 1)                                    # ALPHABET_256 = 0
        else:
            bs.write(1, 1)
            masks = [0] * 32
            for a in alphabet:
                masks[a >> 3] |= 1 << (a & 7)
            last = alphabet[-1] >> 3
            bs.write(last, 5)
            for i in range(last + 1):
                bs.write(masks[i], 8)
        if n_alpha > 1:                                                       # encodeHeader :221-250
            chk = 8 if n_alpha >= 64 else 6
            llr = 3
            while (1 << llr) <= lr:
                llr += 1
            i = 1
            while i < n_alpha:
                endj = min(i + chk, n_alpha)
                mx = max(freqs[alphabet[j]] - 1 for j in range(i, endj))
                log_max = 0
                while (1 << log_max) <= mx:
                    log_max += 1
                bs.write(log_max, llr)
                if log_max:
                    for j in range(i, endj):
                        bs.write(freqs[alphabet[j]] - 1, log_max)
                i += chk
        if n_alpha <= 1:                                                      # :292-295: nothing more for a one-symbol chunk
            continue
        # encodeChunk :337-407: the chunk is coded backwards into the END of a buffer by four interleaved states
        buf = bytearray()                                                     # built reversed: buf[0] is the LAST byte of the buffer
        end4 = len(chunk) & -4
        for i in range(len(chunk) - 1, end4 - 1, -1):
            buf.append(chunk[i])
        st = [ANS_TOP] * 4

        def enc(state, s):
            x_max, bias, cmpl, inv_shift, inv_freq = sym[s]
            if state >= x_max:                                                # :316-321: two bytes leave, low byte at the higher address
                buf.append(state & 0xFF)
                buf.append((state >> 8) & 0xFF)
                state >>= 16
            q = (state * inv_freq) >> inv_shift
            return state + bias + q * cmpl                                    # :327

        i = end4 - 1
        while i > 0:                                                          # :352-357 (i > start)
            st[0] = enc(st[0], chunk[i])
            st[1] = enc(st[1], chunk[i - 1])
            st[2] = enc(st[2], chunk[i - 2])
            st[3] = enc(st[3], chunk[i - 3])
            i -= 4
        _write_varint(bs, len(buf))                                           # :396
        for s in st:
            bs.write(s, 32)
        for b in reversed(buf):
            bs.write(b, 8)
    return bs.bytes(), bs.n
