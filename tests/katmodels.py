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


def _encode_alphabet(bs, alphabet):
    """EntropyUtils.encodeAlphabet :38-75 (alphabet: ascending symbols)"""
    n = len(alphabet)
    if n == 0:
        bs.write(0, 1); bs.write(1, 1)
    elif n == 256:
        bs.write(0, 1); bs.write(0, 1)
    else:
        bs.write(1, 1)
        masks = [0] * 32
        for a in alphabet:
            masks[a >> 3] |= 1 << (a & 7)
        last = alphabet[-1] >> 3
        bs.write(last, 5)
        for i in range(last + 1):
            bs.write(masks[i], 8)


def _exp_golomb_signed(bs, v):
    """ExpGolombEncoder.encodeByte, signed form (:123-132; the cache row is the plain code: |v| + 1 in Exp-Golomb, then the sign)"""
    if v == 0:
        bs.write(1, 1)
        return
    n = abs(v) + 1
    length = n.bit_length()
    bs.write(0, length - 1)
    bs.write(n, length)
    bs.write(1 if v < 0 else 0, 1)


def huffman_encode(data):
    """K/entropy/HuffmanEncoder.java for blocks of ONE chunk (< 65536 bytes) whose minimum-redundancy code lengths do not depend
    on how ties between equal weights are broken (powers of two, all equal, ...): the lengths come from a textbook two-queue
    Huffman construction here, NOT from the in-place Moffat-Katajainen passes the encoder (and oracle/kzo_huffman.c) use, and
    must not exceed 12.  updateFrequencies :103-176 (alphabet, canonical codes HuffmanCommon.java:71-111, signed Exp-Golomb
    deltas of the lengths starting from 2), encode :380-413, encodeChunk :417-489 (four fragments, bit counts as varints).
    Returns (bytes, number of bits)."""
    import heapq
    bs = _Bits()
    count = len(data)
    assert count < 65536
    if count < 32:                                                            # :399-401
        for b in data:
            bs.write(b, 8)
        return bs.bytes(), bs.n
    freqs = [0] * 256
    for b in data:
        freqs[b] += 1
    alphabet = [i for i in range(256) if freqs[i]]
    _encode_alphabet(bs, alphabet)
    sizes = {}
    if len(alphabet) == 1:
        sizes[alphabet[0]] = 1
    else:
        heap = [(freqs[a], i, (a,)) for i, a in enumerate(alphabet)]
        heapq.heapify(heap)
        depth = {a: 0 for a in alphabet}
        tick = len(heap)
        while len(heap) > 1:
            f1, _, s1 = heapq.heappop(heap)
            f2, _, s2 = heapq.heappop(heap)
            for a in s1 + s2:
                depth[a] += 1
            heapq.heappush(heap, (f1 + f2, tick, s1 + s2))
            tick += 1
        sizes = depth
        assert max(sizes.values()) <= 12
    prev = 2
    for a in alphabet:                                                        # :163-173
        _exp_golomb_signed(bs, sizes[a] - prev)
        prev = sizes[a]
    if len(alphabet) == 1:                                                    # :407-408: a one-symbol chunk is its header
        return bs.bytes(), bs.n
    codes, code, cur = {}, 0, None
    for a in sorted(alphabet, key=lambda x: (sizes[x], x)):                   # canonical: by (length, symbol)
        if cur is None:
            cur = sizes[a]
        code <<= sizes[a] - cur
        cur = sizes[a]
        codes[a] = code
        code += 1
    frag = count // 4
    parts = []
    for j in range(4):
        fb = _Bits()
        for b in data[j * frag:(j + 1) * frag]:
            fb.write(codes[b], sizes[b])
        parts.append(fb)
    for fb in parts:
        _write_varint(bs, fb.n)
    for fb in parts:
        if fb.n:
            bs.write(fb.v, fb.n)
    for b in data[4 * frag:]:
        bs.write(b, 8)
    return bs.bytes(), bs.n


def lz_decode(src, out_cap):
    """K/transform/LZCodec.java inverseV6 :617-756 + readLength :241-258, as a pure-Python reader of the LZ / LZX frame: three little-
    endian lengths (offset of the token stream = 13 + literal bytes, token bytes, match-index bytes; the match lengths fill the rest), a flag byte (bit 0: the 2^24 window, bits 1..3:
    minMatch - 2), the literals, then the token, match-index and match-length streams.  Returns the decoded bytes or None."""
    count = len(src)
    if count == 0:
        return b""
    if count < 13:
        return None
    tk_len = int.from_bytes(src[0:4], "little", signed=True)
    m_idx_len = int.from_bytes(src[4:8], "little", signed=True)
    m_len_len = int.from_bytes(src[8:12], "little", signed=True)
    if tk_len < 13 or tk_len > count or m_idx_len < 0 or m_len_len < 0 or m_idx_len > count - tk_len or m_len_len > count - tk_len - m_idx_len:
        return None
    tk = tk_len
    mi = tk + m_idx_len
    ml = mi + m_len_len
    src_end = tk - 13
    max_dist = (1 << 16) - 2 if (src[12] & 1) == 0 else (1 << 24) - 2          # MAX_DISTANCE1 / MAX_DISTANCE2 :152-153
    min_match = ((src[12] >> 1) & 7) + 2
    pos = 13
    dst = bytearray()
    repd0 = repd1 = count

    def read_length(p):
        r = src[p]; p += 1
        if r < 254:
            return r, p
        if r == 254:
            return r + (src[p] << 8) + src[p + 1], p + 2
        return r + (src[p] << 16) + (src[p + 1] << 8) + src[p + 2], p + 3

    while True:
        token = src[tk]; tk += 1
        if token >= 32:
            if token >= 0xE0:
                n, pos = read_length(pos)
                lit = 7 + n
            else:
                lit = token >> 5
            if lit > out_cap - len(dst) or lit > tk_len - pos:
                return None
            dst += src[pos:pos + lit]
            pos += lit
            if pos >= src_end + 13:                                            # srcIdx >= srcEnd, srcEnd = token stream - 13 + the 13-byte header
                break
        f = token & 0x18
        if f == 0:
            mlen = token & 3
            if mlen == 3:
                n, ml = read_length(ml)
                mlen += min_match + n
            else:
                mlen += min_match
            dist = repd0 if (token & 4) == 0 else repd1
        else:
            mlen = token & 7
            if mlen == 7:
                n, ml = read_length(ml)
                mlen += min_match + n
            else:
                mlen += min_match
            dist = src[mi]; mi += 1
            if f == 0x18:
                dist = (dist << 16) | (src[mi] << 8) | src[mi + 1]; mi += 2
            elif f == 0x10:
                dist = (dist << 8) | src[mi]; mi += 1
        repd1, repd0 = repd0, dist
        ref = len(dst) - dist
        if ref < 0 or dist > max_dist or len(dst) + mlen > out_cap:
            return None
        for i in range(mlen):
            dst.append(dst[ref + i])
    return bytes(dst) if pos == tk_len else None

