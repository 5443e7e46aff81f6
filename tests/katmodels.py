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


# ---- round 5: the Huffman code-length passes themselves, from the Java ----
def normalize_frequencies_java(freqs, alen, total, scale):
    """EntropyUtils.normalizeFrequencies :141-250 with an alphabet array of `alen` entries (HuffmanEncoder.java:252-262 calls it
    with compacted arrays of `count` entries, ANSRangeEncoder with 256).  freqs is modified in place; returns the alphabet list.
    The `totalFreq == scale` shortcut reads freqs[0..255] whatever alen is (:159-165): an IndexError here is Java's
    ArrayIndexOutOfBoundsException."""
    if alen == 0 or total == 0:
        return []
    if total == scale:
        return [i for i in range(256) if freqs[i] != 0]
    alphabet, sum_scaled, sum_freq, idx_max = [], 0, 0, 0
    for i in range(alen):
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


def huffman_phase1(data, n):                                                  # computeInPlaceSizesPhase1 :317-339
    s = r = 0
    for t in range(n - 1):
        total = 0
        for _ in range(2):
            if s >= n or (r < t and data[r] < data[s]):
                total += data[r]
                data[r] = t
                r += 1
                continue
            total += data[s]
            if s > t:
                data[s] = 0
            s += 1
        data[t] = total


def huffman_phase2(data, n):                                                  # computeInPlaceSizesPhase2 :349-376
    if n < 2:
        return 0
    level_top, depth, i, total_nodes = n - 2, 1, n, 2
    while i > 0:
        k = level_top
        while k > 0 and data[k - 1] >= level_top:
            k -= 1
        internal = level_top - k
        leaves = total_nodes - internal
        for _ in range(leaves):
            i -= 1
            data[i] = depth
        total_nodes = internal << 1
        level_top = k
        depth += 1
    return depth - 1


def huffman_compute_code_lengths(sizes, ranks, count):                        # computeCodeLengths :285-308
    ranks[:count] = sorted(ranks[:count])                                     # Arrays.sort(int[]): (freq << 8) | symbol ascending
    freqs = [0] * 256
    for i in range(count):
        freqs[i] = ranks[i] >> 8
        ranks[i] &= 0xFF
        if freqs[i] == 0:
            return 0
    huffman_phase1(freqs, count)
    max_len = huffman_phase2(freqs, count)
    for i in range(count):
        sizes[ranks[i]] = freqs[i]
    return max_len


def huffman_limit_code_lengths(alphabet, freqs, sizes, ranks, count):         # limitCodeLengths :191-273
    MAXLEN = 12                                                               # HuffmanCommon.MAX_SYMBOL_SIZE_V4
    n = debt = 0
    while sizes[ranks[n]] >= MAXLEN:                                          # (ranks has 256 entries: index `count` exists, :126)
        debt += sizes[ranks[n]] - MAXLEN
        sizes[ranks[n]] = MAXLEN
        n += 1
    ll = [[] for _ in range(6)]                                               # LinkedList: add at the tail, removeFirst
    while n < count:
        idx = MAXLEN - 1 - sizes[ranks[n]]
        if idx >= len(ll) or debt < (1 << idx):
            break
        ll[idx].append(ranks[n])
        n += 1
    idx = len(ll) - 1
    while debt > 0 and idx >= 0:
        if not ll[idx] or debt < (1 << idx):
            idx -= 1
            continue
        r = ll[idx].pop(0)
        sizes[r] += 1
        debt -= 1 << idx
    idx = 0
    while debt > 0 and idx < len(ll):
        if not ll[idx]:
            idx += 1
            continue
        r = ll[idx].pop(0)
        sizes[r] += 1
        debt -= 1 << idx
    if debt > 0:                                                              # :250-270
        f = [freqs[alphabet[i]] for i in range(count)]
        total = sum(f)
        normalize_frequencies_java(f, count, total, 16384 >> 3)               # HuffmanCommon.MAX_CHUNK_SIZE >> 3
        for i in range(count):
            freqs[alphabet[i]] = f[i]
            ranks[i] = (f[i] << 8) | alphabet[i]
        return huffman_compute_code_lengths(sizes, ranks, count)
    return MAXLEN


def huffman_sizes_and_codes(freqs):
    """updateFrequencies :103-160 for a chunk's 256 counts: (alphabet, sizes[256], codes[256]) exactly as the encoder derives them:
    in-place Moffat-Katajainen lengths, the 12-bit limiter with its linked lists and its renormalising fallback, the 8-bit flat
    code when even that fails, canonical codes (HuffmanCommon.generateCanonicalCodes :71-111)."""
    freqs = list(freqs)
    alphabet = [i for i in range(256) if freqs[i] > 0]
    count = len(alphabet)
    sizes, codes = [0] * 256, [0] * 256
    if count == 0:
        return alphabet, sizes, codes
    if count == 1:
        sizes[alphabet[0]] = 1
        return alphabet, sizes, codes
    ranks = [0] * 256
    for i in range(count):
        ranks[i] = (freqs[alphabet[i]] << 8) | alphabet[i]
    max_len = huffman_compute_code_lengths(sizes, ranks, count)
    if max_len == 0:
        raise JavaException("Could not generate Huffman codes: invalid code length 0")
    if max_len > 12:
        max_len = huffman_limit_code_lengths(alphabet, freqs, sizes, ranks, count)
        if max_len == 0:
            raise JavaException("Could not generate Huffman codes: invalid code length 0")
    if max_len > 12:                                                          # :146-155
        for n, a in enumerate(alphabet):
            codes[a] = n
            sizes[a] = 8
    else:                                                                     # generateCanonicalCodes: symbols by (length, symbol)
        order = sorted(ranks[:count], key=lambda x: (sizes[x], x))
        code, cur = 0, sizes[order[0]]
        for a in order:
            if sizes[a] > 12 or sizes[a] < 1:
                raise JavaException("invalid code length")
            code <<= sizes[a] - cur
            cur = sizes[a]
            codes[a] = code
            code += 1
    return alphabet, sizes, codes


def huffman_encode_exact(data):
    """HuffmanEncoder.encode :380-416 on ANY block: 16 KiB chunks, code lengths from huffman_sizes_and_codes.  (bytes, bits)."""
    bs = _Bits()
    pos, end = 0, len(data)
    while pos < end:
        size = min(16384, end - pos)
        chunk = data[pos:pos + size]
        if size < 32:
            for b in chunk:
                bs.write(b, 8)
        else:
            freqs = [0] * 256
            for b in chunk:
                freqs[b] += 1
            alphabet, sizes, codes = huffman_sizes_and_codes(freqs)
            _encode_alphabet(bs, alphabet)
            prev = 2
            for a in alphabet:
                d = sizes[a] - prev
                _exp_golomb_signed(bs, d - 256 if d > 127 else d)             # (byte) cast
                prev = sizes[a]
            if len(alphabet) > 1:
                frag = size // 4
                parts = []
                for j in range(4):
                    fb = _Bits()
                    for b in chunk[j * frag:(j + 1) * frag]:
                        fb.write(codes[b], sizes[b])
                    parts.append(fb)
                for fb in parts:
                    _write_varint(bs, fb.n)
                for fb in parts:
                    if fb.n:
                        bs.write(fb.v, fb.n)
                for b in chunk[4 * frag:]:
                    bs.write(b, 8)
        pos += size
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
            if pos >= src_end:                                                 # srcIdx >= srcEnd (:676), srcEnd = tkIdx - 13 (:647): a literal run that ends within 13
                break                                                          # bytes of the token stream ends the walk; only one that ends AT it succeeds (:754)
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



# =====================================================================================================================
# Round 3: models of the stages whose OUTPUT IS A CHOICE (any valid parse / order round-trips), written from the Java lines
# cited, not from oracle/*.c.  They run on inputs of up to a few hundred KiB.

class JavaException(Exception):
    """the reference would throw here (ArrayIndexOutOfBoundsException): the block fails with ERR_PROCESS_BLOCK"""


def _emit_length(buf, idx, length):                                           # LZCodec.java:211-231
    if length < 254:
        buf[idx] = length
        return idx + 1
    if length < 65536 + 254:
        length -= 254
        buf[idx] = 254
        buf[idx + 1] = (length >> 8) & 0xFF
        buf[idx + 2] = length & 0xFF
        return idx + 3
    length -= 255
    buf[idx] = 255
    buf[idx + 1] = (length >> 16) & 0xFF
    buf[idx + 2] = (length >> 8) & 0xFF
    buf[idx + 3] = length & 0xFF
    return idx + 4


class _JavaBytes(list):
    """byte[] of fixed length: an index past the end throws, like Java"""

    def __setitem__(self, i, v):
        if not 0 <= i < len(self):
            raise JavaException("ArrayIndexOutOfBounds %d / %d" % (i, len(self)))
        list.__setitem__(self, i, v & 0xFF)


def lz_forward(data, extra=False, data_type="UNDEFINED"):
    """K/transform/LZCodec.java LZXCodec.forward :299-597 (LZ: extra = false, 2^16 hash entries; LZX: extra = true, 2^19 entries and the
    second lazy probe), hash :904-911, findMatch :271-287, emitLength :211-231, differentInts :134-137.  data_type = the context's
    "dataType" entry (:343-352).  Returns (applied, bytes); raises JavaException where the reference's fixed tkBuf / mLenBuf overflow."""
    src = bytes(data)
    count = len(src)
    if count == 0:
        return True, b""
    max_enc = (count + 16 if count <= 1024 else count + count // 64) + 2          # getMaxEncodedLength :961-964 (the caller provides it)
    if count < 24:                                                                # MIN_BLOCK_LENGTH :312
        return False, b""
    SEED, MAXD1, MAXD2, MAX_MATCH = 0x1E35A7BD, (1 << 16) - 2, (1 << 24) - 2, 65535 + 254 + 4
    log = 19 if extra else 16
    hashes = [0] * (1 << log)
    min_buf = max(count // 5, 256)                                                # :325
    mbuf, mlenbuf, tkbuf = _JavaBytes([0] * min_buf), _JavaBytes([0] * min_buf), _JavaBytes([0] * min_buf)
    dst = _JavaBytes([0] * (max_enc + 16))
    pad = src + bytes(16)

    def h(i):                                                                     # :904-911: (long << 24) * seed >>> (64 - log)
        v = int.from_bytes(pad[i:i + 8], "little")
        return ((((v << 24) & M64) * SEED) & M64) >> (64 - log)

    def differ(a, b):
        return src[a:a + 4] != src[b:b + 4]

    def find_match(s, r, max_match):                                              # :271-287 (8 bytes at a time, stops below max_match - 7)
        best = 0
        while best + 8 <= max_match:
            x = int.from_bytes(pad[s + best:s + best + 8], "little") ^ int.from_bytes(pad[r + best:r + best + 8], "little")
            if x:
                best += ((x & -x).bit_length() - 1) >> 3
                break
            best += 8
        return best

    src_end = count - 16 - 2
    max_dist = MAXD1 if src_end < 4 * MAXD1 else MAXD2
    dst[12] = 0 if max_dist == MAXD1 else 1
    mm = 4
    if data_type == "DNA":
        mm = 6
    elif data_type == "SMALL_ALPHABET":
        return False, b""
    dst[12] = dst[12] | (((mm - 2) & 7) << 1)
    src_idx = anchor = 0
    dst_idx = 13
    m_idx = mlen_idx = tk_idx = 0
    repd = [count, count]
    rep_idx = 0
    src_inc = 0
    while src_idx < src_end:
        best = 0
        h0 = h(src_idx)
        ref0 = hashes[h0]
        hashes[h0] = src_idx
        s1 = src_idx + 1
        ref = s1 - repd[rep_idx]
        min_ref = max(src_idx - max_dist, 0)
        if ref > min_ref and not differ(ref, s1):                                # repd first :378-387
            best = find_match(s1, ref, min(src_end - s1, MAX_MATCH))
        else:
            ref = s1 - repd[rep_idx ^ 1]
            if ref > min_ref and not differ(ref, s1):
                best = find_match(s1, ref, min(src_end - s1, MAX_MATCH))
        if best < mm:
            ref = ref0                                                            # :391-395
            if ref > min_ref and not differ(ref, src_idx):
                best = find_match(src_idx, ref, min(src_end - src_idx, MAX_MATCH))
            if best < mm:                                                         # :398-403
                src_idx = s1 + (src_inc >> 6)
                src_inc += 1
                rep_idx = 0
                continue
            if ref != src_idx - repd[0] and ref != src_idx - repd[1]:             # :405-443 lazy probes
                h1 = h(s1)
                ref1 = hashes[h1]
                hashes[h1] = s1
                if ref1 > min_ref + 1 and not differ(ref1 + best - 3, s1 + best - 3):
                    b1 = find_match(s1, ref1, min(src_end - s1, MAX_MATCH))
                    if b1 >= best:
                        ref, best, src_idx = ref1, b1, s1
                if extra:
                    s2 = s1 + 1
                    h2 = h(s2)
                    ref2 = hashes[h2]
                    hashes[h2] = s2
                    if ref2 > min_ref + 2 and not differ(ref2 + best - 3, s2 + best - 3):
                        b2 = find_match(s2, ref2, min(src_end - s2, MAX_MATCH))
                        if b2 >= best:
                            ref, best, src_idx = ref2, b2, s2
            while src_idx > anchor and ref > min_ref and src[src_idx - 1] == src[ref - 1]:   # :446-450 extend backwards
                best += 1
                ref -= 1
                src_idx -= 1
            if best > MAX_MATCH:                                                  # :452-456
                ref += best - MAX_MATCH
                src_idx += best - MAX_MATCH
                best = MAX_MATCH
        else:                                                                     # :457-466 repeat match found at srcIdx + 1
            if best >= MAX_MATCH or src[src_idx] != src[ref - 1]:
                src_idx += 1
                hashes[h(src_idx)] = src_idx
            else:
                best += 1
                ref -= 1
        src_inc = 0
        dist = src_idx - ref
        if dist == repd[0]:
            token, th = 0x00, 3
        elif dist == repd[1]:
            token, th = 0x04, 3
        else:                                                                     # :492-500
            mbuf[m_idx] = dist >> 16
            inc1 = 1 if dist >= 65536 else 0
            m_idx += inc1
            mbuf[m_idx] = dist >> 8
            inc2 = 1 if dist >= 256 else 0
            m_idx += inc2
            mbuf[m_idx] = dist
            m_idx += 1
            token, th = (inc1 + inc2 + 1) << 3, 7
        mlen = best - mm
        if mlen >= th:
            token += th
            mlen_idx = _emit_length(mlenbuf, mlen_idx, mlen - th)
        else:
            token += mlen
        repd[1] = repd[0]
        repd[0] = dist
        rep_idx = 1
        lit = src_idx - anchor
        if lit == 0:
            tkbuf[tk_idx] = token
            tk_idx += 1
        else:
            if lit >= 7:
                if lit >= 1 << 24:
                    return False, b""
                tkbuf[tk_idx] = (7 << 5) | token
                tk_idx += 1
                dst_idx = _emit_length(dst, dst_idx, lit - 7)
            else:
                tkbuf[tk_idx] = (lit << 5) | token
                tk_idx += 1
            for k in range(lit):
                dst[dst_idx + k] = src[anchor + k]
            dst_idx += lit
        if m_idx >= len(mbuf) - 8:                                                # :538-549 (tkBuf is never grown)
            mbuf.extend([0] * ((len(mbuf) * 3) // 2 - len(mbuf)))
            if mlen_idx >= len(mlenbuf) - 4:
                mlenbuf.extend([0] * ((len(mlenbuf) * 3) // 2 - len(mlenbuf)))
        anchor = src_idx + best                                                   # :552-564 hash fill
        while src_idx + 4 < anchor:
            src_idx += 4
            for k in (3, 2, 1, 0):
                hashes[h(src_idx - k)] = src_idx - k
        src_idx += 1
        while src_idx < anchor:
            hashes[h(src_idx)] = src_idx
            src_idx += 1
    lit = count - anchor                                                          # :567-596
    if dst_idx + lit + tk_idx + m_idx + mlen_idx >= count:
        return False, b""
    if lit >= 7:
        tkbuf[tk_idx] = 7 << 5
        tk_idx += 1
        dst_idx = _emit_length(dst, dst_idx, lit - 7)
    else:
        tkbuf[tk_idx] = lit << 5
        tk_idx += 1
    for k in range(lit):
        dst[dst_idx + k] = src[anchor + k]
    dst_idx += lit
    out = bytearray(dst[:dst_idx])
    out[0:4] = dst_idx.to_bytes(4, "little")
    out[4:8] = tk_idx.to_bytes(4, "little")
    out[8:12] = m_idx.to_bytes(4, "little")
    out += bytes(tkbuf[:tk_idx]) + bytes(mbuf[:m_idx]) + bytes(mlenbuf[:mlen_idx])
    return len(out) <= count - count // 100, bytes(out)


def srt_forward(data):
    """K/transform/SRT.java forward :73-168, preprocess :266-302 (shell sort by (freq desc, symbol asc)), encodeHeader :312-325."""
    src = bytes(data)
    count = len(src)
    freqs, r2s, s2r = [0] * 256, [0] * 256, [0] * 256
    i = b = 0
    while i < count:                                                              # :100-117 first appearances and run-wise counts
        c = src[i]
        if freqs[c] == 0:
            r2s[b] = c
            s2r[c] = b
            b += 1
        j = i + 1
        while j < count and src[j] == c:
            j += 1
        freqs[c] += j - i
        i = j
    symbols = [s for s in range(256) if freqs[s] > 0]                             # preprocess
    nb = len(symbols)
    hgap = 4
    while hgap < nb:
        hgap = hgap * 3 + 1
    while True:
        hgap //= 3
        for i in range(hgap, nb):
            t = symbols[i]
            bb = i - hgap
            while bb >= 0 and (freqs[symbols[bb]] < freqs[t] or (freqs[t] == freqs[symbols[bb]] and t < symbols[bb])):
                symbols[bb + hgap] = symbols[bb]
                bb -= hgap
            symbols[bb + hgap] = t
        if hgap == 1:
            break
    buckets = [0] * 256
    pos = 0
    for s in symbols:
        buckets[s] = pos
        pos += freqs[s]
    header = bytearray()
    for f in freqs:                                                               # encodeHeader
        while f >= 128:
            header.append(0x80 | (f & 0x7F))
            f >>= 7
        header.append(f)
    out = bytearray(count)
    i = 0
    while i < count:                                                              # :133-163
        c = src[i]
        r = s2r[c]
        p = buckets[c]
        out[p] = r
        p += 1
        if r != 0:
            while r != 0:
                r2s[r] = r2s[r - 1]
                s2r[r2s[r]] = r
                r -= 1
            r2s[0] = c
            s2r[c] = 0
        i += 1
        while i < count and src[i] == c:
            out[p] = 0
            p += 1
            i += 1
        buckets[c] = p
    return bytes(header) + bytes(out)


def sbrt_forward(data, mode):
    """K/transform/SBRT.java forward :87-151; mode 1 = MTF, 2 = RANK, 3 = TIMESTAMP (:36-38): m1 masks the index, m2 the previous
    occurrence, RANK halves their sum."""
    src = bytes(data)
    m1 = 0 if mode == 3 else -1
    m2 = 0 if mode == 1 else -1
    s = 1 if mode == 2 else 0
    p, q = [0] * 256, [0] * 256
    s2r, r2s = list(range(256)), list(range(256))
    out = bytearray(len(src))
    for i, c in enumerate(src):
        r = s2r[c]
        out[i] = r
        qc = ((i & m1) + (p[c] & m2)) >> s
        p[c] = i
        q[c] = qc
        while r > 0 and q[r2s[r - 1]] <= qc:
            r2s[r] = r2s[r - 1]
            s2r[r2s[r]] = r
            r -= 1
        r2s[r] = c
        s2r[c] = r
    return bytes(out)


_DNA, _NUMERIC = b"acgntuACGNTU", b"0123456789+-*/=,.:; "
_BASE64 = b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/"


def detect_simple_type(count, freqs0):
    """K/Global.java detectSimpleType :556-605"""
    if count == 0:
        return "UNDEFINED"
    if sum(freqs0[c] for c in _DNA) > count - count // 12:
        return "DNA"
    if sum(freqs0[c] for c in _NUMERIC) == count:
        return "NUMERIC"
    if (1 if freqs0[0x3D] == 1 else 0) + sum(freqs0[c] for c in _BASE64) == count:
        return "BASE64"
    present = sum(1 for f in freqs0 if f > 0)
    if present == 256:
        return "BIN"
    if present <= 4:
        return "SMALL_ALPHABET"
    return "UNDEFINED"


def alias_forward(data, data_type="UNDEFINED", only_dna=False):
    """K/transform/AliasCodec.java forward :78-278 with a context (so the detected type is stored back); Alias.compareTo :476-489
    (TreeSet order: frequency descending, then value descending); Global.computeHistogramOrder1 :341-420 as called here (its first
    lane starts from prv = 0: the pair (0, src[0]) is counted too).  Returns (applied, bytes, dataType left in the context)."""
    src = bytes(data)
    count = len(src)
    if count == 0:
        return True, b"", data_type
    if count < 1024:
        return False, b"", data_type
    dt = data_type
    if dt in ("MULTIMEDIA", "UTF8", "EXE", "BIN"):
        return False, b"", dt
    if only_dna and dt not in ("UNDEFINED", "DNA"):
        return False, b"", dt
    freqs0 = [0] * 256
    for c in src:
        freqs0[c] += 1
    absent = [i for i in range(256) if freqs0[i] == 0]
    n0 = len(absent)
    if n0 < 16:
        return False, b"", dt
    left = data_type
    if dt == "UNDEFINED":
        dt = detect_simple_type(count, freqs0)
        if dt != "UNDEFINED":
            left = dt
        if dt != "DNA" and only_dna:
            return False, b"", left
    out = bytearray()
    si = 0
    if n0 >= 240:
        out.append(n0)
        if n0 == 255:
            out.append(src[0])
            out += count.to_bytes(4, "little")
            si = count
        else:
            map8 = {}
            for i in range(256):
                if freqs0[i]:
                    out.append(i)
                    map8[i] = len(map8)
            if n0 >= 252:
                out.append(count & 3)
                for _ in range(count & 3):
                    out.append(src[si])
                    si += 1
                while si < count:
                    out.append((map8[src[si]] << 6) | (map8[src[si + 1]] << 4) | (map8[src[si + 2]] << 2) | map8[src[si + 3]])
                    si += 4
            else:
                out.append(count & 1)
                if count & 1:
                    out.append(src[si])
                    si += 1
                while si < count:
                    out.append((map8[src[si]] << 4) | map8[src[si + 1]])
                    si += 2
    else:
        freqs1 = {}
        prv = 0
        for c in src:
            freqs1[(prv << 8) | c] = freqs1.get((prv << 8) | c, 0) + 1
            prv = c
        order = sorted(freqs1.items(), key=lambda kv: (-kv[1], -kv[0]))           # TreeSet.pollFirst order
        if len(order) < n0:
            n0 = len(order)
            if n0 < 16:
                return False, b"", left
        map16 = {}
        savings = 0
        out += bytes([n0, 0])
        for i in range(n0):
            val, f = order[i]
            savings += f
            map16[val] = absent[i] | 0x200
            out += bytes([val >> 8, val & 0xFF, absent[i]])
        if savings < count // 20:
            return False, b"", left
        src_end = count - 1
        while si < src_end:
            alias = map16.get((src[si] << 8) | src[si + 1], src[si] | 0x100)
            out.append(alias & 0xFF)
            si += alias >> 8
        if si != src_end + 1:
            out[1] = 1
            out.append(src[si])
            si += 1
    return len(out) < count, bytes(out), left


def _utf_len_seq(b):                                                              # UTFCodec.java LEN_SEQ :34-43, by rule
    if b < 0x80:
        return 1
    if 0xC2 <= b <= 0xDF:
        return 2
    if 0xE0 <= b <= 0xEF:
        return 3
    if 0xF0 <= b <= 0xF4:
        return 4
    return 0


def _utf_pack(src, i):                                                            # UTFCodec.pack :437-466, SIZES :32
    b0 = src[i]
    s = (1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 2, 2, 3, 4)[b0 >> 4]
    if s == 1:
        return 1, b0
    if s == 2:
        return 2, (1 << 19) | (b0 << 8) | src[i + 1]
    if s == 3:
        return 3, (2 << 19) | ((b0 & 0x0F) << 12) | ((src[i + 1] & 0x3F) << 6) | (src[i + 2] & 0x3F)
    if s == 4:
        return 4, (4 << 19) | ((b0 & 0x07) << 18) | ((src[i + 1] & 0x3F) << 12) | ((src[i + 2] & 0x3F) << 6) | (src[i + 3] & 0x3F)
    return 0, 0


def _utf_validate(block, start, count):                                           # UTFCodec.validate :313-434
    freqs0 = [0] * 256
    freqs1 = {}
    prv = 0
    end = start + count
    end4 = start + (count & -4)

    def bad1():
        return freqs0[0xC0] + freqs0[0xC1] + sum(freqs0[0xF5:0x100]) != 0

    i = start
    while i < end4:
        for k in range(4):
            cur = block[i + k]
            freqs0[cur] += 1
            freqs1[(prv, cur)] = freqs1.get((prv, cur), 0) + 1
            prv = cur
        if (i & 0x0FFF) == start and bad1():
            return False
        i += 4
    if end4 != end:
        for i in range(end4, end):
            cur = block[i]
            freqs0[cur] += 1
            freqs1[(prv, cur)] = freqs1.get((prv, cur), 0) + 1
            prv = cur
        if bad1():
            return False
    sum2 = 0
    f1 = lambda a, b: freqs1.get((a, b), 0)
    for i in range(256):
        s1 = 0
        if i < 0xA0 or i > 0xBF:
            s1 += f1(0xE0, i)
        if i < 0x80 or i > 0x9F:
            s1 += f1(0xED, i)
        if i < 0x90 or i > 0xBF:
            s1 += f1(0xF0, i)
        if i < 0x80 or i > 0x8F:
            s1 += f1(0xF4, i)
        if i < 0x80 or i > 0xBF:
            s1 += sum(f1(j, i) for j in range(0xC2, 0xE0)) + sum(f1(j, i) for j in range(0xE1, 0xED))
            s1 += f1(0xF1, i) + f1(0xF2, i) + f1(0xF3, i) + f1(0xEE, i) + f1(0xEF, i)
        else:
            sum2 += freqs0[i]
        if s1:
            return False
    return sum2 >= count // 8


def utf_forward(data, data_type="UNDEFINED"):
    """K/transform/UTFCodec.java forward :68-218 with a context; the sort (QuickSort with SymbolComparator :553-565) is a total order:
    frequency ascending, then symbol ascending, read backwards.  Returns (applied, bytes, dataType left in the context)."""
    src = bytes(data)
    count = len(src)
    if count == 0:
        return True, b"", data_type
    if count < 1024:
        return False, b"", data_type
    if data_type not in ("UNDEFINED", "UTF8"):
        return False, b"", data_type
    must_validate = data_type != "UTF8"
    src_end = count - 4
    start = 0
    if src[0] == 0xEF and src[1] == 0xBB and src[2] == 0xBF:
        start = 3
    else:
        while start < 4 and _utf_len_seq(src[start]) == 0:
            start += 1
    if must_validate and not _utf_validate(src, start, src_end - start):
        return False, b"", data_type
    left = "UTF8"
    alias_map = {}
    syms = []
    res = True
    i = start
    pad = src + bytes(8)
    while i < src_end:
        s, val = _utf_pack(pad, i)
        res = s != 0
        res = res and (s != 3 or 0x80 <= pad[i + 2] <= 0xBF)
        res = res and (s != 4 or (((pad[i + 2] << 8) | pad[i + 3]) & 0xC0C0) == 0x8080)
        if alias_map.get(val, 0) == 0:
            syms.append(val)
            res = res and len(syms) < 32768
        if not res:
            break
        alias_map[val] = alias_map.get(val, 0) + 1
        i += s
    n = len(syms)
    max_target = count - count // 10
    if not res or n == 0 or 3 * n + 6 >= max_target:
        return False, b"", left
    ranked = sorted(syms, key=lambda v: (alias_map[v], v), reverse=True)
    out = bytearray([0, 0, n >> 8, n & 0xFF])
    estimate = 4 + 6
    alias = {}
    for r, v in enumerate(ranked):
        out += bytes([(v >> 16) & 0xFF, (v >> 8) & 0xFF, v & 0xFF])
        estimate += alias_map[v] if r < 128 else 2 * alias_map[v]
        alias[v] = r if r < 128 else (0x10080 | ((r << 1) & 0xFF00) | (r & 0x7F))
    if estimate >= max_target:
        return False, b"", left
    out += src[:start]
    i = start
    while i < src_end:
        s, val = _utf_pack(pad, i)
        i += s
        a = alias[val]
        out.append(a & 0xFF)
        if a >> 16:
            out.append((a >> 8) & 0xFF)
    out[0] = start
    out[1] = i - src_end
    out += src[i:src_end + 4]
    return len(out) < max_target, bytes(out), left


# ---- MM = FSDCodec.forward (K/transform/FSDCodec.java:63-246): which step and which coding are CHOICES made from sampled entropies ----
import math as _math

_LOG2_4096 = [0] + [int(_math.floor(4096.0 * _math.log2(x) + 0.5)) for x in range(1, 257)]   # Global.LOG2_4096 (:103-127) = round(4096 log2 x)


def log2_1024(x):
    """K/Global.java log2_1024 :221-235"""
    if x < 256:
        return (_LOG2_4096[x] + 2) >> 2
    log = x.bit_length() - 1
    if x & (x - 1) == 0:
        return log << 10
    return (log - 7) * 1024 + ((_LOG2_4096[x >> (log - 7)] + 2) >> 2)


def first_order_entropy_1024(length, histo):
    """K/Global.java computeFirstOrderEntropy1024 :440-456 (Java long arithmetic, >> 3 per symbol, integer division at the end)"""
    if length == 0:
        return 0
    ll = log2_1024(length)
    total = 0
    for h in histo:
        if h:
            total += (h * (ll - log2_1024(h))) >> 3
    return total // length


def magic_type(src):
    """K/Magic.java getType :154-183"""
    if len(src) < 4:
        return 0
    key = int.from_bytes(src[:4], "big")
    if key & ~0x0F & 0xFFFFFFFF == 0xFFD8FFE0:
        return key
    if (key >> 8) in (0x425A68, 0x494433):
        return key >> 8
    if key in (0x47494638, 0x25504446, 0x504B0304, 0x377ABCAF, 0x89504E47, 0x7F454C46, 0xFEEDFACE, 0xCEFAEDFE, 0xFEEDFACF, 0xCFFAEDFE,
               0x28B52FFD, 0x81CFB2CE, 0x4D534346, 0x52494646, 0x664C6143, 0xFD377A58, 0x4B414E5A, 0x52617221):
        return key
    key16 = key >> 16
    if key16 in (0x1F8B, 0x424D, 0x4D5A):
        return key16
    if key16 in (0x5034, 0x5035, 0x5036) and ((key >> 8) & 0xFF) in (0x07, 0x0A, 0x0D, 0x20):
        return key16
    return 0


def fsd_forward(data, data_type="UNDEFINED"):
    """K/transform/FSDCodec.java forward :63-246 with a context.  Returns (applied, bytes, dataType left in the context)."""
    src = bytes(data)
    count = len(src)
    if count == 0:
        return True, b"", data_type
    if count < 1024:
        return False, b"", data_type
    if data_type not in ("UNDEFINED", "MULTIMEDIA", "BIN"):
        return False, b"", data_type
    if magic_type(src) not in (0x424D, 0x52494646, 0x5034, 0x5035, 0x5036, 0):
        return False, b"", data_type
    dist_of = (0, 1, 2, 3, 4, 8, 16)
    count10 = count // 10
    count5 = 2 * count10
    histo = [[0] * 256 for _ in range(7)]
    for st in (0, 2 * count5, 4 * count5):
        for i in range(count10, count5):
            b = src[st + i]
            histo[0][b] += 1
            for k in range(1, 7):
                histo[k][b ^ src[st + i - dist_of[k]]] += 1
    ent = [first_order_entropy_1024(3 * count10, h) for h in histo]
    min_idx = 0
    for i in range(7):
        if ent[i] < ent[min_idx]:
            min_idx = i
    if ent[min_idx] >= ent[0]:
        return False, b"", detect_simple_type(3 * count10, histo[0])
    left = "MULTIMEDIA"
    dist = dist_of[min_idx]
    large = 0
    for i in range(2 * count5, 3 * count5):
        delta = src[i] - src[i - dist]
        if delta < -127 or delta > 127:
            large += 1
    xor_coding = large > (count5 >> 5)
    max_len = count + max(64, count >> 4)                                         # getMaxEncodedLength :320-323
    out = bytearray([1 if xor_coding else 0, dist]) + src[:dist]
    si = dist
    if not xor_coding:
        while si < count and len(out) < max_len - 1:
            delta = src[si] - src[si - dist]
            if delta < -127 or delta > 127:
                out += bytes([255, src[si] ^ src[si - dist]])
            else:
                out.append(((delta >> 31) ^ (delta << 1)) & 0xFF)                 # zigzag
            si += 1
    else:
        while si < count:
            out.append(src[si] ^ src[si - dist])
            si += 1
    if si != count:
        return False, b"", left
    h = [0] * 256
    for i in range(count10):
        h[out[count5 + i]] += 1
        h[out[3 * count5 + i]] += 1
    if first_order_entropy_1024(count5, h) >= ent[0]:
        return False, b"", left
    return True, bytes(out), left


# ---------------------------------------------------------------------------------------------------------------------
# TextCodec (K/transform/TextCodec.java), written from the Java: computeStats :269-384, detectType :387-466, createDictionary
# :205-236, TextCodec1 (:536-1040: reset :585-616, forward :618-795, expandDictionary :798-812, emitSymbols :815-857,
# emitWordIndex :860-873), TextCodec2 (:1042-1620: reset :1092-1120, forward :1124-1291, emitSymbols :1310-1380, emitWordIndex
# :1383-1407) and the wrapper TextCodec.forward :482-509.  Java bytes are signed: the hashes multiply the SIGNED byte.
_DATA_TYPES = ("UNDEFINED", "TEXT", "MULTIMEDIA", "EXE", "NUMERIC", "BASE64", "DNA", "BIN", "UTF8", "SMALL_ALPHABET")   # Global.java:40-80


def _i32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


def _sb(b):                                                                      # a Java byte
    return b - 256 if b >= 128 else b


_T_HASH1 = 0x7FEB352D
_T_HASH2 = _i32(0x846CA68B)
_T_THRESHOLD1, _T_THRESHOLD2, _T_THRESHOLD3, _T_THRESHOLD4 = 128, 128 * 128, 64, 64 * 128
_T_MAX_DICT_SIZE, _T_MAX_WORD_LENGTH = 1 << 19, 31
_T_ESC1, _T_ESC2, _T_LF, _T_CR = 0x0F, 0x0E, 0x0A, 0x0D
_T_MASK_NOT_TEXT, _T_MASK_CRLF, _T_MASK_XML_HTML, _T_MASK_TEXT_CODEC, _T_MASK_DT, _T_MASK_LENGTH = 0x80, 0x40, 0x20, 0x10, 0x0F, 0x0007FFFF


def _t_is_text(b):                                                               # :239-251 on an unsigned byte value
    v = _sb(b | 0x20) if b < 128 else _sb(b)                                      # (byte) (val | 0x20): negative bytes stay negative
    return 0x61 <= v <= 0x7A


def _t_is_upper(b):
    return 0x41 <= b <= 0x5A


_T_DELIMS = [False] * 256
for _c in range(256):                                                            # initDelimiterChars :57-85
    if 0x20 <= _c <= 0x2F or 0x3A <= _c <= 0x3F or _c in (0x0A, 0x09, 0x0D, 0x5F, 0x7C, 0x7B, 0x7D, 0x5B, 0x5D):
        _T_DELIMS[_c] = True


class _DictEntry:                                                                # :1622-1633
    __slots__ = ("buf", "pos", "hash", "data")

    def __init__(self, buf, pos, hsh, idx, length):
        self.buf, self.pos, self.hash, self.data = buf, pos, hsh, (length << 24) | idx


def text_static_dictionary(words_bytes):
    """createDictionary :205-236 over DICT_EN_1024 (the bytes are lower-cased in place as the words are cut)"""
    words = bytearray(words_bytes)
    d = []
    anchor, h = 0, _T_HASH1
    for i in range(len(words)):
        if len(d) >= 1024:
            break
        if not _t_is_text(words[i]):
            continue
        if _t_is_upper(words[i]):
            if i > anchor:
                d.append(_DictEntry(words, anchor, h, len(d), i - anchor))
                anchor, h = i, _T_HASH1
            words[i] ^= 0x20
        h = _i32(_i32(h * _T_HASH1) ^ _i32(_sb(words[i]) * _T_HASH2))
    if len(d) < 1024:
        d.append(_DictEntry(words, anchor, h, len(d), len(words) - anchor))
    return d


def text_compute_stats(block, strict):
    """computeStats :269-384 -> (mode byte, freqs0)"""
    count = len(block)
    freqs0 = [0] * 256
    if not strict and magic_type(block) != 0:                                    # :272-273
        return _T_MASK_NOT_TEXT, freqs0
    freqs = {}
    prv = 0
    for cur in block:                                                            # :284-306
        freqs0[cur] += 1
        freqs[(prv, cur)] = freqs.get((prv, cur), 0) + 1
        prv = cur
    f = lambda a, b: freqs.get((a, b), 0)
    nb_text = freqs0[_T_CR] + freqs0[_T_LF]
    nb_ascii = 0
    for i in range(128):                                                         # :311-316
        if _t_is_text(i):
            nb_text += freqs0[i]
        nb_ascii += freqs0[i]
    nb_bin = count - nb_ascii
    not_text = nb_bin > (count >> 2)                                             # :320
    if not not_text:
        not_text = nb_text < (count // 4)
        if strict:
            not_text = not_text or (freqs0[0] >= (count // 100)) or ((nb_ascii // 95) < (count // 100))
        else:
            not_text = not_text or (freqs0[32] < (count // 50))
    if not_text:
        return _text_detect_type(freqs0, f, count), freqs0
    res = 0
    if nb_bin <= count - count // 10:                                            # :337-357
        f1, f2 = freqs0[0x3C], freqs0[0x3E]
        f3 = f(0x26, 0x61) + f(0x26, 0x67) + f(0x26, 0x6C) + f(0x26, 0x71)
        min_freq = max((count - nb_bin) >> 9, 2)
        if f1 >= min_freq and f2 >= min_freq and f3 > 0:
            if f1 < f2:
                if f1 >= f2 - f2 // 100:
                    res |= _T_MASK_XML_HTML
            elif f2 < f1:
                if f2 >= f1 - f1 // 100:
                    res |= _T_MASK_XML_HTML
            else:
                res |= _T_MASK_XML_HTML
    if freqs0[_T_CR] != 0 and freqs0[_T_CR] == freqs0[_T_LF]:                    # :360-374
        res |= _T_MASK_CRLF
        for i in range(256):
            if i != _T_LF and f(_T_CR, i) != 0:
                res &= ~_T_MASK_CRLF
                break
            if i != _T_CR and f(i, _T_LF) != 0:
                res &= ~_T_MASK_CRLF
                break
    return res, freqs0


def _text_detect_type(freqs0, f, count):                                         # :387-466
    dt = detect_simple_type(count, freqs0)
    if dt != "UNDEFINED":
        return _T_MASK_NOT_TEXT | _DATA_TYPES.index(dt)
    s = freqs0[0xC0] + freqs0[0xC1] + sum(freqs0[i] for i in range(0xF5, 0x100))
    if s != 0:
        return _T_MASK_NOT_TEXT
    sum1 = sum2 = 0
    for i in range(256):
        if i < 0xA0 or i > 0xBF:
            sum1 += f(0xE0, i)
        if i < 0x80 or i > 0x9F:
            sum1 += f(0xED, i)
        if i < 0x90 or i > 0xBF:
            sum1 += f(0xF0, i)
        if i < 0x80 or i > 0x8F:
            sum1 += f(0xF4, i)
        if i < 0x80 or i > 0xBF:
            for j in range(0xC2, 0xE0):
                sum1 += f(j, i)
            for j in range(0xE1, 0xED):
                sum1 += f(j, i)
            sum1 += f(0xF1, i) + f(0xF2, i) + f(0xF3, i) + f(0xEE, i) + f(0xEF, i)
        else:
            sum2 += freqs0[i]
        if sum1 != 0:
            return _T_MASK_NOT_TEXT
    return (_T_MASK_NOT_TEXT | _DATA_TYPES.index("UTF8")) if sum2 >= count // 8 else _T_MASK_NOT_TEXT


def _ilog2(x):                                                                   # Global.log2 :207-212
    return x.bit_length() - 1


class _TextCodecModel:
    """state shared by both variants: dictMap (hash & mask -> entry), dictList, dictSize"""

    def __init__(self, variant, block_size, static_dict):
        self.variant = variant
        if variant == 1:                                                         # TextCodec1(ctx) :558-582
            log = max(min(_ilog2(block_size // 8), 26), 13) if block_size >= 8 else 13
            self.static_size = len(static_dict) + 2
        else:                                                                    # TextCodec2(ctx) :1067-1090
            log = max(min(_ilog2(block_size // 32), 24), 13) if block_size >= 32 else 13
            self.static_size = len(static_dict)
        self.hash_mask = (1 << log) - 1
        self.static_dict = static_dict
        self.is_crlf = False

    def reset(self, count):                                                      # :585-616 / :1092-1120 (a fresh instance per block)
        log = 13 if count < 1024 else max(min(_ilog2(count // 128), 18), 13)
        self.dict_size = 1 << log
        self.dict_map = {}
        self.dict_list = [None] * self.dict_size
        n = min(len(self.static_dict), self.dict_size)
        self.dict_list[:n] = self.static_dict[:n]
        if self.variant == 1:
            k = len(self.static_dict)
            self.dict_list[k] = _DictEntry(bytes([_T_ESC2]), 0, 0, k, 1)
            self.dict_list[k + 1] = _DictEntry(bytes([_T_ESC1]), 0, 0, k + 1, 1)
        for i in range(self.static_size):
            e = self.dict_list[i]
            self.dict_map[e.hash & self.hash_mask] = e
        for i in range(self.static_size, self.dict_size):
            self.dict_list[i] = _DictEntry(None, -1, 0, i, 0)

    def expand(self):                                                            # expandDictionary :798-812 / :1294-1308
        if self.dict_size >= _T_MAX_DICT_SIZE:
            return False
        self.dict_list.extend(_DictEntry(None, -1, 0, i, 0) for i in range(self.dict_size, self.dict_size * 2))
        self.dict_size <<= 1
        return True

    # ---- TextCodec1 ----
    def emit_symbols1(self, src, src_idx, dst, dst_idx, src_end, dst_end):       # :815-857
        for i in range(src_idx, src_end):
            if dst_idx >= dst_end:
                return dst_end + 1
            cur = src[i]
            if cur == _T_ESC1 or cur == _T_ESC2:
                dst[dst_idx] = _T_ESC1
                dst_idx += 1
                idx = self.static_size - 1 if cur == _T_ESC1 else self.static_size - 2
                len_idx = 2
                if idx >= _T_THRESHOLD2:
                    len_idx = 3
                elif idx < _T_THRESHOLD1:
                    len_idx = 1
                if dst_idx + len_idx >= dst_end:
                    return dst_end + 1
                dst_idx = self.emit_word_index1(dst, dst_idx, idx)
            elif cur == _T_CR:
                if not self.is_crlf:
                    dst[dst_idx] = cur
                    dst_idx += 1
            else:
                dst[dst_idx] = cur
                dst_idx += 1
        return dst_idx

    @staticmethod
    def emit_word_index1(dst, dst_idx, val):                                     # :860-873
        if val >= _T_THRESHOLD1:
            if val >= _T_THRESHOLD2:
                dst[dst_idx] = (0xE0 | (val >> 14)) & 0xFF
                dst_idx += 1
            dst[dst_idx] = (0x80 | (val >> 7)) & 0xFF
            dst[dst_idx + 1] = 0x7F & val
            return dst_idx + 2
        dst[dst_idx] = val
        return dst_idx + 1

    # ---- TextCodec2 ----
    def emit_symbols2(self, src, src_idx, dst, dst_idx, src_end, dst_end):       # :1310-1380
        if dst_idx + 2 * (src_end - src_idx) < dst_end:
            for i in range(src_idx, src_end):
                cur = src[i]
                if cur == _T_ESC1:
                    dst[dst_idx] = _T_ESC1
                    dst[dst_idx + 1] = _T_ESC1
                    dst_idx += 2
                elif cur == _T_CR:
                    if not self.is_crlf:
                        dst[dst_idx] = cur
                        dst_idx += 1
                else:
                    dst[dst_idx] = _T_ESC1
                    dst_idx += 1 if cur >= 128 else 0                            # cur >>> 31 on the sign-extended byte
                    dst[dst_idx] = cur
                    dst_idx += 1
        else:
            for i in range(src_idx, src_end):
                cur = src[i]
                if cur == _T_ESC1:
                    if dst_idx >= dst_end - 1:
                        return dst_end + 1
                    dst[dst_idx] = _T_ESC1
                    dst[dst_idx + 1] = _T_ESC1
                    dst_idx += 2
                elif cur == _T_CR:
                    if not self.is_crlf:
                        if dst_idx >= dst_end:
                            return dst_end + 1
                        dst[dst_idx] = cur
                        dst_idx += 1
                else:
                    if cur & 0x80:
                        if dst_idx >= dst_end:
                            return dst_end + 1
                        dst[dst_idx] = _T_ESC1
                        dst_idx += 1
                    if dst_idx >= dst_end:
                        return dst_end + 1
                    dst[dst_idx] = cur
                    dst_idx += 1
        return dst_idx

    @staticmethod
    def emit_word_index2(dst, dst_idx, w_idx):                                   # :1383-1407
        w_idx += 1
        if w_idx >= _T_THRESHOLD3:
            if w_idx >= _T_THRESHOLD4:
                dst[dst_idx] = (0xF0 | (w_idx >> 16)) & 0xFF
                dst[dst_idx + 1] = (w_idx >> 8) & 0xFF
                dst[dst_idx + 2] = w_idx & 0xFF
                return dst_idx + 3
            dst[dst_idx] = (0xC0 | (w_idx >> 8)) & 0xFF
            dst[dst_idx + 1] = w_idx & 0xFF
            return dst_idx + 2
        dst[dst_idx] = 0x80 | w_idx
        return dst_idx + 1


def text_forward(data, variant, block_size, static_dict, data_type="UNDEFINED"):
    """TextCodec.forward :482-509 over TextCodec1.forward :618-795 (variant 1: the FPAQ / TPAQ / CM streams,
    TransformFactory.java:275-286) or TextCodec2.forward :1124-1291 (variant 2), with a context.
    Returns (applied, bytes, dataType left in the context)."""
    src = bytes(data)
    count = len(src)
    if count == 0:
        return True, b"", data_type
    if count < 1024 or count > (1 << 30):                                        # MIN_BLOCK_SIZE / MAX_BLOCK_SIZE :491-492
        return False, b"", data_type
    if data_type not in ("UNDEFINED", "TEXT", "BIN"):                            # :641-649 / :1141-1149
        return False, b"", data_type
    m = _TextCodecModel(variant, block_size, static_dict)
    mode, _freqs0 = text_compute_stats(src, strict=(variant == 1))               # :652 (true) / :1155 (false)
    if mode & _T_MASK_NOT_TEXT:
        return False, b"", _DATA_TYPES[mode & _T_MASK_DT]                          # :655-668: the detected type goes to the context
    data_type = "TEXT"
    m.reset(count)
    dst_end = count                                                              # getMaxEncodedLength :1033 / :1614
    dst_end_guard = dst_end - (4 if variant == 1 else 3)                         # dstEnd4 / dstEnd3
    dst = bytearray(count + 8)
    src_idx = dst_idx = 0
    src_end = count
    emit_anchor = 0
    words = m.static_size
    m.is_crlf = (mode & _T_MASK_CRLF) != 0
    dst[dst_idx] = mode & 0xFF
    dst_idx += 1
    res = True
    while src_idx < src_end and src[src_idx] == 0x20:                            # :686-690
        dst[dst_idx] = 0x20
        dst_idx += 1
        src_idx += 1
        emit_anchor += 1
    delim_anchor = src_idx - 1 if _t_is_text(src[src_idx]) else src_idx          # :692
    emit_symbols = m.emit_symbols1 if variant == 1 else m.emit_symbols2
    while src_idx < src_end:
        cur = src[src_idx]
        if _t_is_text(cur):
            src_idx += 1
            continue
        if src_idx > delim_anchor + 2 and _T_DELIMS[cur]:                        # at least 2 letters
            length = src_idx - delim_anchor - 1
            if length <= _T_MAX_WORD_LENGTH:
                val = _sb(src[delim_anchor + 1])
                h1 = _i32(_i32(_T_HASH1 * _T_HASH1) ^ _i32(val * _T_HASH2))      # :710-711
                h2 = _i32(_i32(_T_HASH1 * _T_HASH1) ^ _i32((val ^ 0x20) * _T_HASH2))
                for i in range(delim_anchor + 2, src_idx):
                    h = _i32(_sb(src[i]) * _T_HASH2)
                    h1 = _i32(_i32(h1 * _T_HASH1) ^ h)
                    h2 = _i32(_i32(h2 * _T_HASH1) ^ h)
                e = None
                e1 = m.dict_map.get(h1 & m.hash_mask)                            # :721-731
                if e1 is not None and e1.hash == h1 and (e1.data >> 24) == length:
                    e = e1
                else:
                    e2 = m.dict_map.get(h2 & m.hash_mask)
                    if e2 is not None and e2.hash == h2 and (e2.data >> 24) == length:
                        e = e2
                if e is not None:                                                # sameWords :469-479 on all but the first letter
                    if src[delim_anchor + 2:delim_anchor + 1 + length] != bytes(e.buf[e.pos + 1:e.pos + length]):
                        e = None
                if e is None:
                    if (length > 3 or (length == 3 and words < _T_THRESHOLD2)) and e1 is None:   # :742-762
                        e = m.dict_list[words]
                        if (e.data & _T_MASK_LENGTH) >= m.static_size:
                            if m.dict_map.get(e.hash & m.hash_mask) is not None:
                                m.dict_map[e.hash & m.hash_mask] = None          # the slot is cleared whoever sits in it
                            e.buf, e.pos, e.hash, e.data = src, delim_anchor + 1, h1, (length << 24) | words
                        m.dict_map[h1 & m.hash_mask] = e
                        words += 1
                        if words >= m.dict_size:
                            if not m.expand():
                                words = m.static_size
                else:
                    if emit_anchor != delim_anchor or src[delim_anchor] != 0x20:  # :766-769
                        dst_idx = emit_symbols(src, emit_anchor, dst, dst_idx, delim_anchor + 1, dst_end)
                    if dst_idx >= dst_end_guard:
                        res = False
                        break
                    if variant == 1:                                             # :776-778
                        dst[dst_idx] = _T_ESC1 if e is e1 else _T_ESC2
                        dst_idx += 1
                        dst_idx = m.emit_word_index1(dst, dst_idx, e.data & _T_MASK_LENGTH)
                    else:                                                        # :1262-1265: case flip is encoded as 0x80
                        dst[dst_idx] = 0x80
                        dst_idx += 0 if e is e1 else 1
                        dst_idx = m.emit_word_index2(dst, dst_idx, e.data & _T_MASK_LENGTH)
                    emit_anchor = delim_anchor + 1 + (e.data >> 24)
        delim_anchor = src_idx
        src_idx += 1
    if res:
        d_idx = emit_symbols(src, emit_anchor, dst, dst_idx, src_end, dst_end)
        if d_idx > dst_end:
            res = False
        else:
            dst_idx = d_idx
        res = res and src_idx == src_end
    if not res:
        return False, b"", data_type
    if variant == 1:                                                             # TextCodec.forward :501-506 (bsVersion 7)
        dst[0] &= ~_T_MASK_TEXT_CODEC & 0xFF
    else:
        dst[0] |= _T_MASK_TEXT_CODEC
    return True, bytes(dst[:dst_idx]), data_type


# ---------------------------------------------------------------------------------------------------------------------
# Bit I/O and the container, written from the Java: DefaultOutputBitStream (K/bitstream/DefaultOutputBitStream.java:103-222,
# 229-296), the stream header (K/io/CompressedOutputStream.java:236-313), a block (:733-985: copy blocks, skip flags, mode byte,
# header checksum, raw "transformed copy" fallback), block framing and the end marker (:1024-1035, :489-492), Sequence.forward
# (K/transform/Sequence.java:56-127), BWTBlockCodec.forward (K/transform/BWTBlockCodec.java:71-128) over the output convention of
# DivSufSort.computeBWT (:204-227), ZRLT.forward (K/transform/ZRLT.java:54-136).
class JavaOutputBitStream:
    """DefaultOutputBitStream with its 64-bit accumulator `current`, `availBits`, the byte buffer and `position`"""

    def __init__(self, buffer_size=16384):
        self.sink = bytearray()
        self.buffer = bytearray(buffer_size)
        self.position = 0
        self.avail = 64
        self.current = 0
        self.written_bits = 0
        self.closed = False

    def write_bits(self, value, count):                                          # writeBits(long, int) :103-123
        if count == 0:
            return 0
        assert count <= 64
        value &= M64                                                             # a Java long, two's complement
        self.current |= ((value << (64 - count)) & M64) >> (64 - self.avail)
        if count >= self.avail:
            remaining = count - self.avail
            self._push_current()
            if remaining != 0:
                self.current = (value << (64 - remaining)) & M64
                self.avail -= remaining
        else:
            self.avail -= count
        return count

    def write_bytes(self, bits, start, count):                                   # writeBits(byte[], int, int) :136-205
        remaining = count
        if (self.avail & 7) == 0:
            while self.avail != 64 and remaining >= 8:
                self.write_bits(_sb(bits[start]), 8)                             # the signed byte, sign-extended into the long
                start += 1
                remaining -= 8
            max_pos = len(self.buffer) - 8
            while (remaining >> 3) >= max_pos - self.position:
                k = max_pos - self.position
                self.buffer[self.position:self.position + k] = bits[start:start + k]
                start += k
                remaining -= k << 3
                self.position = max_pos
                self._flush()
            r = (remaining >> 6) << 3
            if r > 0:
                self.buffer[self.position:self.position + r] = bits[start:start + r]
                self.position += r
                start += r
                remaining -= r << 3
        elif remaining >= 64:
            r = 64 - self.avail
            while remaining >= 64:
                value = int.from_bytes(bits[start:start + 8], "big")
                self.current |= value >> r
                self._push_current()
                self.current = (value << (64 - r)) & M64                          # value << -r: Java masks the shift count to 6 bits
                start += 8
                remaining -= 64
            self.avail -= r
        while remaining >= 8:
            self.write_bits(bits[start] & 0xFF, 8)
            start += 1
            remaining -= 8
        if remaining > 0:
            self.write_bits((bits[start] & 0xFF) >> (8 - remaining), remaining)  # bits[start] >>> (8 - remaining) then masked to `remaining` bits
        return count

    def _push_current(self):                                                     # :211-220
        self.buffer[self.position:self.position + 8] = self.current.to_bytes(8, "big")
        self.avail = 64
        self.current = 0
        self.position += 8
        if self.position >= len(self.buffer) - 8:
            self._flush()

    def _flush(self):                                                            # :229-243
        if self.position > 0:
            self.sink += self.buffer[:self.position]
            self.written_bits += self.position << 3
            self.position = 0

    def written(self):                                                           # :318-320
        return self.written_bits + (self.position << 3) + (64 - self.avail)

    def close(self):                                                             # :253-296: the last byte may be incomplete (zero padded)
        if self.closed:
            return
        shift = 56
        while self.avail < 64:
            self.buffer[self.position] = (self.current >> shift) & 0xFF
            self.position += 1
            self.avail += 8
            shift -= 8
        self.written_bits -= self.avail - 64
        self.avail = 64
        self._flush()
        self.closed = True
        self.position = 0
        self.avail = 0
        self.written_bits -= 64


def _mix32(checksum, hsh, value):                                                # CompressedOutputStream.mix32 :89-93
    checksum ^= (hsh * (~value & 0xFFFFFFFF)) & 0xFFFFFFFF
    checksum = ((checksum << 13) | (checksum >> 19)) & 0xFFFFFFFF
    return (checksum * 5 + 0x52DCE729) & 0xFFFFFFFF


_TRANSFORM_IDS = {"NONE": 0, "BWT": 1, "BWTS": 2, "LZ": 3, "SNAPPY": 4, "RLT": 5, "ZRLT": 6, "MTFT": 7, "RANK": 8, "EXE": 9, "TEXT": 10,
                  "ROLZ": 11, "ROLZX": 12, "SRT": 13, "LZP": 14, "MM": 15, "LZX": 16, "UTF": 17, "PACK": 18, "DNA": 19}   # TransformFactory.java:36-58
_ENTROPY_IDS = {"NONE": 0, "HUFFMAN": 1, "FPAQ": 2, "PAQ": 3, "RANGE": 4, "ANS0": 5, "CM": 6, "TPAQ": 7, "ANS1": 8, "TPAQX": 9}   # EntropyCodecFactory.java


def transform_type_word(names):
    """TransformFactory.getType :132-158: eight 6-bit tokens, the first one in the top bits of a 48-bit word"""
    w = 0
    for i, n in enumerate(names):
        w |= _TRANSFORM_IDS[n] << (42 - 6 * i)
    return w


def suffix_array_by_doubling(data):
    """suffix array of `data` (shorter suffix first) by rank doubling with numpy sorts: a third suffix sorter, independent of the
    oracle's induced sorting and of the HIP path's trie / radix rounds"""
    import numpy as np
    a = np.frombuffer(bytes(data), dtype=np.uint8).astype(np.int64)
    n = len(a)
    rank = a + 1
    k = 1
    while True:
        r2 = np.zeros(n, dtype=np.int64)
        r2[:n - k] = rank[k:] if k < n else 0
        key = rank * (n + 2) + r2
        order = np.argsort(key, kind="stable")
        ks = key[order]
        newr = np.empty(n, dtype=np.int64)
        newr[order] = np.concatenate(([0], np.cumsum(ks[1:] != ks[:-1]))) + 1
        rank = newr
        if rank.max() == n or k >= n:
            return order
        k *= 2


def bwt_block_forward(data):
    """BWTBlockCodec.forward :71-128: mode byte, the primary indexes (minus one) big endian, then the BWT of DivSufSort.computeBWT
    :204-227 (out[0] = in[n-1]; the suffix array row of suffix 0 is left out, rows before it shift by one) with
    primary[k] = ISA[k * step] + 1 (constructBWT :233-325), chunks from BWT.getBWTChunks :561-563"""
    src = bytes(data)
    n = len(src)
    if n == 0:
        return True, b""
    log_bs = _ilog2(n)
    if n & (n - 1):
        log_bs += 1
    p_size = (log_bs + 7) >> 3
    if p_size <= 0 or p_size >= 5:
        return False, b""
    chunks = 1 if n < 256 else 8
    if n == 1:                                                                   # BWT.forward :174-177 copies; the primary index stays 0
        body = src
        primaries = [0] * chunks
    else:
        sa = suffix_array_by_doubling(src)
        isa = [0] * n
        for r, s in enumerate(sa):
            isa[int(s)] = r
        p = isa[0]
        out = bytearray(n)
        out[0] = src[n - 1]
        for i in range(n):
            if i < p:
                out[1 + i] = src[int(sa[i]) - 1]
            elif i > p:
                out[i] = src[int(sa[i]) - 1]
        body = bytes(out)
        st = n // chunks
        step = st + 1 if st * chunks != n else st
        primaries = [isa[k * step] + 1 if k * step < n else 0 for k in range(chunks)]
    hdr = bytearray([(_ilog2(chunks) << 2) | (p_size - 1)])
    for k in range(chunks):
        hdr += ((primaries[k] - 1) & ((1 << (8 * p_size)) - 1)).to_bytes(p_size, "big")
    return True, bytes(hdr) + body


def zrlt_forward(data):
    """ZRLT.forward :54-136: the output may not reach the input's length"""
    src = bytes(data)
    count = len(src)
    if count == 0:
        return True, b""
    dst = bytearray()
    dst_end = count
    i = 0
    res = True
    while i < count:
        if src[i] == 0:
            run = 1
            while i + run < count and src[i + run] == 0:
                run += 1
            i += run
            run += 1
            log2 = _ilog2(run)
            if len(dst) >= dst_end - log2:
                res = False
                break
            while log2 > 0:
                log2 -= 1
                dst.append((run >> log2) & 1)
            continue
        val = src[i]
        if val >= 0xFE:
            if len(dst) >= dst_end - 1:
                res = False
                break
            dst += bytes([0xFF, val - 0xFE])
        else:
            if len(dst) >= dst_end:
                res = False
                break
            dst.append(val + 1)
        i += 1
    return res and i == count, bytes(dst)


def _magic_class(magic):                                                         # Magic.isCompressed / isMultimedia / isExecutable :185-258
    compressed = magic in (0xFFD8FFE0, 0x47494638, 0x89504E47, 0x377ABCAF, 0x28B52FFD, 0x81CFB2CE, 0x4D534346, 0x504B0304, 0x1F8B,
                           0x425A68, 0x664C6143, 0x494433, 0xFD377A58, 0x4B414E5A, 0x52617221)
    multimedia = magic in (0xFFD8FFE0, 0x47494638, 0x89504E47, 0x52494646, 0x664C6143, 0x494433, 0x424D, 0x5034, 0x5035, 0x5036)
    executable = magic in (0x7F454C46, 0x4D5A, 0xFEEDFACE, 0xCEFAEDFE, 0xFEEDFACF, 0xCFFAEDFE)
    return compressed, multimedia, executable


def knz_stream(data, names, entropy, block_size, static_words, input_size=0):
    """the whole .knz a CompressedOutputStream (no checksum, one job; input_size = the "fileSize" context entry, 0 = unknown) writes for `data`:
    -t names -e entropy -b block_size.  Stages: TEXT, UTF, BWT, RANK, MTFT, SRT, ZRLT; coders: NONE, ANS0, FPAQ."""
    data = bytes(data)
    ttype = transform_type_word(names)
    etype = _ENTROPY_IDS[entropy]
    obs = JavaOutputBitStream(65536)
    # ---- writeHeader :236-313 ----
    obs.write_bits(0x4B414E5A, 32)
    obs.write_bits(7, 4)
    chk_size = 0
    obs.write_bits(chk_size, 2)
    obs.write_bits(etype, 5)
    obs.write_bits(ttype, 48)
    obs.write_bits(block_size >> 4, 28)
    sz_mask = 0                                                                  # :266-280
    if input_size != 0 and input_size < (1 << 48):
        if input_size >= (1 << 32):
            sz_mask = 3
        else:
            isz = input_size
            if isz > (1 << 30):
                isz >>= 4
                sz_mask += 1
            sz_mask += (_ilog2(isz) >> 4) + 1
    obs.write_bits(sz_mask, 2)
    if sz_mask > 0:
        obs.write_bits(input_size, 16 * sz_mask)
    obs.write_bits(0, 15)
    HASH = 0x1E35A7BD
    ck = (HASH * ((0x01030507 * 7) & 0xFFFFFFFF)) & 0xFFFFFFFF
    ck = _mix32(ck, HASH, chk_size)
    ck = _mix32(ck, HASH, etype)
    ck = _mix32(ck, HASH, (ttype >> 32) & 0xFFFFFFFF)
    ck = _mix32(ck, HASH, ttype & 0xFFFFFFFF)
    ck = _mix32(ck, HASH, block_size)
    if sz_mask > 0:
        ck = _mix32(ck, HASH, (input_size >> 32) & 0xFFFFFFFF)
        ck = _mix32(ck, HASH, input_size & 0xFFFFFFFF)
    ck = (ck >> 23) ^ (ck >> 3)
    obs.write_bits(ck, 24)
    # ---- blocks ----
    for off in range(0, len(data), block_size):
        block = data[off:off + block_size]
        blob, written = _knz_block(block, names, entropy, block_size, static_words)
        lw = 3 if written < 8 else _ilog2(written >> 3) + 4                       # :1024-1035
        obs.write_bits(lw - 3, 5)
        obs.write_bits(written, lw)
        obs.write_bytes(blob, 0, written)
    obs.write_bits(0, 5)                                                         # close :489-492
    obs.write_bits(0, 3)
    obs.close()
    return bytes(obs.sink)


def _knz_block(block, names, entropy, block_size, static_words):
    """EncodingTask.encodeBlock :733-985 for one non-empty block -> (bytes, bits written)"""
    n = len(block)
    mode = 0
    if n <= 15:                                                                  # SMALL_BLOCK_SIZE :764-767
        names, entropy = ["NONE"], "NONE"
        mode |= 0x80
    # the block's "dataType" context entry :795-804
    data_type = "UNDEFINED"
    if n >= 4:
        c, m, x = _magic_class(magic_type(block))
        data_type = "BIN" if c else ("MULTIMEDIA" if m else ("EXE" if x else "UNDEFINED"))
    # ---- Sequence.forward :56-127 ----
    skip_flags = 0xFF
    cur = block
    tc_variant = 1 if entropy in ("FPAQ", "TPAQ", "TPAQX", "CM") else 2          # TransformFactory.java:275-286
    for i, name in enumerate(names):
        if name == "NONE":
            ok, out = True, cur                                                  # NullTransform copies
        elif name == "TEXT":
            ok, out, data_type = text_forward(cur, tc_variant, block_size, text_static_dictionary(static_words), data_type)
        elif name == "UTF":
            ok, out, data_type = utf_forward(cur, data_type)
        elif name == "BWT":
            ok, out = bwt_block_forward(cur)
        elif name == "RANK":
            ok, out = True, sbrt_forward(cur, 2)
        elif name == "MTFT":
            ok, out = True, sbrt_forward(cur, 1)
        elif name == "SRT":
            ok, out = True, srt_forward(cur)
        elif name == "ZRLT":
            ok, out = zrlt_forward(cur)
        else:
            raise ValueError(name)
        if not ok:
            continue
        skip_flags &= ~(1 << (7 - i)) & 0xFF
        cur = out
    post = len(cur)
    data_size = 1 if post < 256 else (_ilog2(post) >> 3) + 1                     # :825-826
    nb_functions = len(names)
    mode |= ((data_size - 1) & 3) << 5

    def header(os, mode_byte, with_flags):
        os.write_bits(mode_byte, 8)
        if with_flags:
            os.write_bits(skip_flags, 8)
        os.write_bits(post, 8 * data_size)
        os.write_bits(0, 8)                                                      # the header checksum byte, patched below

    os = JavaOutputBitStream(16384)
    header_skip_flags = skip_flags
    if (mode & 0x80) or nb_functions <= 4:                                       # :866-877
        mode |= skip_flags >> 4
        header_skip_flags = 0 if (mode & 0x80) else ((mode << 4) | 0x0F) & 0xFF
        header(os, mode, False)
        ck_index = 1 + data_size
    else:
        mode |= 0x10
        header(os, mode, True)
        ck_index = 2 + data_size
    # ---- entropy coder on the block's private stream :905-921 ----
    if entropy == "NONE":
        os.write_bytes(cur, 0, 8 * post)                                         # NullEntropyEncoder.encode :66-81
    elif entropy == "ANS0":
        bits, nbits = ans0_encode(cur)
        os.write_bytes(bits, 0, nbits)
    elif entropy == "FPAQ":
        bits = fpaq_encode(cur)
        os.write_bytes(bits, 0, 8 * len(bits))
    else:
        raise ValueError(entropy)
    os.close()
    written = os.written()                                                       # :923, read after close(): availBits = 0 and written -= 64 cancel
    blob = bytearray(os.sink)
    if not (mode & 0x80) and post < ((written + 7) >> 3):                        # raw "transformed copy" fallback :926-973
        copy_mode = mode | 0x80 | 0x10
        os = JavaOutputBitStream(16384)
        header(os, copy_mode, nb_functions > 4)
        if nb_functions > 4:
            ck_index = 2 + data_size
            header_skip_flags = skip_flags
        else:
            ck_index = 1 + data_size
            header_skip_flags = ((copy_mode << 4) | 0x0F) & 0xFF
        os.write_bytes(cur, 0, post << 3)
        os.close()
        written = os.written()
        blob = bytearray(os.sink)
        mode = copy_mode
    HASH = 0x1E35A7BD                                                            # :975-985
    ck = (HASH * 0x01030507) & 0xFFFFFFFF
    ck = _mix32(ck, HASH, mode & 0xFF)
    ck = _mix32(ck, HASH, header_skip_flags & 0xFF)
    ck = _mix32(ck, HASH, post)
    ck = _mix32(ck, HASH, (written >> 32) & 0xFFFFFFFF)
    ck = _mix32(ck, HASH, written & 0xFFFFFFFF)
    ck = (ck >> 23) ^ (ck >> 3)
    blob[ck_index] = ck & 0xFF
    return bytes(blob), written


# ---- decoders, written from the Java (not from oracle/*.c): what a malformed input does is part of the model.  A Java exception
# (array index past the physical array, null table entry) is JavaException: the block fails like a "false" ----
def zrlt_inverse(data, cap):
    """K/transform/ZRLT.java inverse :146-231 with output.length = cap.  -> (ok, bytes written).  The run is refused when it would
    reach dstEnd (`>=`, :186) but the trailing run may end exactly there (`>`, :221)."""
    src = bytes(data)
    count = len(src)
    if count == 0:
        return True, b""
    dst = bytearray()
    dst_end = cap
    i = 0
    run = 0
    while True:
        val = src[i]
        if val <= 1:
            run = 1
            out_of_input = False
            while True:
                run += run + val
                i += 1
                if i >= count:
                    out_of_input = True
                    break
                val = src[i]
                if val > 1:
                    break
            if out_of_input:
                break                                                             # break mainLoop :177 (run still holds its +1)
            run -= 1
            if run > 0:
                if len(dst) + run >= dst_end:
                    break
                dst += bytes(run)
                run = 0
        if val == 0xFF:
            i += 1
            if i >= count:
                break
            dst.append((0xFE + src[i]) & 0xFF)
        else:
            dst.append(val - 1)
        i += 1
        if i >= count or len(dst) >= dst_end:
            break
    if run > 0:
        run -= 1
        if len(dst) + run > dst_end:
            return False, bytes(dst)
        dst += bytes(run)
    return i == count, bytes(dst)


def sbrt_inverse(data, mode):
    """K/transform/SBRT.java inverse :154-214 (no failing input exists: every rank names a symbol)."""
    src = bytes(data)
    m1 = 0 if mode == 3 else -1
    m2 = 0 if mode == 1 else -1
    s = 1 if mode == 2 else 0
    p, q, r2s = [0] * 256, [0] * 256, list(range(256))
    out = bytearray(len(src))
    for i, r in enumerate(src):
        c = r2s[r]
        out[i] = c
        qc = ((i & m1) + (p[c] & m2)) >> s
        p[c] = i
        q[c] = qc
        while r > 0 and q[r2s[r - 1]] <= qc:
            r2s[r] = r2s[r - 1]
            r -= 1
        r2s[r] = c
    return bytes(out)


def srt_inverse(data, cap):
    """K/transform/SRT.java inverse :178-263, decodeHeader :327-346 (at most five bytes per count: a fifth byte's bits land at 28 and
    may make the int negative), preprocess :266-302.  Reads past the input are a Java array fault here (the model's array is the input)."""
    src = bytes(data)
    n_in = len(src)
    if n_in == 0:
        return True, b""

    def rd(k):
        if k < 0 or k >= n_in:
            raise JavaException("ArrayIndexOutOfBounds")
        return src[k]

    freqs = [0] * 256
    k = 0
    for i in range(256):
        val = rd(k); k += 1
        res = val & 0x7F
        shift = 7
        while val >= 128:
            val = rd(k); k += 1
            res = _i32(res | ((val & 0x7F) << shift))
            if shift > 21:
                break
            shift += 7
        freqs[i] = res
    header = k
    count = n_in - header
    if count > cap:
        return False, b""
    symbols = [i for i in range(256) if freqs[i] > 0]
    nb = len(symbols)
    h = 4
    while h < nb:
        h = h * 3 + 1
    while True:
        h //= 3
        for i in range(h, nb):
            t = symbols[i]
            b = i - h
            while b >= 0 and (freqs[symbols[b]] < freqs[t] or (freqs[t] == freqs[symbols[b]] and t < symbols[b])):
                symbols[b + h] = symbols[b]
                b -= h
            symbols[b + h] = t
        if h == 1:
            break
    buckets, ends, r2s = [0] * 256, [0] * 256, [0] * 256
    pos = 0
    for i in range(nb):
        c = symbols[i]
        if header + pos < 0 or header + pos >= n_in:
            return False, b""
        r2s[rd(header + pos)] = c
        buckets[c] = pos + 1
        pos = _i32(pos + freqs[c])
        ends[c] = pos
    c = r2s[0]
    out = bytearray(max(count, 0))
    for i in range(count):
        out[i] = c
        if buckets[c] < ends[c]:
            r = rd(header + buckets[c])
            buckets[c] += 1
            if r == 0:
                continue
            r2s[0:r] = r2s[1:r + 1]
            r2s[r] = c
            c = r2s[0]
        else:
            if nb == 1:
                continue
            nb -= 1
            if nb < 0:
                continue                                                          # (nb 0: nothing was ever listed; the loop shifts nothing)
            r2s[0:nb] = r2s[1:nb + 1]
            c = r2s[0]
    return True, bytes(out)


def utf_inverse(data, cap):
    """K/transform/UTFCodec.java inverse :224-306 (bit stream version >= 4: unpackV1 :508-541) with output.length = cap; the physical
    output array is taken as cap + 64 bytes for writeInt32's four-byte store; a null table entry (alias >= n) is a JavaException."""
    src = bytes(data)
    count = len(src)
    if count == 0:
        return True, b""
    if count < 4:
        return False, b""
    start = src[0] & 3
    adjust = src[1] & 3
    n = (src[2] << 8) + src[3]
    src_end = count - 4 + adjust
    dst_end = cap - 4
    if n == 0 or n >= 32768 or 3 * n >= count:
        return False, b""
    table = []
    k = 4
    for _ in range(n):
        s = (src[k] << 16) | (src[k + 1] << 8) | src[k + 2]
        tag = s >> 19
        if tag == 0:
            le, ln = s, 1
        elif tag == 1:
            le, ln = ((s & 0xFF) << 8) | ((s >> 8) & 0xFF), 2
        elif tag == 2:
            le, ln = (((s >> 12) & 0x0F) | 0xE0) | ((((s >> 6) & 0x3F) | 0x80) << 8) | (((s & 0x3F) | 0x80) << 16), 3
        elif 4 <= tag <= 7:
            le = (((s >> 18) & 7) | 0xF0) | ((((s >> 12) & 0x3F) | 0x80) << 8) | ((((s >> 6) & 0x3F) | 0x80) << 16) | (((s & 0x3F) | 0x80) << 24)
            ln = 4
        else:
            return False, b""
        table.append((le, ln))
        k += 3
    if dst_end < 0:
        return False, b""
    dst = bytearray(cap + 64)
    d = 0

    def rd(j):
        if j >= count:
            raise JavaException("ArrayIndexOutOfBounds")
        return src[j]

    for _ in range(start):
        dst[d] = rd(k); d += 1; k += 1
    while k < src_end and d < dst_end:
        alias = rd(k); k += 1
        if alias >= 128:
            alias = (rd(k) << 7) + (alias & 0x7F); k += 1
        if alias >= n:
            raise JavaException("NullPointer")
        le, ln = table[alias]
        dst[d:d + 4] = le.to_bytes(4, "little")
        d += ln
    if k < src_end or d >= dst_end - count + src_end:
        return False, bytes(dst[:d])
    for _ in range(src_end, count):
        dst[d] = rd(k); d += 1; k += 1
    return True, bytes(dst[:d])


def bwt_block_inverse(data, cap):
    """K/transform/BWTBlockCodec.java inverse :131-201 (bit stream version > 5: mode byte, primary indexes stored minus one) +
    K/transform/BWT.java inverse :203-235 and inverseMergeTPSI :289-381 (blocks up to 8 MiB).  A damaged primary index that passes
    the range tests yields WRONG BYTES with a success verdict: that is the reference's behaviour and part of the model.  Under 256
    bytes the first entry's link (0xFF) can point past the table: JavaException."""
    src = bytes(data)
    block_size = len(src)
    if block_size == 0:
        return True, b""
    mode = src[0]
    chunks = 1 << ((mode >> 2) & 7)
    p_size = (mode & 3) + 1
    header = 1 + chunks * p_size
    if block_size < header:
        return False, b""
    count = block_size - header
    if chunks != (1 if count < 256 else 8):
        return False, b""
    primary = [0] * 8
    k = 1
    for i in range(chunks):
        v = int.from_bytes(src[k:k + p_size], "big")
        k += p_size
        if v >= 0x7FFFFFFF:
            return False, b""
        if i >= 8:                                                                # setPrimaryIndex :132-138 (cannot happen: chunks is 1 or 8)
            return False, b""
        primary[i] = v + 1
    if count == 0:
        return True, b""
    if cap <= 0 or header > count or count > cap:                                 # BWT.inverse :207-219 (src.index > src.length is one of its tests)
        return False, b""
    body = src[header:]
    if count == 1:
        return True, body
    if count > 8 * 1024 * 1024:
        raise NotImplementedError("inverseBiPSIv2")
    p_idx = primary[0]
    if p_idx <= 0 or p_idx > count:
        return False, b""
    b = [0] * 256
    for c in body:
        b[c] += 1
    s = 0
    for i in range(256):
        b[i], s = s, s + b[i]
    table = [0] * max(count, 64)
    for i, val in enumerate(body):
        table[b[val]] = (0xFF00 | val) if i == 0 else ((((i - 1) if i < p_idx else i) << 8) | val)
        b[val] += 1

    def at(t):
        if t >= len(table):
            raise JavaException("ArrayIndexOutOfBounds")
        return table[t]

    out = bytearray(count)
    if count < 256:
        t = p_idx - 1
        for i in range(count):
            ptr = at(t)
            out[i] = ptr & 0xFF
            t = ptr >> 8
        return True, bytes(out)
    ck = (count >> 3) if (count & 7) == 0 else (count >> 3) + 1
    ts = [primary[j] - 1 for j in range(8)]
    if any(t < 0 or t >= count for t in ts):
        return False, b""
    end = count - ck * 7
    for n in range(ck):
        for j in range(8 if n < end else 7):
            ptr = table[ts[j]]
            out[n + ck * j] = ptr & 0xFF
            ts[j] = ptr >> 8
    return True, bytes(out)


def fsd_inverse(data, cap):
    """K/transform/FSDCodec.java inverse :249-313 with output.length = cap; the physical output array is taken as cap bytes too (the
    XOR branch does not test dstEnd: a write past it is the Java's array fault)."""
    src = bytes(data)
    count = len(src)
    if count == 0:
        return True, b""
    if count < 2:
        raise JavaException("ArrayIndexOutOfBounds")
    mode, dist = src[0], src[1]
    if dist < 1 or (dist > 4 and dist != 8 and dist != 16):
        return False, b""
    dst = bytearray()

    def put(v):
        if len(dst) >= cap:
            raise JavaException("ArrayIndexOutOfBounds")
        dst.append(v & 0xFF)

    k = 2
    for _ in range(dist):
        if k >= count:
            raise JavaException("ArrayIndexOutOfBounds")
        put(src[k]); k += 1
    if mode == 0:
        while k < count and len(dst) < cap:
            if src[k] == 0xFF:
                k += 1
                if k == count:
                    break
                put(src[k] ^ dst[len(dst) - dist])
                k += 1
                continue
            delta = (src[k] >> 1) ^ -(src[k] & 1)
            put(dst[len(dst) - dist] + delta)
            k += 1
    elif mode == 1:
        while k < count:
            put(src[k] ^ dst[len(dst) - dist])
            k += 1
    else:
        return False, b""
    return k == count, bytes(dst)


def alias_inverse(data, cap):
    """K/transform/AliasCodec.java inverse :281-418 with output.length = cap and a physical output array of cap bytes (the packed
    branches store without testing the output's end: a store past it is the Java's array fault, like a read past the input)."""
    src = bytes(data)
    count = len(src)
    if count == 0:
        return True, b""
    dst = bytearray()

    def rd(k):
        if k < 0 or k >= count:
            raise JavaException("ArrayIndexOutOfBounds")
        return src[k]

    def put_at(pos, v):
        if pos >= cap:
            raise JavaException("ArrayIndexOutOfBounds")
        while len(dst) <= pos:
            dst.append(0)
        dst[pos] = v & 0xFF

    k = 0
    n = rd(k); k += 1
    d = 0
    if n < 16:
        return False, b""
    if n >= 240:
        n = 256 - n
        if n == 1:
            val = rd(k); k += 1
            size = _i32(rd(k) | (rd(k + 1) << 8) | (rd(k + 2) << 16) | (rd(k + 3) << 24))
            if size > cap:
                return False, b""
            if size < 0:                                                          # :306-312: the test passes, nothing is written, output.index moves BACKWARDS and
                return False, b""                                                 # the call returns true; the block then fails in Sequence (negative length): a failure
            for i in range(size):
                put_at(i, val)
            return True, bytes(dst[:max(size, 0)])
        idx2symb = [0] * 16
        for i in range(n):
            idx2symb[i] = rd(k); k += 1
        adjust = rd(k); k += 1
        if adjust >= 4:
            return False, b""
        if n <= 4:
            for _ in range(adjust):
                put_at(d, rd(k)); d += 1; k += 1
            while k < count:
                b = src[k]; k += 1
                for j in range(4):                                                # writeInt32, little endian, of the word built :322-331: the low byte is bits 6..7's symbol
                    put_at(d + j, idx2symb[(b >> (6 - 2 * j)) & 3])
                d += 4
        else:
            if adjust != 0:
                put_at(d, rd(k)); d += 1; k += 1
            while k < count:
                b = src[k]; k += 1
                put_at(d, idx2symb[b >> 4])
                put_at(d + 1, idx2symb[b & 15])
                d += 2
        return True, bytes(dst[:d])
    adjust = rd(k); k += 1
    src_end = count - adjust
    map16 = [0x10000 | i for i in range(256)]
    for _ in range(n):
        map16[rd(k + 2)] = 0x20000 | rd(k) | (rd(k + 1) << 8)
        k += 3
    nb = src_end - k
    if nb <= ((cap - d) >> 1):
        while k < src_end:
            val = map16[rd(k)]; k += 1
            put_at(d, val); put_at(d + 1, val >> 8)
            d += val >> 16
    else:
        while k < src_end and d + 1 < cap:
            val = map16[rd(k)]; k += 1
            put_at(d, val); put_at(d + 1, val >> 8)
            d += val >> 16
        while k < src_end:
            val = map16[rd(k)]; k += 1
            inc = val >> 16
            if d + inc > cap:
                return False, bytes(dst[:d])
            put_at(d + inc - 1, val >> 8)
            put_at(d, val)
            d += inc
    if adjust != 0:
        if d >= cap:
            return False, bytes(dst[:d])
        put_at(d, rd(k)); d += 1; k += 1
    return True, bytes(dst[:d])


def fpaq_decode(data, nbits, count):
    """K/entropy/FPAQDecoder.java decode :164-234 (bit stream version >= 4: decodeBitV2 :294-314), read :322-334,
    EntropyUtils.readVarInt :284-300, on a block's bit string of nbits bits.  -> (return value, bytes, bits consumed); a read past
    the end of the bits is the bit stream's exception, a negative chunk size Arrays.fill's: JavaException."""
    M56, MASK_24_56, PSCALE = (1 << 56) - 1, 0x00FFFFFFFF000000, 65536
    src = bytes(data)
    big = int.from_bytes(src, "big") if src else 0
    total = len(src) * 8
    pos = 0

    def read_bits(n):
        nonlocal pos
        if pos + n > nbits or pos + n > total:
            raise JavaException("BitStreamException: end of stream")
        v = (big >> (total - pos - n)) & ((1 << n) - 1)
        pos += n
        return v

    if count == 0:
        return 0, b"", 0
    low, high, current = 0, M56, 0
    probs = [[PSCALE >> 1] * 256 for _ in range(4)]
    out = bytearray(count)
    start = 0
    while start < count:
        value = read_bits(8)
        sz = value & 0x7F
        shift = 7
        while value >= 128:
            value = read_bits(8)
            sz |= (value & 0x7F) << shift
            if shift == 28:
                break
            shift += 7
        sz = _i32(sz)
        if sz >= 2 * count:
            return 0, bytes(out), pos
        current = read_bits(56)
        if sz < 0:
            raise JavaException("ArrayIndexOutOfBounds (Arrays.fill from a negative index)")
        buf = read_bits(8 * sz).to_bytes(sz, "big") + bytes(max(sz + (sz >> 2), 1024) - sz + 4) if sz else bytes(1028)
        idx = 0
        end = start + min(4 * 1024 * 1024, count - start)
        p = probs[0]
        for i in range(start, end):
            ctx = 1
            for _ in range(8):
                split = (((((high - low) & M64) >> 8) * p[ctx] & M64) >> 8) + low
                if split >= current:
                    high = split
                    p[ctx] -= (p[ctx] - PSCALE + 64) >> 6
                    ctx = (ctx << 1) + 1
                else:
                    low = split + 1
                    p[ctx] -= p[ctx] >> 6
                    ctx <<= 1
                while ((low ^ high) & MASK_24_56) == 0:
                    low = (low << 32) & M56
                    high = ((high << 32) | 0xFFFFFFFF) & M56
                    if idx + 4 > sz:
                        current = (current << 32) & M56
                        idx = sz + 1
                    else:
                        current = ((current << 32) | int.from_bytes(buf[idx:idx + 4], "big")) & M56
                        idx += 4
            out[i] = ctx & 0xFF
            if idx > sz:
                return 0, bytes(out), pos
            p = probs[(ctx & 0xFF) >> 6]
        if idx > sz:
            return 0, bytes(out), pos
        start = end
    return count, bytes(out), pos


def ans0_decode(data, nbits, count):
    """K/entropy/ANSRangeDecoder.java, order 0, bit stream version >= 4: decode :158-206, decodeHeader :374-466,
    EntropyUtils.decodeAlphabet :86-118, decodeChunkV2 :281-366, decodeSymbol :266-279 (32-bit int states: the compare with ANS_TOP
    is SIGNED), readVarInt.  -> (return value, bytes, bits consumed, clean): a chunk whose byte count does not come out (`n == sz`
    false, :365) only ENDS the walk -- decode still returns count (:192-193, :204); clean says that did not happen.  BitStreamException
    and array faults are JavaException."""
    src = bytes(data)
    big = int.from_bytes(src, "big") if src else 0
    total = len(src) * 8
    pos = 0

    def read_bits(n):
        nonlocal pos
        if n == 0:
            return 0
        if pos + n > nbits or pos + n > total:
            raise JavaException("BitStreamException: end of stream")
        v = (big >> (total - pos - n)) & ((1 << n) - 1)
        pos += n
        return v

    out = bytearray(count)
    if count <= 32:
        for i in range(count):
            out[i] = read_bits(8)
        return count, bytes(out), pos, True
    freqs = [0] * 256
    f2s = bytearray()
    sym_freq, sym_cum = [0] * 256, [0] * 256
    buf_len = 0
    start = 0
    while start < count:
        end = min(start + 16384, count)
        # decodeHeader
        log_range = 8 + read_bits(3)
        scale = 1 << log_range
        if read_bits(1) == 0:                                                     # FULL_ALPHABET
            alphabet = [] if read_bits(1) == 1 else list(range(256))                # ALPHABET_0 = 1, ALPHABET_256 = 0 (EntropyUtils.java:31-34)
        else:
            last = read_bits(5)
            alphabet = []
            for i in range(last + 1):
                m = read_bits(8)
                alphabet += [(i << 3) + j for j in range(8) if m & (1 << j)]
        asz = len(alphabet)
        if asz == 0:
            return start, bytes(out), pos, True
        llr = 3
        while (1 << llr) <= log_range:
            llr += 1
        if asz != 256:
            freqs = [0] * 256
        if len(f2s) < scale:
            f2s = bytearray(scale)
        chk = 8 if asz >= 64 else 6
        s = 0
        for i in range(1, asz, chk):
            log_max = read_bits(llr)
            if (1 << log_max) > scale:
                raise JavaException("BitStreamException: incorrect frequency size")
            for j in range(i, min(i + chk, asz)):
                fr = 1 if log_max == 0 else 1 + read_bits(log_max)
                if fr <= 0 or fr >= scale:
                    raise JavaException("BitStreamException: incorrect frequency")
                freqs[alphabet[j]] = fr
                s += fr
        if scale <= s:
            raise JavaException("BitStreamException: incorrect frequency (first symbol)")
        freqs[alphabet[0]] = scale - s
        s = 0
        for i in range(256):
            if freqs[i] == 0:
                continue
            if s + freqs[i] > len(f2s):
                raise JavaException("ArrayIndexOutOfBounds")
            f2s[s:s + freqs[i]] = bytes([i]) * freqs[i]
            sym_cum[i] = s
            sym_freq[i] = scale - 1 if freqs[i] >= scale else freqs[i]
            s += freqs[i]
        if asz == 1:
            out[start:end] = bytes([alphabet[0]]) * (end - start)
            start = end
            continue
        # decodeChunkV2
        value = read_bits(8)
        sz = value & 0x7F
        shift = 7
        while value >= 128:
            value = read_bits(8)
            sz |= (value & 0x7F) << shift
            if shift == 28:
                break
            shift += 7
        sz = _i32(sz)
        if sz >= (1 << 27):
            return count, bytes(out), pos, False
        st = [read_bits(32) for _ in range(4)]                                   # st0 .. st3
        buf_len = max(buf_len, 2 * (end - start), 256)
        if sz < 0 or sz > buf_len:
            raise JavaException("ArrayIndexOutOfBounds / negative length")
        buf = (read_bits(8 * sz).to_bytes(sz, "big") if sz else b"") + bytes(buf_len - sz)
        n = 0
        mask = scale - 1
        end4 = start + ((end - start) & -4)
        for i in range(start, end4, 4):
            for lane in (3, 2, 1, 0):
                x = st[lane]
                cur = f2s[x & mask]
                out[i + 3 - lane] = cur
                x = (sym_freq[cur] * (x >> log_range) + (x & mask) - sym_cum[cur]) & 0xFFFFFFFF
                if _i32(x) < (1 << 15):
                    if n + 1 >= buf_len:
                        raise JavaException("ArrayIndexOutOfBounds")
                    x = ((x << 16) | (buf[n] << 8) | buf[n + 1]) & 0xFFFFFFFF
                    n += 2
                st[lane] = x
        for i in range(end4, end):
            if n >= buf_len:
                raise JavaException("ArrayIndexOutOfBounds")
            out[i] = buf[n]
            n += 1
        if n != sz:
            return count, bytes(out), pos, False
        start = end
    return count, bytes(out), pos, True


def huffman_decode(data, nbits, count):
    """K/entropy/HuffmanDecoder.java, bit stream version >= 6: decodeV6 :353-383, readLengths :116-150 (ExpGolombDecoder.decodeByte
    :60-78 signed, HuffmanCommon.generateCanonicalCodes :71-111), buildDecodingTables :153-172, decodeChunk :386-560 (four fragments,
    64-bit states, Java shift counts taken mod 64 / mod 32), 16 KiB chunks.  -> (return value, bytes, bits consumed).  Bit stream
    exceptions and array faults are JavaException."""
    src = bytes(data)
    big = int.from_bytes(src, "big") if src else 0
    total = len(src) * 8
    pos = 0

    def read_bits(n):
        nonlocal pos
        if n == 0:
            return 0
        if n < 0 or pos + n > nbits or pos + n > total:
            raise JavaException("BitStreamException")
        v = (big >> (total - pos - n)) & ((1 << n) - 1)
        pos += n
        return v

    def varint():
        value = read_bits(8)
        res = value & 0x7F
        shift = 7
        while value >= 128:
            value = read_bits(8)
            res |= (value & 0x7F) << shift
            if shift == 28:
                break
            shift += 7
        return _i32(res)

    def s64(x):
        x &= M64
        return x - (1 << 64) if x >> 63 else x

    out = bytearray(count)
    if count == 0:
        return 0, b"", 0
    sizes = [0] * 256
    codes = [0] * 256
    table = [0] * 4096
    CH, MAXS = 16384, 12
    start = 0
    while start < count:
        n_chunk = min(CH, count - start)
        end = start + n_chunk
        if n_chunk < 32:
            for i in range(start, end):
                out[i] = read_bits(8)
            start = end
            continue
        # readLengths
        if read_bits(1) == 0:
            alphabet = [] if read_bits(1) == 1 else list(range(256))
        else:
            last = read_bits(5)
            alphabet = []
            for i in range(last + 1):
                m = read_bits(8)
                alphabet += [(i << 3) + j for j in range(8) if m & (1 << j)]
        asz = len(alphabet)
        if asz == 0:
            return start, bytes(out), pos
        cur = 2
        for s in alphabet:
            codes[s] = 0
            if read_bits(1) == 1:
                delta = 0
            else:
                log2 = 1
                while read_bits(1) == 0:
                    log2 += 1
                if log2 + 1 > 64:
                    raise JavaException("IllegalArgumentException: readBits count")
                res = read_bits(log2 + 1)
                sgn = res & 1
                res = (res >> 1) + _i32(1 << (log2 & 31)) - 1
                delta = ((res - sgn) ^ -sgn) & 0xFF
                delta = delta - 256 if delta >= 128 else delta
            cur += delta
            if cur <= 0 or cur > MAXS:
                raise JavaException("BitStreamException: incorrect size")
            sizes[s] = cur
        if asz > 1:
            alphabet = sorted(alphabet, key=lambda s: (sizes[s], s))
        code = 0
        cur_len = sizes[alphabet[0]]
        for s in alphabet:
            code <<= sizes[s] - cur_len
            cur_len = sizes[s]
            codes[s] = code
            code += 1
        if asz == 1:
            out[start:end] = bytes([alphabet[0]]) * n_chunk
            start = end
            continue
        # buildDecodingTables
        table = [7] * 4096
        length = 0
        for s in alphabet:
            if sizes[s] > length:
                length = sizes[s]
            val = (sizes[s] << 8) | s
            idx = codes[s] << (MAXS - length)
            stop = idx + (1 << (MAXS - length))
            if stop > 4096:
                raise JavaException("ArrayIndexOutOfBounds")
            for t in range(idx, stop):
                table[t] = val
        # decodeChunk
        sz_bits = [varint() for _ in range(4)]
        if min(sz_bits) < 0:
            return start, bytes(out), pos
        buf = bytearray(2 * CH)
        stride = (2 * CH) // 4
        for f in range(4):
            nb = sz_bits[f]
            if (nb >> 3) > len(buf) - f * stride:
                raise JavaException("IllegalArgumentException: readBits count")
            whole = nb >> 3
            if whole:
                if pos + 8 * whole > nbits:
                    raise JavaException("BitStreamException")
                buf[f * stride:f * stride + whole] = read_bits(8 * whole).to_bytes(whole, "big")
            if nb & 7:
                if f * stride + whole >= len(buf):
                    raise JavaException("ArrayIndexOutOfBounds")
                buf[f * stride + whole] = read_bits(nb & 7) << (8 - (nb & 7))
        frag = n_chunk // 4

        def long_at(i):
            if i < 0 or i + 8 > len(buf):
                raise JavaException("ArrayIndexOutOfBounds")
            return int.from_bytes(buf[i:i + 8], "big")

        consumed = []
        for f in range(4):
            state, bits, idx = 0, 0, f * stride
            o = start + f * frag
            n = 0

            def refill():
                nonlocal state, bits, idx
                shift = _i32((56 - bits) & -8)
                state = ((state << (shift & 63)) | ((long_at(idx) >> ((63 - shift) & 63)) >> 1)) & M64
                idx = _i32(idx + ((shift & 0xFFFFFFFF) >> 3))
                return bits + shift - MAXS

            def take(bs):
                v = table[(s64(state) >> (bs & 63)) & 0xFFF]
                return v, bs - (v >> 8)

            while n < frag - 4:
                bs = refill()
                for k in range(4):
                    v, bs = take(bs)
                    out[o + n + k] = v & 0xFF
                n += 4
                bits = bs + MAXS
            bs = refill()
            while n < frag:
                v, bs = take(bs)
                out[o + n] = v & 0xFF
                n += 1
            consumed.append(((idx - f * stride) << 3) - (bs + MAXS))
        for i in range(4 * frag, n_chunk):
            out[start + i] = read_bits(8)
        if consumed != sz_bits:
            return start, bytes(out), pos
        start = end
    return count, bytes(out), pos


def text_inverse(data, block_size, static_dict, dst_cap):
    """TextCodec.inverse :512-534 over TextCodec1.inverse :876-1031 (first byte without MASK_TEXT_CODEC) or TextCodec2.inverse
    :1410-1603 (with it; bitstream version >= 6 word indexes), written from the Java.  dst_cap = dst.length = output.length (the
    dictionary's first size comes from it: reset(output.length) :886 / :1420).  Returns (ok, bytes); a read outside the coded block
    or outside the dictionary (in Java: stale buffer bytes or an ArrayIndexOutOfBoundsException, depending on the arrays' real
    lengths) raises JavaException."""
    src = bytes(data)
    count = len(src)
    if count == 0:
        return True, b""
    if count > (1 << 30):                                                        # MAX_BLOCK_SIZE, no minimum :520-521
        return False, b""
    variant = 1 if (src[0] & _T_MASK_TEXT_CODEC) == 0 else 2                      # :527-530 (bsVersion 7)
    m = _TextCodecModel(variant, block_size, static_dict)
    m.reset(dst_cap)
    dst = bytearray(dst_cap + 4)
    src_end, dst_end = count, dst_cap

    def rd(i):                                                                   # src[i] inside the coded block only
        if i >= src_end:
            raise JavaException("read behind the coded block")
        return src[i]

    src_idx, dst_idx = 1, 0
    is_crlf = (src[0] & _T_MASK_CRLF) != 0
    if src_idx >= src_end:                                                       # a one-byte block: src[srcIdx] of :893 / :1429 reads the (longer) array's
        return True, b""                                                         # next byte, only for delimAnchor; the loop does not run: true, no output
    delim_anchor = src_idx - 1 if _t_is_text(src[src_idx]) else src_idx
    words = m.static_size
    word_run = False
    res = True
    while src_idx < src_end and dst_idx < dst_end:
        cur = src[src_idx]
        if _t_is_text(cur):
            dst[dst_idx] = cur
            src_idx += 1
            dst_idx += 1
            continue
        if src_idx > delim_anchor + 3 and _T_DELIMS[cur]:                        # a word of more than two letters ended: learn it like the encoder
            length = src_idx - delim_anchor - 1
            if length <= _T_MAX_WORD_LENGTH:
                h1 = _T_HASH1
                for i in range(delim_anchor + 1, src_idx):
                    h1 = _i32(_i32(h1 * _T_HASH1) ^ _i32(_sb(src[i]) * _T_HASH2))
                e = None
                e1 = m.dict_map.get(h1 & m.hash_mask)
                if e1 is not None and e1.hash == h1 and (e1.data >> 24) == length:
                    if src[delim_anchor + 2:delim_anchor + 1 + length] == bytes(e1.buf[e1.pos + 1:e1.pos + length]):
                        e = e1
                if e is None:
                    if (length > 3 or words < _T_THRESHOLD2) and e1 is None:
                        e = m.dict_list[words]
                        if (e.data & _T_MASK_LENGTH) >= m.static_size:
                            if m.dict_map.get(e.hash & m.hash_mask) is not None:
                                m.dict_map[e.hash & m.hash_mask] = None
                            e.buf, e.pos, e.hash, e.data = src, delim_anchor + 1, h1, (length << 24) | words
                        m.dict_map[h1 & m.hash_mask] = e
                        words += 1
                        if words >= m.dict_size:
                            if not m.expand():
                                words = m.static_size
        src_idx += 1
        flip = 0
        is_ref = (cur == _T_ESC1 or cur == _T_ESC2) if variant == 1 else (cur & 0x80) != 0
        if is_ref:
            if variant == 1:                                                     # :945-963: varint 5 + 7 + 7 bits
                idx = rd(src_idx)
                src_idx += 1
                if idx >= 128:
                    idx &= 0x7F
                    idx2 = _sb(rd(src_idx))
                    src_idx += 1
                    if idx2 & 0x80:
                        idx = ((idx & 0x1F) << 7) | (idx2 & 0x7F)
                        idx2 = rd(src_idx) & 0x7F
                        src_idx += 1
                    idx = (idx << 7) | idx2                                      # (idx2 < 0 here would have taken the branch above)
                    if idx >= m.dict_size:
                        res = False
                        break
                flip = 0x20 if cur == _T_ESC2 else 0
            else:                                                                # :1503-1540
                if cur == 0x80:                                                  # MASK_FLIP_CASE
                    flip = 0x20
                    cur = rd(src_idx)
                    src_idx += 1
                idx = cur & 0x7F
                if idx >= 64:
                    if idx >= 112:
                        idx = ((idx & 0x0F) << 16) | (rd(src_idx) << 8) | rd(src_idx + 1)
                        src_idx += 2
                    else:
                        idx = ((idx & 0x1F) << 8) | rd(src_idx)
                        src_idx += 1
                    if idx > m.dict_size:
                        res = False
                        break
                elif idx == 0:
                    res = False
                    break
                idx -= 1
            if idx < 0 or idx >= m.dict_size:
                raise JavaException("dictList[%d]" % idx)
            e = m.dict_list[idx]
            length = (e.data >> 24) & 0xFF
            if word_run and length > 1:
                if dst_idx >= len(dst):
                    raise JavaException("dst")
                dst[dst_idx] = 0x20
                dst_idx += 1
            if e.pos < 0 or dst_idx + length >= dst_end:
                res = False
                break
            dst[dst_idx] = e.buf[e.pos] ^ flip
            dst_idx += 1
            if length > 1:
                dst[dst_idx:dst_idx + length - 1] = e.buf[e.pos + 1:e.pos + length]
                dst_idx += length - 1
                word_run = True
                delim_anchor = src_idx
            else:
                word_run = False
                delim_anchor = src_idx - 1
        else:
            if variant == 2 and cur == _T_ESC1:                                  # an escaped byte :1580-1581
                dst[dst_idx] = rd(src_idx)
                dst_idx += 1
                src_idx += 1
            else:
                if is_crlf and cur == _T_LF:
                    dst[dst_idx] = _T_CR
                    dst_idx += 1
                    if dst_idx >= dst_end:
                        res = False
                        break
                dst[dst_idx] = cur
                dst_idx += 1
            word_run = False
            delim_anchor = src_idx - 1
    return (res and src_idx == src_end), bytes(dst[:dst_idx])
