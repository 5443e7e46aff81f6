"""Inputs for the TEXT / UTF stages: English-like prose over the static dictionary's words plus invented ones, CRLF text,
XML-like text, text with the escape bytes and bytes >= 0x80 in it, UTF-8 (Cyrillic, CJK, emoji), and non-text."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dict_words():
    txt = open(os.path.join(ROOT, "kanzi_amd", "csrc", "kz_text_dict.h")).read()
    return re.findall(r"[A-Z][a-z]+", "".join(re.findall(r'"([A-Za-z]+)"', txt)))


def english(n, seed, crlf=False, invented=400, sprinkle=b""):
    rng = np.random.default_rng(seed)
    wl = dict_words()
    letters = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz", dtype=np.uint8)
    made = ["".join(chr(c) for c in rng.choice(letters, rng.integers(3, 12))) for _ in range(invented)]
    out, tot = [], 0
    while tot < n:
        k = int(rng.integers(3, 15))
        ws = []
        for _ in range(k):
            w = str(rng.choice(wl)) if rng.random() < 0.7 else made[int(rng.integers(0, len(made)))]
            r = rng.random()
            ws.append(w.lower() if r < 0.8 else (w.capitalize() if r < 0.95 else w.upper()))
        s = " ".join(ws) + ("." if rng.random() < 0.7 else ", " + str(int(rng.integers(0, 9999))) + ";") + ("\r\n" if crlf else "\n")
        b = s.encode()
        if sprinkle and rng.random() < 0.05:
            b += sprinkle
        out.append(b)
        tot += len(b)
    return b"".join(out)[:n]


def many_words(n, seed, alphabet=20):
    """mostly distinct invented words: the dictionary doubles up to 2^19 entries and wraps"""
    rng = np.random.default_rng(seed)
    letters = rng.integers(97, 97 + alphabet, n, dtype=np.uint8)
    cuts = rng.random(n) < 0.16
    letters[cuts] = 32
    letters[rng.random(n) < 0.01] = 10
    return letters.tobytes()


def xml(n, seed):
    rng = np.random.default_rng(seed)
    wl = dict_words()
    out, tot = [], 0
    while tot < n:
        t = str(rng.choice(wl)).lower()
        s = "<%s id=\"%d\">%s &amp; %s &lt;%s&gt;</%s>\n" % (t, int(rng.integers(0, 1000)), rng.choice(wl), rng.choice(wl), rng.choice(wl), t)
        out.append(s.encode())
        tot += len(s)
    return b"".join(out)[:n]


def utf8(n, seed, bom=False):
    rng = np.random.default_rng(seed)
    cps = list(range(0x410, 0x450)) * 3 + list(range(0x4E00, 0x4E60)) + [0x1F600, 0x1F601, 0x1F4A9] + [32] * 40 + [10] * 4 + list(range(0x61, 0x7B)) * 2
    s = "".join(chr(int(c)) for c in rng.choice(cps, n))
    b = s.encode("utf-8")
    return ((b"\xef\xbb\xbf" if bom else b"") + b)[:n]


def cases(scale=1):
    """name -> bytes"""
    rng = np.random.default_rng(99)
    c = {
        "english": english(300000 * scale, 1),
        "english_crlf": english(120000 * scale, 2, crlf=True),
        "english_escapes": english(90000 * scale, 3, sprinkle=bytes([0x0F, 0x0E, 0x80, 0xFF, 0x0F, 0x0F, 0x81])),
        "english_lone_cr": english(50000 * scale, 4, crlf=True).replace(b"\r\n", b"\r\n", 1) + b"\rtrailing carriage return\n" + english(5000, 5),
        "many_words": many_words(600000 * scale, 6),
        "xml": xml(150000 * scale, 7),
        "utf8": utf8(120000 * scale, 8),
        "utf8_bom": utf8(60000 * scale, 9, bom=True),
        "utf8_cut": utf8(60000 * scale, 10)[1:],
        "random": rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(),
        "digits": bytes(rng.choice(np.frombuffer(b"0123456789 ,.", dtype=np.uint8), 50000)),
        "dna": bytes(rng.choice(np.frombuffer(b"acgt", dtype=np.uint8), 50000)),
        "spaces_then_text": b" " * 700 + english(40000, 11),
        "short": english(1023, 12),
        "min": english(1024, 13),
        "gif_magic_text": b"GIF8" + english(30000, 14),
    }
    return c


def bulk_text(n, seed, kind="english"):
    """Full-size (MiB) text blocks for bench.py's level-exact rows, assembled with array operations only (the generators above
    loop per line in Python).  kind: "english" (dictionary + invented words, Zipf reuse, sentence punctuation, LF or CRLF by
    seed), "xml" (tags around dictionary words), "utf8" (Cyrillic words between ASCII ones)."""
    rng = np.random.default_rng(seed)
    wl = [w.lower() for w in dict_words()]
    letters = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz", dtype=np.uint8)
    made = ["".join(chr(c) for c in rng.choice(letters, rng.integers(3, 12))) for _ in range(600)]
    eol = "\r\n" if (seed & 3) == 3 else "\n"
    vocab = []
    if kind == "utf8":
        cyr = ["".join(chr(int(c)) for c in rng.integers(0x430, 0x450, rng.integers(2, 10))) for _ in range(1500)]
        for w in cyr + wl[:300]:
            vocab += [w + " ", w + ", ", w + "." + eol]
    elif kind == "xml":
        for w in wl[:400]:
            vocab += ["<" + w + ">", "</" + w + ">" + eol, w + " ", w + " &amp; ", "<" + w + " id=\"" + str(len(w) * 37) + "\">"]
    else:
        for w in wl + made:
            vocab += [w + " ", w + " ", w + " ", w.capitalize() + " ", w + ", ", w + "." + eol, w.upper() + " "]
    enc = [v.encode("utf-8") for v in vocab]
    lens = np.array([len(e) for e in enc], dtype=np.int64)
    starts = np.concatenate(([0], np.cumsum(lens)[:-1]))
    blob = np.frombuffer(b"".join(enc), dtype=np.uint8)
    need = int(n // max(lens.mean() * 0.8, 1)) + 64
    ids = (len(enc) * rng.random(need) ** 2.5).astype(np.int64)           # power-law reuse, no single word dominating
    ids = rng.permutation(len(enc))[ids]                              # frequent ids spread over the vocabulary
    L = lens[ids]
    total = int(L.sum())
    off = np.cumsum(L) - L
    src = np.repeat(starts[ids] - off, L) + np.arange(total)
    out = blob[src]
    if len(out) < n:
        out = np.resize(out, n)
    return np.ascontiguousarray(out[:n])


def vocab_words(n, seed, vocab, lo=3, hi=9):
    """n bytes of words drawn uniformly from `vocab` distinct invented words (array operations only): with ~150 000 words in a
    4 MiB block TextCodec's word list grows past 2^17 entries (its map doubles, TextCodec.java:1071-1081) without wrapping at 2^19"""
    rng = np.random.default_rng(seed)
    L = rng.integers(lo, hi + 1, vocab)
    starts = np.concatenate(([0], np.cumsum(L + 1)[:-1]))
    blob = rng.integers(97, 123, int((L + 1).sum()), dtype=np.uint8)
    blob[starts + L] = 32
    need = int(n // (L.mean() + 1)) + 64
    ids = rng.integers(0, vocab, need)
    Li = L[ids] + 1
    off = np.cumsum(Li) - Li
    src = np.repeat(starts[ids] - off, Li) + np.arange(int(Li.sum()))
    out = blob[src]
    out[rng.random(len(out)) < 0.002] = 10
    if len(out) < n:
        out = np.resize(out, n)
    return np.ascontiguousarray(out[:n])


def big_blocks(bs):
    """Blocks of `bs` bytes (meant for BASELINE's 4 MiB) for the device TEXT / UTF parity tests: English (LF), English (CRLF), XML,
    UTF-8 (Cyrillic), invented words that push the word list past 2^17 entries, invented words that make it wrap at 2^19,
    English with escape bytes and bytes >= 0x80 sprinkled in, binary, and a ragged last block"""
    rng = np.random.default_rng(4242)
    esc = bulk_text(bs, 31, "english").copy()
    pos = rng.integers(0, bs, bs // 3000)
    esc[pos] = rng.choice(np.array([0x0F, 0x0E, 0x80, 0xFF, 0x0D, 0xC3], dtype=np.uint8), len(pos))
    blocks = [bulk_text(bs, 20, "english"), bulk_text(bs, 23, "english"), bulk_text(bs, 21, "xml"), bulk_text(bs, 22, "utf8"),
              vocab_words(bs, 24, 150000), np.frombuffer(many_words(bs, 25), dtype=np.uint8), esc,
              rng.integers(0, 256, bs, dtype=np.uint8), bulk_text(bs // 3 + 12345, 26, "english")]
    return [np.ascontiguousarray(b).tobytes() for b in blocks]
