"""Developer diagnostic (not a pytest): runs every HIP stage against the oracle and reports ALL
mismatches instead of stopping at the first. Usage: python tests/gpu_diag.py [size]"""
import ctypes
import os
import sys
import time
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kanzi_amd as kz
import oracle
import datagen

ctx = kz.Context(0)
FAILS = []


def first_diff(a, b):
    a = np.frombuffer(a, dtype=np.uint8); b = np.frombuffer(b, dtype=np.uint8)
    n = min(len(a), len(b))
    d = np.nonzero(a[:n] != b[:n])[0]
    return (int(d[0]) if len(d) else n), len(d)


def check(name, got, exp):
    if got == exp:
        print("  ok   %s (%d bytes)" % (name, len(exp)))
        return True
    fd, nd = first_diff(got, exp)
    print("  FAIL %s: len got %d exp %d, first diff @%d, %d diffs; got %s exp %s" % (
        name, len(got), len(exp), fd, nd, got[fd:fd + 8].hex(), exp[fd:fd + 8].hex()))
    FAILS.append(name)
    return False


def tfwd(tname, ttype, data):
    cap = ctx.lib.kz_transform_max_encoded_len(ttype, len(data))
    out = np.zeros(cap + 64, dtype=np.uint8)
    p = ctypes.c_int32(0)
    a = np.frombuffer(data, dtype=np.uint8)
    rc = ctx.lib.kz_transform_forward(ctx.h, ttype, a.ctypes.data, len(a), out.ctypes.data, cap, ctypes.addressof(p))
    if rc < 0:
        raise RuntimeError("%s fwd rc=%d %s" % (tname, rc, ctx.error()))
    return rc == 1, out[:p.value].tobytes()


def tinv(tname, ttype, data, cap):
    out = np.zeros(cap + 64, dtype=np.uint8)
    p = ctypes.c_int32(0)
    a = np.frombuffer(data, dtype=np.uint8)
    rc = ctx.lib.kz_transform_inverse(ctx.h, ttype, a.ctypes.data, len(a), out.ctypes.data, cap, ctypes.addressof(p))
    if rc < 0:
        raise RuntimeError("%s inv rc=%d %s" % (tname, rc, ctx.error()))
    return rc == 1, out[:p.value].tobytes()


def inputs(n):
    rng = np.random.default_rng(7)
    yield "text", datagen.block(0, n).tobytes()
    yield "skew", datagen.block(1, n).tobytes()
    yield "records", datagen.block(2, n).tobytes()
    yield "random", datagen.block(3, n).tobytes()
    yield "zeros90", datagen.block(4, n).tobytes()
    yield "allzero", bytes(n)
    yield "ramp", bytes((np.arange(n) & 0xFF).astype(np.uint8))
    yield "mississippi", (b"mississippi" * (n // 11 + 1))[:n]
    yield "ff", bytes([0xFF, 0xFE, 0, 0, 0, 1, 2] * (n // 7 + 1))[:n]


def stage_tests(n):
    for nm, data in inputs(n):
        print("[%s n=%d]" % (nm, len(data)))
        try:
            # BWT
            ok_o, bwt_o = oracle.transform_forward("BWT", data)
            t0 = time.time(); ok_g, bwt_g = tfwd("BWT", kz.BWT_TYPE, data); t1 = time.time()
            print("   (bwt fwd %.1f ms)" % ((t1 - t0) * 1e3))
            if ok_o != ok_g: print("  FAIL bwt applied flag", ok_o, ok_g); FAILS.append(nm + ":bwt_flag")
            check(nm + ":bwt_fwd", bwt_g, bwt_o)
            ok_g, back = tinv("BWT", kz.BWT_TYPE, bwt_o, len(data) + 64)
            check(nm + ":bwt_inv", back, data)
            # RANK / MTF
            for tname, tid in (("RANK", kz.RANK_TYPE), ("MTFT", kz.MTFT_TYPE)):
                _, r_o = oracle.transform_forward(tname, bwt_o)
                _, r_g = tfwd(tname, tid, bwt_o)
                check(nm + ":%s_fwd" % tname, r_g, r_o)
                _, rb = tinv(tname, tid, r_o, len(r_o) + 64)
                check(nm + ":%s_inv" % tname, rb, bwt_o)
            _, rank_o = oracle.transform_forward("RANK", bwt_o)
            # ZRLT on rank output and on raw data
            for label, zin in (("rank", rank_o), ("raw", data)):
                okz_o, z_o = oracle.transform_forward("ZRLT", zin)
                okz_g, z_g = tfwd("ZRLT", kz.ZRLT_TYPE, zin)
                if okz_o != okz_g:
                    print("  FAIL zrlt(%s) applied flag oracle=%s gpu=%s" % (label, okz_o, okz_g)); FAILS.append(nm + ":zrlt_flag_" + label)
                elif okz_o:
                    check(nm + ":zrlt_fwd_" + label, z_g, z_o)
                else:
                    print("  ok   zrlt(%s) declined on both" % label)
                if okz_o:
                    okb, zb = tinv("ZRLT", kz.ZRLT_TYPE, z_o, len(zin) + 1024)
                    check(nm + ":zrlt_inv_" + label, zb, zin)
            # ANS0 on zrlt(rank) output (or rank output if zrlt declined)
            okz_o, z_o = oracle.transform_forward("ZRLT", rank_o)
            ein = z_o if okz_o else rank_o
            for label, e_in in (("chain", ein), ("raw", data)):
                eb_o, nb_o = oracle.entropy_encode("ANS0", e_in)
                a = np.frombuffer(e_in, dtype=np.uint8)
                cap = int(ctx.lib.kz_max_block_stream_bytes(len(a)))
                out = np.zeros(cap, dtype=np.uint8)
                nb_g = ctx.lib.kz_entropy_encode(ctx.h, kz.E_ANS0, a.ctypes.data, len(a), out.ctypes.data, cap)
                if nb_g != nb_o:
                    print("  FAIL ans bits(%s) gpu=%d oracle=%d" % (label, nb_g, nb_o)); FAILS.append(nm + ":ans_bits_" + label)
                check(nm + ":ans_enc_" + label, out[:(max(nb_g, 0) + 7) // 8].tobytes(), eb_o)
                src = np.frombuffer(eb_o + b"\0" * 64, dtype=np.uint8)
                dec = np.zeros(len(a) + 8, dtype=np.uint8)
                used = ctypes.c_int64(0)
                rc = ctx.lib.kz_entropy_decode(ctx.h, kz.E_ANS0, src.ctypes.data, nb_o, dec.ctypes.data, len(a), ctypes.addressof(used))
                if rc != len(a):
                    print("  FAIL ans dec(%s) rc=%d %s" % (label, rc, ctx.error())); FAILS.append(nm + ":ans_dec_rc_" + label)
                else:
                    check(nm + ":ans_dec_" + label, dec[:len(a)].tobytes(), e_in)
        except Exception as e:
            traceback.print_exc()
            FAILS.append(nm + ":exception")


def stream_tests(bs, nblocks):
    data = datagen.stream(nblocks, bs).tobytes() + b"tail-bytes-0123"
    for chain, ent in (("BWT+RANK+ZRLT", "ANS0"), ("BWT+MTFT+ZRLT", "ANS0"), ("ZRLT", "NONE"), ("BWT", "ANS0")):
        print("[stream %s & %s bs=%d n=%d]" % (chain, ent, bs, len(data)))
        try:
            ref = oracle.compress(chain, ent, bs, data, jobs=8)
            cos = kz.CompressedOutputStream(ctx, chain, ent, bs)
            cos.write(data); t0 = time.time(); cos.close(); t1 = time.time()
            check("stream:%s&%s" % (chain, ent), cos.output, ref)
            t2 = time.time(); back = kz.CompressedInputStream(ctx, ref).read(); t3 = time.time()
            check("stream_dec:%s&%s" % (chain, ent), back, data)
            print("   enc %.1f ms dec %.1f ms" % ((t1 - t0) * 1e3, (t3 - t2) * 1e3))
        except Exception:
            traceback.print_exc()
            FAILS.append("stream:%s&%s:exception" % (chain, ent))


if __name__ == "__main__":
    sizes = [int(x) for x in sys.argv[1:]] or [1000, 70000]
    for n in sizes:
        stage_tests(n)
    stream_tests(65536, 7)
    print("=" * 60)
    print("FAILS: %d" % len(FAILS))
    for f in FAILS:
        print("  ", f)
    sys.exit(1 if FAILS else 0)
