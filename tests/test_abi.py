"""CPU tests of the boundary: the C-ABI library loads and exports every symbol include/kanzi_hip.h
declares (no compute without a GPU), host-only helpers agree with the oracle, and the product path
fails loudly when no GPU is present."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

import kanzi_amd as kz
import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "kanzi_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(kz_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 25
    L = ctypes.CDLL(kz.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), "libkanzi_hip.so does not export %s" % name
    assert sorted(kz.ABI_SYMBOLS) == declared, "python binding list out of sync with the header"
    assert kz.load_library().kz_abi_version() == 3


def test_max_encoded_len_matches_reference_values(built):
    L = kz.load_library()
    O = oracle.lib()
    for n in (0, 1, 100, 1024, 1025, 65536, 4 * 1024 * 1024):
        # BWT n+33 (BWTBlockCodec.java:40,222); SRT n+1024 (SRT.java:30,365); LZ/LZX (LZCodec.java:961-964); ZRLT/SBRT n
        assert L.kz_transform_max_encoded_len(kz.BWT_TYPE, n) == n + 33
        assert L.kz_transform_max_encoded_len(kz.SRT_TYPE, n) == n + 1024
        assert L.kz_transform_max_encoded_len(kz.LZ_TYPE, n) == ((n + 16) if n <= 1024 else n + n // 64) + 2
        assert L.kz_transform_max_encoded_len(kz.ZRLT_TYPE, n) == n
        for t in (kz.BWT_TYPE, kz.SRT_TYPE, kz.LZ_TYPE, kz.LZX_TYPE, kz.ZRLT_TYPE, kz.RANK_TYPE, kz.MTFT_TYPE):
            assert L.kz_transform_max_encoded_len(t, n) == O.kzo_transform_max_encoded_len(t, n)


def test_transform_type_word(built):
    # TransformFactory.java:29-31,144-157: 8 slots x 6 bits, first transform in the top slot
    assert kz.transform_type("BWT+RANK+ZRLT") == (1 << 42) | (8 << 36) | (6 << 30)
    assert kz.transform_type("BWT+RANK+ZRLT") == oracle.ttype("BWT+RANK+ZRLT")
    ids = (ctypes.c_int32 * 3)(1, 8, 6)
    assert kz.load_library().kz_transform_type(ids, 3) == kz.transform_type("BWT+RANK+ZRLT")
    with pytest.raises(ValueError):
        kz.transform_type("+".join(["ZRLT"] * 9))


def test_host_container_assembly_matches_oracle(built):
    """kz_knz_assemble (product, host-only) over oracle-encoded block streams == oracle stream."""
    rng = np.random.default_rng(11)
    data = bytes(np.minimum(rng.geometric(0.08, 150000) - 1, 255).astype(np.uint8)) + bytes(10000) + b"xyz"
    bs = 32768
    chain, ent = "BWT+RANK+ZRLT", "ANS0"
    blocks = [data[i:i + bs] for i in range(0, len(data), bs)]
    streams, bits = [], []
    for b in blocks:
        s, w, _, _ = oracle.encode_block(chain, ent, b)
        streams.append(s)
        bits.append(w)
    knz = kz.knz_assemble(chain, ent, bs, len(data), streams, bits)
    assert knz == oracle.compress(chain, ent, bs, data, jobs=2)
    idx = kz.knz_index(knz)
    assert idx["blockSize"] == bs and idx["inputSize"] == len(data)
    assert idx["transform"] == kz.transform_type(chain) and idx["entropy"] == kz.E_ANS0
    assert [w for _, w in idx["blocks"]] == bits
    for (off, w), s in zip(idx["blocks"], streams):
        assert kz.extract_bits(knz, off, w) == s[:(w + 7) // 8]


def test_stream_header_tamper_widths(built):
    # T/test/TestCompressedStream.java:177-291 pins the header field widths 32/4/2/5/48/28/2/16k/15/24
    knz = oracle.compress("ZRLT", "NONE", 1024, b"a" * 3000, jobs=1)
    assert kz.knz_index(knz)["blockSize"] == 1024
    bad = bytearray(knz)
    bad[10] ^= 0x10            # flip a bit inside the transform word -> header checksum must fail
    with pytest.raises(kz.KanziError) as e:
        kz.knz_index(bytes(bad))
    assert e.value.code == 19  # ERR_CRC_CHECK


def test_product_path_fails_loudly_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        kz.Context(0)
    assert kz.load_library().kz_ctx_create(0) is None


def test_no_oracle_in_product_path():
    """The product (kanzi_amd/, include/, bench timing path) must never import or link the oracle."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "kanzi_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"kzo_|libkzo|import oracle|from oracle|oracle/", txt):
                    bad.append(f)
    assert not bad, bad


def test_reference_levels_map_to_chains():
    """BlockCompressor.getTransformAndCodec (K/app/BlockCompressor.java:537-573): levels 0-3, 5 and 6 are made of built stages
    (TEXT and UTF as host stages), the rest are refused with ERR_INVALID_CODEC."""
    assert kz.level_chain(1) == ("LZX", "NONE") and kz.level_chain(2) == ("DNA+LZ", "HUFFMAN") and kz.level_chain(0) == ("NONE", "NONE")
    assert kz.level_chain(3) == ("TEXT+UTF+PACK+MM+LZX", "HUFFMAN")
    assert kz.level_chain(5) == ("TEXT+UTF+BWT+RANK+ZRLT", "ANS0")
    assert kz.level_chain(6) == ("TEXT+UTF+BWT+SRT+ZRLT", "FPAQ")
    assert kz.level_chain(5, allow_partial=True) == ("BWT+RANK+ZRLT", "ANS0")
    for lvl in (4, 7, 8, 9):
        with pytest.raises(kz.KanziError) as e:
            kz.level_chain(lvl)
        assert e.value.code == 3
    for t, _ in (kz.level_chain(l) for l in (0, 1, 2, 3, 5, 6)):
        kz.transform_type(t)


def test_jni_shim_compiles_against_the_stub_header():
    """integration/jni/kanzi_hip_jni.c cannot be built here (no JDK); it is kept compilable with gcc -fsyntax-only against
    integration/jni/stub/jni.h, which declares the JNI functions it uses with the specification's signatures."""
    import subprocess
    src = os.path.join(ROOT, "integration", "jni", "kanzi_hip_jni.c")
    r = subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-std=c11", "-I" + os.path.join(ROOT, "integration", "jni", "stub"),
                        "-I" + os.path.join(ROOT, "include"), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # every native method KanziHip.java declares has its JNI function, and vice versa
    java = open(os.path.join(ROOT, "integration", "java", "KanziHip.java")).read()
    natives = set(re.findall(r"static native \w+ (\w+)\(", java))
    funcs = set(re.findall(r"CLS\((\w+)\)\(", open(src).read()))
    assert natives == funcs and "decodeBlocks" in natives, (natives ^ funcs)


def test_reference_patch_applies():
    """integration/kanzi-hip.patch is a diff against the reference's Java tree (factories, DecodingTask, processBlock)."""
    import shutil
    import subprocess
    import tempfile
    ref = "/root/reference/java"
    if not os.path.isdir(ref):
        pytest.skip("the reference tree only exists in the build container")
    with tempfile.TemporaryDirectory() as tmp:
        for rel in ("transform/TransformFactory.java", "entropy/EntropyCodecFactory.java", "io/CompressedOutputStream.java", "io/CompressedInputStream.java"):
            dst = os.path.join(tmp, "java/src/main/java/io/github/flanglet/kanzi", rel)
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            shutil.copy(os.path.join(ref, "src/main/java/io/github/flanglet/kanzi", rel), dst)
        r = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(ROOT, "integration", "kanzi-hip.patch")], cwd=tmp, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr


def test_patch_hooks_exist_in_the_java_adapters():
    """Every io.github.flanglet.kanzi.hip.<Class>.<method>( the patch calls is a static method of integration/java/<Class>.java with
    that many parameters (no JDK here: the cheapest check that the patch and the adapters agree), both batched hooks are there (the
    writer's processBlock and, since round 3, the reader's), and the Java sources have balanced brackets."""
    patch = open(os.path.join(ROOT, "integration", "kanzi-hip.patch")).read()
    added = "\n".join(l[1:] for l in patch.splitlines() if l.startswith("+") and not l.startswith("+++"))
    jdir = os.path.join(ROOT, "integration", "java")

    def args_at(txt, i):                                                  # number of top-level arguments of the call opening at txt[i] == "("
        depth, n, any_ = 0, 0, False
        for ch in txt[i:]:
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
                if depth == 0:
                    return n + 1 if any_ else 0
            elif ch == "," and depth == 1:
                n += 1
            elif not ch.isspace():
                any_ = True
        raise AssertionError("unbalanced")

    calls = list(re.finditer(r"io\.github\.flanglet\.kanzi\.hip\.(\w+)\.(\w+)\(", added))
    assert {(m.group(1), m.group(2)) for m in calls} >= {("HipBlockBatch", "encode"), ("HipBlockBatch", "decode"), ("HipRuntime", "enabled")}
    for m in calls:
        cls, meth = m.group(1), m.group(2)
        src = re.sub(r"<[\w, ]*>", "", open(os.path.join(jdir, cls + ".java")).read())     # generic parameters carry commas
        d = re.search(r"public static [\w\[\]]+ " + meth + r"\(", src)
        assert d, (cls, meth)
        assert args_at(src, d.end() - 1) == args_at(added, m.end() - 1), (cls, meth)
    assert "CompressedInputStream.java" in patch and "HipBlockBatch.decode" in patch and "this.maxBufferId" in added
    for f in os.listdir(jdir):
        src = re.sub(r'"(\\.|[^"\\])*"|//[^\n]*|/\*.*?\*/', "", open(os.path.join(jdir, f)).read(), flags=re.S)
        for a, b in ("()", "{}", "[]"):
            assert src.count(a) == src.count(b), (f, a)


def test_fpaq_inline_asm_lds_reads_are_waited_for():
    """kz_fpaq.hip issues ds_read_b32 from one inline-asm statement and waits for it in a later one; the compiler does not track
    that.  Walk the device assembly this hipcc produces and make sure nothing touches those registers before the s_waitcnt."""
    import shutil
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import asm_lds_hazards as H
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc")
    # the checker itself: a copy between the read and the wait is reported, the same copy behind the wait is not
    bad, reads = H.hazards("ds_read_b32 v5, v1\nv_mov_b32 v9, v5\ns_waitcnt lgkmcnt(0)\nv_mov_b32 v8, v5")
    assert reads == 1 and [b[0] for b in bad] == [1]
    bad, _ = H.hazards("ds_read_b32 v5, v1\nds_read_b32 v6, v1 offset:4\nds_write_b32 v2, v3\ns_waitcnt lgkmcnt(1)\nv_cndmask_b32 v7, v5, v6, vcc")
    assert bad == []
    text = H.device_asm(os.path.join(ROOT, "kanzi_amd", "csrc", "kz_fpaq.hip"), hipcc)
    for name in H.DEFAULT_KERNELS:
        bad, reads = H.hazards(H.kernel_body(text, name))
        assert reads > 0 and bad == [], (name, bad[:5])
