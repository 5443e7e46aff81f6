"""kanzi_amd -- MI355X-native (gfx950 HIP) implementation of Kanzi's per-block hot path.

Python host-side mirror of the reference's plugin interfaces, bound to the C-ABI in
``include/kanzi_hip.h`` with ctypes.  The product path is the HIP library ``libkanzi_hip.so``;
there is NO CPU fallback: importing works everywhere (so the ABI can be inspected), but creating a
:class:`Context` without the library or without a GPU raises.

Reference interfaces mirrored (K/ = java/src/main/java/io/github/flanglet/kanzi/):
  * ``ByteTransform``  K/ByteTransform.java:36,48,56      -> :class:`BWTBlockCodec`, :class:`SBRT`, :class:`ZRLT`
  * ``EntropyEncoder`` K/EntropyEncoder.java:34,41,48     -> :class:`ANSRangeEncoder`, :class:`HuffmanEncoder`, :class:`NullEntropyEncoder`
  * ``EntropyDecoder`` K/EntropyDecoder.java:33           -> :class:`ANSRangeDecoder`, :class:`NullEntropyDecoder`
  * ``Sequence`` / block span of EncodingTask.encodeBlock -> :func:`encode_blocks` / :func:`decode_blocks`
  * ``CompressedOutputStream`` / ``CompressedInputStream`` -> same-named classes (whole .knz stream)
"""
# The decoder keeps four HIP streams busy side by side; HIP shares 4 hardware queues among all streams of a process unless
# GPU_MAX_HW_QUEUES says otherwise WHEN THE RUNTIME STARTS (first HIP call, also PyTorch's).  That is the APPLICATION's setting
# (bench.py and tests/conftest.py export GPU_MAX_HW_QUEUES=8 before importing torch); neither this binding nor the library
# touches the process environment.  Without it the library measures that its streams share queues and falls back to a
# three-stream schedule.
import ctypes
import weakref
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkanzi_hip.so")

# transform ids (K/transform/TransformFactory.java:36-60) and entropy ids (K/entropy/EntropyCodecFactory.java)
NONE_TYPE, BWT_TYPE, LZ_TYPE, ZRLT_TYPE, MTFT_TYPE, RANK_TYPE, SRT_TYPE, MM_TYPE, LZX_TYPE, PACK_TYPE, DNA_TYPE = 0, 1, 3, 6, 7, 8, 13, 15, 16, 18, 19
E_NONE, E_HUFFMAN, E_FPAQ, E_ANS0 = 0, 1, 2, 5
TRANSFORM_IDS = {"NONE": 0, "BWT": 1, "LZ": 3, "ZRLT": 6, "MTFT": 7, "RANK": 8, "TEXT": 10, "SRT": 13, "MM": 15, "LZX": 16, "UTF": 17, "PACK": 18, "DNA": 19}
TEXT_TYPE, UTF_TYPE = 10, 17
# Global.DataType (K/Global.java:40-80), numbered as KZ_DT_* in include/kanzi_hip.h
DATA_TYPES = {"UNDEFINED": 0, "DNA": 1, "SMALL_ALPHABET": 2, "TEXT": 3, "MULTIMEDIA": 4, "EXE": 5, "NUMERIC": 6, "BASE64": 7, "BIN": 8, "UTF8": 9}
ENTROPY_IDS = {"NONE": 0, "HUFFMAN": 1, "FPAQ": 2, "ANS0": 5}
MEM_HOST, MEM_DEVICE = 0, 1

STAGE_NAMES = ["bwt_fwd", "sbrt_fwd", "zrlt_fwd", "entropy_enc", "frame_enc",
               "entropy_dec", "zrlt_inv", "sbrt_inv", "bwt_inv", "frame_dec", "lz_fwd", "lz_inv", "srt_fwd", "srt_inv", "host_fwd", "host_inv"]


class KanziError(RuntimeError):
    """Mirror of KanziIOException(code) (K/io/KanziIOException.java); code = K/Error.java value."""

    def __init__(self, code, msg=""):
        super().__init__("kanzi error %d %s" % (code, msg))
        self.code = code


class BlockResult(ctypes.Structure):
    _fields_ = [("bits", ctypes.c_int64), ("length", ctypes.c_int32), ("status", ctypes.c_int32),
                ("skipFlags", ctypes.c_uint8), ("mode", ctypes.c_uint8), ("pad", ctypes.c_uint8 * 6)]


_lib = None


def load_library():
    """Load libkanzi_hip.so and declare every symbol of include/kanzi_hip.h. Raises if missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libkanzi_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the product path)")
    L = ctypes.CDLL(LIB_PATH)
    c = ctypes
    vp, u8p, i32p, i64p = c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p
    sig = {
        "kz_abi_version": (c.c_int32, []),
        "kz_ctx_create": (vp, [c.c_int32]),
        "kz_ctx_destroy": (None, [vp]),
        "kz_last_error": (c.c_char_p, [vp]),
        "kz_ctx_stream": (vp, [vp]),
        "kz_pin_to_device_numa": (c.c_int32, [c.c_int32]),
        "kz_host_cpus": (c.c_int32, []),
        "kz_host_share": (c.c_int32, [c.c_int32]),
        "kz_ctx_set_checksum": (c.c_int32, [vp, c.c_int32]),
        "kz_ctx_set_skip_blocks": (c.c_int32, [vp, c.c_int32]),
        "kz_ctx_set_block_size": (c.c_int32, [vp, c.c_int32]),
        "kz_ctx_reload_switches": (None, [vp]),
        "kz_host_stage_blocks": (c.c_int64, [c.c_int32, c.c_int32]),
        "kz_ctx_set_entropy": (c.c_int32, [vp, c.c_uint32]),
        "kz_ctx_set_data_type": (c.c_int32, [vp, c.c_int32]),
        "kz_ctx_get_data_type": (c.c_int32, [vp]),
        "kz_ctx_reset": (c.c_int32, [vp]),
        "kz_transform_forward": (c.c_int32, [vp, c.c_uint32, u8p, c.c_int32, u8p, c.c_int32, i32p]),
        "kz_transform_inverse": (c.c_int32, [vp, c.c_uint32, u8p, c.c_int32, u8p, c.c_int32, i32p]),
        "kz_transform_max_encoded_len": (c.c_int32, [c.c_uint32, c.c_int32]),
        "kz_host_stage_forward": (c.c_int32, [c.c_uint32, c.c_uint32, c.c_int32, i32p, u8p, c.c_int32, u8p, c.c_int32, i32p]),
        "kz_host_stage_inverse": (c.c_int32, [c.c_uint32, c.c_int32, u8p, c.c_int32, u8p, c.c_int32, i32p]),
        "kz_entropy_encode": (c.c_int64, [vp, c.c_uint32, u8p, c.c_int32, u8p, c.c_int64]),
        "kz_entropy_decode": (c.c_int32, [vp, c.c_uint32, u8p, c.c_int64, u8p, c.c_int32, i64p]),
        "kz_encode_blocks": (c.c_int32, [vp, c.c_uint64, c.c_uint32, u8p, c.c_int64, i32p, c.c_int32, u8p, c.c_int64, vp, c.c_int32]),
        "kz_decode_blocks": (c.c_int32, [vp, c.c_uint64, c.c_uint32, c.c_int32, u8p, c.c_int64, i64p, c.c_int32, u8p, c.c_int64, vp, c.c_int32]),
        "kz_max_block_stream_bytes": (c.c_int64, [c.c_int32]),
        "kz_submit_encode_blocks": (c.c_int64, [vp, c.c_uint64, c.c_uint32, u8p, c.c_int64, i32p, c.c_int32, u8p, c.c_int64, vp, c.c_int32]),
        "kz_submit_decode_blocks": (c.c_int64, [vp, c.c_uint64, c.c_uint32, c.c_int32, u8p, c.c_int64, i64p, c.c_int32, u8p, c.c_int64, vp, c.c_int32]),
        "kz_wait": (c.c_int32, [vp, c.c_int64]),
        "kz_poll": (c.c_int32, [vp, c.c_int64]),
        "kz_compress": (c.c_int64, [vp, c.c_uint64, c.c_uint32, c.c_int32, u8p, c.c_int64, u8p, c.c_int64]),
        "kz_decompress": (c.c_int64, [vp, u8p, c.c_int64, u8p, c.c_int64]),
        "kz_compress_bound": (c.c_int64, [c.c_int64, c.c_int32]),
        "kz_transform_type": (c.c_uint64, [i32p, c.c_int32]),
        "kz_knz_assemble": (c.c_int64, [c.c_uint64, c.c_uint32, c.c_int32, c.c_int64, c.c_int32, u8p, c.c_int64, i64p, c.c_int32, u8p, c.c_int64]),
        "kz_knz_writer_open": (vp, [c.c_uint64, c.c_uint32, c.c_int32, c.c_int64, c.c_int32, u8p, c.c_int64]),
        "kz_knz_writer_add": (c.c_int32, [vp, u8p, c.c_int64]),
        "kz_knz_writer_close": (c.c_int64, [vp]),
        "kz_knz_index": (c.c_int32, [u8p, c.c_int64, vp, vp, vp, vp, vp, i64p, i64p, c.c_int32]),
        "kz_set_timing": (None, [vp, c.c_int32]),
        "kz_get_stage_count": (c.c_int32, [vp]),
        "kz_get_stage_ms": (c.c_float, [vp, c.c_int32]),
        "kz_get_stage_alg_bytes": (c.c_int64, [vp, c.c_int32]),
        "kz_reset_timing": (None, [vp]),
        "kz_set_kernel_timing": (None, [vp, c.c_int32]),
        "kz_get_kernel_count": (c.c_int32, []),
        "kz_get_kernel_name": (c.c_char_p, [c.c_int32]),
        "kz_get_kernel_ms": (c.c_double, [vp, c.c_int32]),
        "kz_get_kernel_max_ms": (c.c_double, [vp, c.c_int32]),
        "kz_get_kernel_launches": (c.c_int64, [vp, c.c_int32]),
        "kz_reset_kernel_timing": (None, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


ABI_SYMBOLS = ["kz_abi_version", "kz_ctx_create", "kz_ctx_destroy", "kz_last_error", "kz_ctx_stream", "kz_pin_to_device_numa", "kz_host_cpus", "kz_host_share", "kz_ctx_set_checksum",
               "kz_ctx_set_data_type", "kz_ctx_get_data_type", "kz_ctx_reset", "kz_ctx_set_skip_blocks", "kz_ctx_set_block_size", "kz_ctx_reload_switches", "kz_host_stage_blocks", "kz_ctx_set_entropy",
               "kz_transform_forward", "kz_transform_inverse", "kz_transform_max_encoded_len", "kz_host_stage_forward", "kz_host_stage_inverse",
               "kz_entropy_encode", "kz_entropy_decode", "kz_encode_blocks", "kz_decode_blocks",
               "kz_max_block_stream_bytes", "kz_submit_encode_blocks", "kz_submit_decode_blocks", "kz_wait", "kz_poll", "kz_compress", "kz_decompress", "kz_compress_bound", "kz_transform_type",
               "kz_knz_assemble", "kz_knz_index", "kz_knz_writer_open", "kz_knz_writer_add", "kz_knz_writer_close",
               "kz_set_timing", "kz_get_stage_count", "kz_get_stage_ms", "kz_get_stage_alg_bytes", "kz_reset_timing",
               "kz_set_kernel_timing", "kz_get_kernel_count", "kz_get_kernel_name", "kz_get_kernel_ms", "kz_get_kernel_max_ms",
               "kz_get_kernel_launches", "kz_reset_kernel_timing"]


# The reference's compression levels (K/app/BlockCompressor.java:537-573, getTransformAndCodec) as "transforms&entropy".
# Levels 0-3, 5 and 6 consist of stages built here (TEXT and UTF run as host stages in front of the GPU chain);
# 4 and 7-9 need ROLZ / EXE / LZP / CM / TPAQ.
LEVELS = {0: "NONE&NONE", 1: "LZX&NONE", 2: "DNA+LZ&HUFFMAN", 3: "TEXT+UTF+PACK+MM+LZX&HUFFMAN", 4: "TEXT+UTF+EXE+PACK+MM+ROLZ&NONE",
          5: "TEXT+UTF+BWT+RANK+ZRLT&ANS0", 6: "TEXT+UTF+BWT+SRT+ZRLT&FPAQ", 7: "LZP+TEXT+UTF+BWT+LZP&CM",
          8: "EXE+RLT+TEXT+UTF+DNA&TPAQ", 9: "EXE+RLT+TEXT+UTF+DNA&TPAQX"}


def level_chain(level, allow_partial=False):
    """(transform string, entropy name) of a reference level.  Raises for a level with stages that are not built, unless
    allow_partial drops the leading TEXT+UTF pair (the result is then NOT level-exact on blocks those stages accept)."""
    t, e = LEVELS[int(level)].split("&")
    names = t.split("+")
    if allow_partial and names[:2] == ["TEXT", "UTF"]:
        names = names[2:]
    missing = [n for n in names if n not in TRANSFORM_IDS] + ([e] if e not in ENTROPY_IDS else [])
    if missing:
        raise KanziError(3, "level %d needs %s, not built here" % (level, "/".join(missing)))       # ERR_INVALID_CODEC
    return "+".join(names), e


def host_stage_forward(name, data, entropy="NONE", block_size=4 * 1024 * 1024, data_type=0, cap=None):
    """TEXT / UTF forward on the host (no context, no GPU) -> (applied, bytes, data type after the call)"""
    L = load_library()
    a = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
    t = TRANSFORM_IDS[name.upper()]
    if cap is None:
        cap = int(L.kz_transform_max_encoded_len(t, len(a)))
    out = np.zeros(max(cap, 1) + 64, dtype=np.uint8)
    p, dt = ctypes.c_int32(0), ctypes.c_int32(int(data_type))
    r = L.kz_host_stage_forward(t, ENTROPY_IDS[entropy.upper()], int(block_size), ctypes.addressof(dt), a.ctypes.data if len(a) else out.ctypes.data, len(a),
                                out.ctypes.data, cap, ctypes.addressof(p))
    if r < 0:
        raise KanziError(-r, "host_stage_forward")
    return bool(r), out[:p.value].tobytes(), int(dt.value)


def host_stage_inverse(name, data, cap, block_size=4 * 1024 * 1024):
    """TEXT / UTF inverse on the host -> (ok, bytes)"""
    L = load_library()
    a = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
    out = np.zeros(max(cap, 1) + 64, dtype=np.uint8)
    p = ctypes.c_int32(0)
    r = L.kz_host_stage_inverse(TRANSFORM_IDS[name.upper()], int(block_size), a.ctypes.data if len(a) else out.ctypes.data, len(a), out.ctypes.data, cap, ctypes.addressof(p))
    if r < 0:
        raise KanziError(-r, "host_stage_inverse")
    return bool(r), out[:p.value].tobytes()


def transform_type(names):
    """'BWT+RANK+ZRLT' or a list of names/ids -> 48-bit id word (TransformFactory.java:132-158)."""
    if isinstance(names, str):
        names = [s for s in names.split("+") if s]
    ids = [TRANSFORM_IDS[x.upper()] if isinstance(x, str) else int(x) for x in names]
    if not 1 <= len(ids) <= 8:
        raise ValueError("Only 1 to 8 transforms allowed")          # Sequence.java:42-43
    t = 0
    for i in range(8):
        t = (t << 6) | (ids[i] if i < len(ids) else 0)
    return t


def _ptr(a):
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return int(a)


_LIVE = weakref.WeakSet()


def reload_switches():
    """every live Context reads the KZ_* environment switches again (they are read once, at creation)"""
    for c in list(_LIVE):
        c.reload_switches()


class Context:
    """One HIP stream + scratch arena on one GPU (kz_ctx). Not thread-safe, like a reference codec instance."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.h = self.lib.kz_ctx_create(int(device))
        if not self.h:
            raise RuntimeError("kz_ctx_create(%d) failed: no usable HIP device (the HIP path has no CPU fallback)" % device)
        _LIVE.add(self)

    def close(self):
        if getattr(self, "h", None):
            self.lib.kz_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reload_switches(self):
        """The KZ_* environment switches are read once, when the context is created: read them again (tests, A/B tools)."""
        if getattr(self, "h", None):
            self.lib.kz_ctx_reload_switches(self.h)

    def error(self):
        return self.lib.kz_last_error(self.h).decode("utf-8", "replace")

    def check(self, rc):
        if rc < 0:
            raise KanziError(-rc, self.error())
        return rc

    def set_checksum(self, bits):
        """0, 32 or 64: block checksum kind (the reference's -x32 / -x64)."""
        self.check(self.lib.kz_ctx_set_checksum(self.h, int(bits)))

    def set_skip_blocks(self, on):
        """The context map's "skipBlocks" entry (CLI --skip): incompressible-looking blocks become copy blocks."""
        self.check(self.lib.kz_ctx_set_skip_blocks(self.h, 1 if on else 0))

    def set_block_size(self, block_size):
        """The context map's "blockSize" entry (the stream's block size): TEXT sizes its hash map by it."""
        self.check(self.lib.kz_ctx_set_block_size(self.h, int(block_size)))

    def set_entropy(self, entropy):
        """The context map's "entropy" entry for single-transform calls: TEXT is TextCodec2 for NONE/ANS0/HUFFMAN/RANGE, else TextCodec1."""
        self.check(self.lib.kz_ctx_set_entropy(self.h, ENTROPY_IDS[entropy.upper()] if isinstance(entropy, str) else int(entropy)))

    def set_data_type(self, data_type):
        """The context map's "dataType" entry (a DATA_TYPES name or value) that the next transform instance will see."""
        dt = DATA_TYPES[data_type.upper()] if isinstance(data_type, str) else int(data_type)
        self.check(self.lib.kz_ctx_set_data_type(self.h, dt))

    def reset(self):
        """Every set_* value back to a fresh context's."""
        self.check(self.lib.kz_ctx_reset(self.h))

    def get_data_type(self):
        """What the last forward transform left in the context's "dataType" entry (DATA_TYPES value)."""
        return int(self.lib.kz_ctx_get_data_type(self.h))

    @property
    def stream(self):
        return self.lib.kz_ctx_stream(self.h)

    # ---- instrumentation ----
    def set_timing(self, on):
        self.lib.kz_set_timing(self.h, 1 if on else 0)

    def reset_timing(self):
        self.lib.kz_reset_timing(self.h)

    def set_kernel_timing(self, on):
        self.lib.kz_set_kernel_timing(self.h, 1 if on else 0)

    def reset_kernel_timing(self):
        self.lib.kz_reset_kernel_timing(self.h)

    def kernel_times(self):
        """{kernel name: {"ms": summed launch durations, "launches": n, "max_ms": longest launch}} since the last reset."""
        out = {}
        for i in range(self.lib.kz_get_kernel_count()):
            n = int(self.lib.kz_get_kernel_launches(self.h, i))
            if n:
                out[self.lib.kz_get_kernel_name(i).decode()] = {"ms": float(self.lib.kz_get_kernel_ms(self.h, i)), "launches": n,
                                                                "max_ms": float(self.lib.kz_get_kernel_max_ms(self.h, i))}
        return out

    def stage_times(self):
        out = {}
        for i, nm in enumerate(STAGE_NAMES):
            ms = float(self.lib.kz_get_stage_ms(self.h, i))
            if ms > 0:
                out[nm] = {"ms": ms, "alg_bytes": int(self.lib.kz_get_stage_alg_bytes(self.h, i))}
        return out


class SliceByteArray:
    """K/SliceByteArray.java:35-37."""

    def __init__(self, array=None, length=None, index=0):
        self.array = array if array is not None else np.zeros(0, dtype=np.uint8)
        self.length = len(self.array) if length is None else length
        self.index = index


class _Transform:
    """ByteTransform mirror. forward/inverse return True (applied) or False (declined), advancing
    src.index / dst.index exactly like the reference codecs do on success."""
    TYPE = None

    def __init__(self, ctx):
        self.ctx = ctx

    def getMaxEncodedLength(self, n):
        return int(self.ctx.lib.kz_transform_max_encoded_len(self.TYPE, n))

    def _run(self, fn, src, dst):
        if src.length == 0:
            return True
        s = np.ascontiguousarray(src.array[src.index:src.index + src.length], dtype=np.uint8)
        cap = len(dst.array) - dst.index if fn is self.ctx.lib.kz_transform_inverse else dst.length - dst.index
        if cap <= 0:
            return False
        out = np.empty(cap, dtype=np.uint8)
        produced = ctypes.c_int32(0)
        rc = fn(self.ctx.h, self.TYPE, s.ctypes.data, src.length, out.ctypes.data, cap, ctypes.addressof(produced))
        self.ctx.check(rc)
        if rc == 0:
            return False
        dst.array[dst.index:dst.index + produced.value] = out[:produced.value]
        src.index += src.length
        dst.index += produced.value
        return True

    def forward(self, src, dst):
        return self._run(self.ctx.lib.kz_transform_forward, src, dst)

    def inverse(self, src, dst):
        return self._run(self.ctx.lib.kz_transform_inverse, src, dst)


class BWTBlockCodec(_Transform):
    TYPE = BWT_TYPE            # K/transform/BWTBlockCodec.java


class TextCodec(_Transform):
    """K/transform/TextCodec.java (host stage).  Like the reference's constructor it takes the variant from the context map:
    entropy (TextCodec2 for NONE / ANS0 / HUFFMAN / RANGE, else TextCodec1) and blockSize (hash map size)."""
    TYPE = TEXT_TYPE

    def __init__(self, ctx, entropy="NONE", blockSize=4 * 1024 * 1024):
        super().__init__(ctx)
        ctx.set_entropy(entropy)
        ctx.set_block_size(blockSize)


class UTFCodec(_Transform):
    TYPE = UTF_TYPE            # K/transform/UTFCodec.java (host stage)


class ZRLT(_Transform):
    TYPE = ZRLT_TYPE           # K/transform/ZRLT.java


class SRT(_Transform):
    TYPE = SRT_TYPE            # K/transform/SRT.java


class FSDCodec(_Transform):
    TYPE = MM_TYPE             # K/transform/FSDCodec.java (transform name "MM")


class AliasCodec(_Transform):
    """K/transform/AliasCodec.java: transform "PACK"; onlyDNA=True is transform "DNA" (TransformFactory.java:341-343)."""

    def __init__(self, ctx, onlyDNA=False):
        super().__init__(ctx)
        self.TYPE = DNA_TYPE if onlyDNA else PACK_TYPE


class LZCodec(_Transform):
    """K/transform/LZCodec.java (LZXCodec): lz=LZ_TYPE or LZX_TYPE."""

    def __init__(self, ctx, lz=LZ_TYPE):
        super().__init__(ctx)
        self.TYPE = lz


class SBRT(_Transform):
    MODE_MTF, MODE_RANK, MODE_TIMESTAMP = 1, 2, 3   # K/transform/SBRT.java:35-37

    def __init__(self, ctx, mode=2):
        super().__init__(ctx)
        if mode not in (1, 2):
            raise ValueError("Invalid mode parameter")
        self.TYPE = RANK_TYPE if mode == 2 else MTFT_TYPE


class _EntropyEncoder:
    TYPE = None

    def __init__(self, ctx):
        self.ctx = ctx
        self.bits = []         # list of (bytes, nbits) appended, like the OutputBitStream given at construction

    def encode(self, block, blkptr, count):
        """EntropyEncoder.encode: returns count on success. Output bit string appended to self.bits."""
        s = np.ascontiguousarray(block[blkptr:blkptr + count], dtype=np.uint8) if count else np.zeros(1, dtype=np.uint8)
        cap = int(self.ctx.lib.kz_max_block_stream_bytes(count))
        out = np.zeros(cap, dtype=np.uint8)
        nbits = self.ctx.lib.kz_entropy_encode(self.ctx.h, self.TYPE, s.ctypes.data, count, out.ctypes.data, cap)
        self.ctx.check(nbits)
        self.bits.append((out[:(nbits + 7) // 8].tobytes(), int(nbits)))
        return count

    def dispose(self):
        pass


class _EntropyDecoder:
    TYPE = None

    def __init__(self, ctx, data, nbits):
        self.ctx, self.data, self.nbits = ctx, data, nbits

    def decode(self, block, blkptr, count):
        if count == 0:
            return 0
        s = np.frombuffer(bytes(self.data) + b"\0" * 16, dtype=np.uint8)
        out = np.empty(count, dtype=np.uint8)
        used = ctypes.c_int64(0)
        rc = self.ctx.lib.kz_entropy_decode(self.ctx.h, self.TYPE, s.ctypes.data, self.nbits, out.ctypes.data, count, ctypes.addressof(used))
        if rc < 0:
            return -1
        self.bits_consumed = used.value          # the Java adapter advances the shared InputBitStream by this many bits
        block[blkptr:blkptr + count] = out
        return count


class ANSRangeEncoder(_EntropyEncoder):
    TYPE = E_ANS0              # K/entropy/ANSRangeEncoder.java (order 0)


class ANSRangeDecoder(_EntropyDecoder):
    TYPE = E_ANS0


class HuffmanEncoder(_EntropyEncoder):
    TYPE = E_HUFFMAN           # K/entropy/HuffmanEncoder.java


class HuffmanDecoder(_EntropyDecoder):
    TYPE = E_HUFFMAN


class FPAQEncoder(_EntropyEncoder):
    TYPE = E_FPAQ              # K/entropy/FPAQEncoder.java (encode + dispose)


class FPAQDecoder(_EntropyDecoder):
    TYPE = E_FPAQ


class NullEntropyEncoder(_EntropyEncoder):
    TYPE = E_NONE


class NullEntropyDecoder(_EntropyDecoder):
    TYPE = E_NONE


def max_block_stream_bytes(n):
    return int(load_library().kz_max_block_stream_bytes(int(n)))


def encode_blocks(ctx, transform, entropy, inp, in_stride, lengths, out, out_stride, mem=MEM_HOST):
    """Fused batched encode (kz_encode_blocks). `inp`/`out`: numpy uint8 arrays (host) or integer device
    pointers (mem=MEM_DEVICE). Returns a ctypes array of BlockResult."""
    tt = transform if isinstance(transform, int) else transform_type(transform)
    et = entropy if isinstance(entropy, int) else ENTROPY_IDS[entropy.upper()]
    lens = np.ascontiguousarray(lengths, dtype=np.int32)
    res = (BlockResult * len(lens))()
    rc = ctx.lib.kz_encode_blocks(ctx.h, tt, et, _ptr(inp), int(in_stride), lens.ctypes.data, len(lens),
                                  _ptr(out), int(out_stride), ctypes.addressof(res), mem)
    ctx.check(rc)
    return res


def decode_blocks(ctx, transform, entropy, block_size, inp, in_stride, bit_lengths, out, out_stride, mem=MEM_HOST):
    tt = transform if isinstance(transform, int) else transform_type(transform)
    et = entropy if isinstance(entropy, int) else ENTROPY_IDS[entropy.upper()]
    bl = np.ascontiguousarray(bit_lengths, dtype=np.int64)
    res = (BlockResult * len(bl))()
    rc = ctx.lib.kz_decode_blocks(ctx.h, tt, et, int(block_size), _ptr(inp), int(in_stride), bl.ctypes.data, len(bl),
                                  _ptr(out), int(out_stride), ctypes.addressof(res), mem)
    ctx.check(rc)
    return res


class Job:
    """A batch queued with submit_encode_blocks / submit_decode_blocks; wait() -> the BlockResult array.  Keeps the arrays the
    C side still points at alive."""

    def __init__(self, ctx, job, res, keep):
        self.ctx, self.job, self.res, self._keep = ctx, job, res, keep

    def done(self):
        return bool(self.ctx.lib.kz_poll(self.ctx.h, self.job))

    def wait(self):
        self.ctx.check(self.ctx.lib.kz_wait(self.ctx.h, self.job))
        return self.res


def submit_encode_blocks(ctx, transform, entropy, inp, in_stride, lengths, out, out_stride, mem=MEM_HOST):
    """kz_submit_encode_blocks: returns a Job at once; the call runs on the context's worker thread."""
    tt = transform if isinstance(transform, int) else transform_type(transform)
    et = entropy if isinstance(entropy, int) else ENTROPY_IDS[entropy.upper()]
    lens = np.ascontiguousarray(lengths, dtype=np.int32)
    res = (BlockResult * len(lens))()
    job = ctx.lib.kz_submit_encode_blocks(ctx.h, tt, et, _ptr(inp), int(in_stride), lens.ctypes.data, len(lens),
                                          _ptr(out), int(out_stride), ctypes.addressof(res), mem)
    ctx.check(min(job, 0))
    return Job(ctx, job, res, (lens, inp, out))


def submit_decode_blocks(ctx, transform, entropy, block_size, inp, in_stride, bit_lengths, out, out_stride, mem=MEM_HOST):
    tt = transform if isinstance(transform, int) else transform_type(transform)
    et = entropy if isinstance(entropy, int) else ENTROPY_IDS[entropy.upper()]
    bl = np.ascontiguousarray(bit_lengths, dtype=np.int64)
    res = (BlockResult * len(bl))()
    job = ctx.lib.kz_submit_decode_blocks(ctx.h, tt, et, int(block_size), _ptr(inp), int(in_stride), bl.ctypes.data, len(bl),
                                          _ptr(out), int(out_stride), ctypes.addressof(res), mem)
    ctx.check(min(job, 0))
    return Job(ctx, job, res, (bl, inp, out))


class CompressedOutputStream:
    """K/io/CompressedOutputStream.java: write() bytes, close() -> .knz bytes in self.output.
    ctx keys mirror the reference's Map (transform, entropy, blockSize)."""

    def __init__(self, ctx, transform="BWT+RANK+ZRLT", entropy="ANS0", blockSize=4 * 1024 * 1024, checksum=0, skipBlocks=False):
        if blockSize > 1024 * 1024 * 1024:
            raise ValueError("The block size must be at most 1 GB")              # CompressedOutputStream.java:165-174
        if blockSize < 1024:
            raise ValueError("The block size must be at least 1024")
        if blockSize & 15:
            raise ValueError("The block size must be a multiple of 16")
        self.ctx = ctx
        self.tt = transform_type(transform)
        self.et = ENTROPY_IDS[entropy.upper()]
        self.blockSize = blockSize
        self.checksum = checksum
        self.skipBlocks = skipBlocks
        self._chunks = []
        self.closed = False
        self.output = None

    def write(self, data):
        if self.closed:
            raise KanziError(12, "Stream closed")                                # ERR_WRITE_FILE
        self._chunks.append(bytes(data))

    def close(self):
        if self.closed:
            return
        self.closed = True
        src = np.frombuffer(b"".join(self._chunks), dtype=np.uint8)
        n = len(src)
        cap = int(self.ctx.lib.kz_compress_bound(n, self.blockSize))
        dst = np.empty(cap, dtype=np.uint8)
        sp = src.ctypes.data if n else dst.ctypes.data
        self.ctx.set_checksum(self.checksum)
        self.ctx.set_skip_blocks(self.skipBlocks)
        try:
            rc = self.ctx.lib.kz_compress(self.ctx.h, self.tt, self.et, self.blockSize, sp, n, dst.ctypes.data, cap)
        finally:
            self.ctx.set_checksum(0)
            self.ctx.set_skip_blocks(False)
        self.ctx.check(rc)
        self.output = dst[:rc].tobytes()


class CompressedInputStream:
    """K/io/CompressedInputStream.java: read() the whole decoded stream."""

    def __init__(self, ctx, data):
        self.ctx, self.data = ctx, bytes(data)

    def read(self, max_size=None):
        src = np.frombuffer(self.data + b"\0" * 16, dtype=np.uint8)
        cap = max_size if max_size is not None else self._declared_size()
        dst = np.empty(max(cap, 1), dtype=np.uint8)
        rc = self.ctx.lib.kz_decompress(self.ctx.h, src.ctypes.data, len(self.data), dst.ctypes.data, cap)
        self.ctx.check(rc)
        return dst[:rc].tobytes()

    def _declared_size(self):
        # The size field of the stream header is informative and untrusted (CompressedOutputStream.java:236-290): what the
        # stream can hold is bounded by its own block walk -- every block decodes to at most blockSize bytes.
        try:
            idx = knz_index(self.data)
            return max(len(idx["blocks"]) * idx["blockSize"], 1)
        except KanziError:
            pass
        # damaged stream: count the length prefixes that can still be walked (kz_decompress reports the fault itself)
        d, nbits = self.data, len(self.data) * 8
        if len(d) < 20:
            return 1 << 20
        v = int.from_bytes(d[:20], "big")
        block_size = ((v >> (160 - 119)) & ((1 << 28) - 1)) << 4
        szmask = (v >> (160 - 121)) & 3
        pos, nb = 121 + 16 * szmask + 15 + 24, 0

        def get(p, k):
            w = int.from_bytes(d[p // 8:(p + k + 7) // 8 + 1].ljust((k + 15) // 8 + 1, b"\0"), "big")
            tot = ((k + 15) // 8 + 1) * 8
            return (w >> (tot - (p % 8) - k)) & ((1 << k) - 1)
        while pos + 8 <= nbits and nb < (1 << 20):
            lr = get(pos, 5) + 3
            rd = get(pos + 5, lr)
            if rd == 0:
                break
            nb += 1
            pos += 5 + lr + rd
        return max((nb + 1) * max(min(block_size, 1 << 30), 1024), 1 << 20)


def knz_assemble(transform, entropy, block_size, input_size, streams, bits, checksum=0):
    """Host-only: build a .knz from per-block private streams (list of bytes) in block-id order.
    checksum = 0 / 32 / 64: what the block streams were coded with (Context.set_checksum)."""
    L = load_library()
    tt = transform if isinstance(transform, int) else transform_type(transform)
    et = entropy if isinstance(entropy, int) else ENTROPY_IDS[entropy.upper()]
    nb = len(streams)
    stride = max([len(s) for s in streams] + [8]) + 8
    buf = np.zeros(nb * stride + 8, dtype=np.uint8)
    for i, s in enumerate(streams):
        buf[i * stride:i * stride + len(s)] = np.frombuffer(s, dtype=np.uint8)
    b = np.ascontiguousarray(bits, dtype=np.int64)
    cap = int(sum((int(x) + 7) // 8 + 8 for x in bits)) + 64
    dst = np.zeros(cap, dtype=np.uint8)
    rc = L.kz_knz_assemble(tt, et, int(block_size), int(input_size), int(checksum), buf.ctypes.data, stride, b.ctypes.data, nb, dst.ctypes.data, cap)
    if rc < 0:
        raise KanziError(-rc, "knz_assemble")
    return dst[:rc].tobytes()


class KnzWriter:
    """Host-only streaming form of knz_assemble (kz_knz_writer_*): add() the block streams in block-id order as they arrive, close()
    -> the .knz bytes.  capacity: an upper bound of the stream's size (kz_compress_bound of the input is one)."""

    def __init__(self, transform, entropy, block_size, input_size, capacity, checksum=0):
        self.L = load_library()
        tt = transform if isinstance(transform, int) else transform_type(transform)
        et = entropy if isinstance(entropy, int) else ENTROPY_IDS[entropy.upper()]
        self.dst = np.zeros(int(capacity) + 64, dtype=np.uint8)
        self.w = self.L.kz_knz_writer_open(tt, et, int(block_size), int(input_size), int(checksum), self.dst.ctypes.data, len(self.dst))
        if not self.w:
            raise KanziError(18, "knz writer")

    def add(self, stream, bits):
        a = np.frombuffer(bytes(stream) + b"\0\0", dtype=np.uint8)
        rc = self.L.kz_knz_writer_add(self.w, a.ctypes.data, int(bits))
        if rc < 0:
            raise KanziError(-rc, "knz writer")

    def close(self):
        n = self.L.kz_knz_writer_close(self.w)
        self.w = None
        if n < 0:
            raise KanziError(-n, "knz writer")
        return self.dst[:n].tobytes()


def knz_index(data):
    """Host-only: -> dict(transform, entropy, blockSize, inputSize, blocks=[(bitOffset, bits), ...])."""
    L = load_library()
    src = np.frombuffer(bytes(data) + b"\0" * 16, dtype=np.uint8)
    tt, et, bs, isz, chk = ctypes.c_uint64(0), ctypes.c_uint32(0), ctypes.c_int32(0), ctypes.c_int64(0), ctypes.c_int32(0)
    cap = max(16, len(data) // 8 + 16)
    off = np.zeros(cap, dtype=np.int64)
    bits = np.zeros(cap, dtype=np.int64)
    nb = L.kz_knz_index(src.ctypes.data, len(data), ctypes.addressof(tt), ctypes.addressof(et), ctypes.addressof(bs),
                        ctypes.addressof(isz), ctypes.addressof(chk), off.ctypes.data, bits.ctypes.data, cap)
    if nb < 0:
        raise KanziError(-nb, "knz_index")
    return {"transform": tt.value, "entropy": et.value, "blockSize": bs.value, "inputSize": isz.value, "checksum": chk.value,
            "blocks": [(int(off[i]), int(bits[i])) for i in range(nb)]}


def usable_cpus():
    """CPUs this process can actually use: its affinity mask cut down to the cgroup CPU quota, if any (kz_host.hip)."""
    return int(load_library().kz_host_cpus())


def pin_host_threads_to_gpu(device, world=2):
    """One process per GPU, `world` of them on one host: keep this process's host threads (TEXT / UTF stages, bit assembly,
    staging copies) on the CPUs of the GPU's NUMA node, and its thread pool inside 1/world of the host's CPU quota.  A single process (world == 1) keeps the whole machine.  Returns the
    number of CPUs pinned to (0: nothing changed)."""
    if world <= 1:
        return 0
    lib = load_library()
    lib.kz_host_share(int(world))                 # and 1/world of the cgroup CPU quota, if there is one
    return max(0, int(lib.kz_pin_to_device_numa(int(device))))


def shard_blocks(n_blocks, world, rank):
    """Round-robin block partition over GPUs: block i -> rank i mod world."""
    return list(range(rank, n_blocks, world))


def extract_bits(data, bit_off, nbits):
    """Host helper: the nbits-long bit string starting at bit_off of `data`, left aligned (MSB first)."""
    v = int.from_bytes(data[bit_off // 8:(bit_off + nbits + 7) // 8 + 1], "big")
    total = ((bit_off + nbits + 7) // 8 + 1 - bit_off // 8) * 8
    have = min(total, len(data) * 8 - (bit_off // 8) * 8)
    v <<= (total - have)
    v = (v >> (total - (bit_off % 8) - nbits)) & ((1 << nbits) - 1)
    pad = (-nbits) % 8
    return (v << pad).to_bytes((nbits + 7) // 8, "big")
