"""ctypes binding of the CPU ORACLE (oracle/libkzo.so). Test infrastructure only: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by kanzi_amd/."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ODIR, "libkzo.so")

T = {"NONE": 0, "BWT": 1, "LZ": 3, "ZRLT": 6, "MTFT": 7, "RANK": 8, "TEXT": 10, "SRT": 13, "MM": 15, "LZX": 16, "UTF": 17, "PACK": 18, "DNA": 19}
# Global.DataType as numbered in oracle/kzo.h
DT = {"UNDEFINED": 0, "DNA": 1, "SMALL_ALPHABET": 2, "TEXT": 3, "MULTIMEDIA": 4, "EXE": 5, "NUMERIC": 6, "BASE64": 7, "BIN": 8, "UTF8": 9}
E = {"NONE": 0, "HUFFMAN": 1, "FPAQ": 2, "ANS0": 5}


def build():
    srcs = [os.path.join(ODIR, f) for f in os.listdir(ODIR) if f.endswith(".c") or f.endswith(".h")]
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs):
        return LIB
    subprocess.check_call(["make", "-C", ODIR, "-s", "libkzo.so"])
    return LIB


_L = None


def lib():
    global _L
    if _L is None:
        build()
        L = ctypes.CDLL(LIB)
        c = ctypes
        L.kzo_compress.restype = c.c_int64
        L.kzo_compress.argtypes = [c.c_uint64, c.c_int, c.c_int, c.c_void_p, c.c_int64, c.c_void_p, c.c_int64, c.c_int]
        L.kzo_compress_x.restype = c.c_int64
        L.kzo_compress_x.argtypes = [c.c_uint64, c.c_int, c.c_int, c.c_int, c.c_void_p, c.c_int64, c.c_void_p, c.c_int64, c.c_int]
        L.kzo_encode_block_x.restype = c.c_int64
        L.kzo_encode_block_x.argtypes = [c.c_uint64, c.c_int, c.c_int, c.c_void_p, c.c_int, c.c_void_p, c.c_size_t, c.c_void_p, c.c_void_p]
        L.kzo_encode_block_y.restype = c.c_int64
        L.kzo_encode_block_y.argtypes = [c.c_uint64, c.c_int, c.c_int, c.c_int, c.c_void_p, c.c_int, c.c_void_p, c.c_size_t, c.c_void_p, c.c_void_p]
        L.kzo_set_transform_ctx.argtypes = [c.c_int, c.c_int]
        L.kzo_set_transform_ctx.restype = None
        L.kzo_xxhash32.restype = c.c_uint32
        L.kzo_xxhash32.argtypes = [c.c_void_p, c.c_int, c.c_uint32]
        L.kzo_xxhash64.restype = c.c_uint64
        L.kzo_xxhash64.argtypes = [c.c_void_p, c.c_int, c.c_uint64]
        L.kzo_decompress.restype = c.c_int64
        L.kzo_decompress.argtypes = [c.c_void_p, c.c_int64, c.c_void_p, c.c_int64, c.c_int]
        L.kzo_encode_block.restype = c.c_int64
        L.kzo_encode_block.argtypes = [c.c_uint64, c.c_int, c.c_void_p, c.c_int, c.c_void_p, c.c_size_t, c.c_void_p, c.c_void_p]
        L.kzo_decode_block.restype = c.c_int
        L.kzo_decode_block.argtypes = [c.c_uint64, c.c_int, c.c_int, c.c_void_p, c.c_int64, c.c_void_p, c.c_int]
        L.kzo_transform_forward.argtypes = [c.c_int, c.c_void_p, c.c_void_p, c.c_int, c.c_void_p, c.c_int, c.c_void_p]
        L.kzo_transform_inverse.argtypes = [c.c_int, c.c_void_p, c.c_int, c.c_void_p, c.c_int, c.c_void_p]
        L.kzo_transform_max_encoded_len.argtypes = [c.c_int, c.c_int]
        L.kzo_transform_type.restype = c.c_uint64
        L.kzo_transform_type.argtypes = [c.c_void_p, c.c_int]
        L.kzo_entropy_encode.argtypes = [c.c_int, c.c_void_p, c.c_void_p, c.c_int]
        L.kzo_entropy_decode.argtypes = [c.c_int, c.c_void_p, c.c_void_p, c.c_int]
        L.kzo_obs_init.argtypes = [c.c_void_p, c.c_size_t]
        L.kzo_obs_free.argtypes = [c.c_void_p]
        L.kzo_ibs_init.argtypes = [c.c_void_p, c.c_void_p, c.c_uint64]
        L.kzo_stream_header.argtypes = [c.c_uint64, c.c_int, c.c_int, c.c_int, c.c_int64, c.c_void_p]
        L.kzo_bwt_forward_raw.argtypes = [c.c_void_p, c.c_int, c.c_void_p, c.c_void_p]
        L.kzo_suffix_array.argtypes = [c.c_void_p, c.c_void_p, c.c_int]
        L.kzo_jrandom_init.argtypes = [c.c_void_p, c.c_int64]
        L.kzo_jrandom_next_int.argtypes = [c.c_void_p, c.c_int32]
        L.kzo_jrandom_next_int.restype = c.c_int32
        L.kzo_write_varint.argtypes = [c.c_void_p, c.c_uint32]
        L.kzo_encode_alphabet.argtypes = [c.c_void_p, c.c_void_p, c.c_int]
        _L = L
    return _L


class _Obs(ctypes.Structure):
    _fields_ = [("buf", ctypes.c_void_p), ("cap", ctypes.c_size_t), ("nbits", ctypes.c_uint64),
                ("owns", ctypes.c_int), ("overflow", ctypes.c_int)]


class _Ibs(ctypes.Structure):
    _fields_ = [("buf", ctypes.c_void_p), ("nbits", ctypes.c_uint64), ("pos", ctypes.c_uint64), ("error", ctypes.c_int)]


def _u8(data):
    return np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)


def ttype(names):
    if isinstance(names, str):
        names = [s for s in names.split("+") if s]
    ids = (ctypes.c_int * len(names))(*[T[x.upper()] for x in names])
    return int(lib().kzo_transform_type(ids, len(names)))


class TransformThrows(RuntimeError):
    """the reference's transform would throw (LZ: more tokens than its fixed token buffer holds): the block fails with ERR_PROCESS_BLOCK"""


def set_transform_ctx(entropy="NONE", block_size=4 * 1024 * 1024):
    """the context entries "entropy" and "blockSize" the next single-transform calls of this thread see (TEXT reads them)"""
    lib().kzo_set_transform_ctx(E[entropy.upper()], int(block_size))


def transform_forward(name, data, cap=None, data_type=None):
    """-> (applied, bytes); with data_type (a DT value: the block's "dataType" context entry) -> (applied, bytes,
    data type after the call)"""
    a = _u8(data)
    t = T[name.upper()]
    if cap is None:
        cap = lib().kzo_transform_max_encoded_len(t, len(a))
    out = np.zeros(max(cap, 1) + 64, dtype=np.uint8)
    p = ctypes.c_int(0)
    dt = ctypes.c_int(0 if data_type is None else int(data_type))
    ok = lib().kzo_transform_forward(t, None if data_type is None else ctypes.addressof(dt), a.ctypes.data if len(a) else out.ctypes.data, len(a), out.ctypes.data, cap, ctypes.byref(p))
    if ok < 0:
        raise TransformThrows(name)
    if data_type is not None:
        return bool(ok), out[:p.value].tobytes(), int(dt.value)
    return bool(ok), out[:p.value].tobytes()


def transform_inverse(name, data, cap):
    a = _u8(data)
    out = np.zeros(max(cap, 1) + 64, dtype=np.uint8)
    p = ctypes.c_int(0)
    ok = lib().kzo_transform_inverse(T[name.upper()], a.ctypes.data if len(a) else out.ctypes.data, len(a), out.ctypes.data, cap, ctypes.byref(p))
    return bool(ok), out[:p.value].tobytes()


def entropy_encode(name, data):
    """-> (bytes, nbits)"""
    a = _u8(data)
    s = _Obs()
    lib().kzo_obs_init(ctypes.byref(s), len(a) * 2 + 4096)
    r = lib().kzo_entropy_encode(E[name.upper()], ctypes.byref(s), a.ctypes.data if len(a) else None, len(a))
    if r != len(a):
        lib().kzo_obs_free(ctypes.byref(s))
        raise RuntimeError("oracle entropy encode failed")
    nb = int(s.nbits)
    out = ctypes.string_at(s.buf, (nb + 7) // 8)
    lib().kzo_obs_free(ctypes.byref(s))
    return out, nb


def entropy_decode(name, data, nbits, count):
    a = _u8(bytes(data) + b"\0" * 16)
    s = _Ibs()
    lib().kzo_ibs_init(ctypes.byref(s), a.ctypes.data, nbits)
    out = np.zeros(max(count, 1), dtype=np.uint8)
    r = lib().kzo_entropy_decode(E[name.upper()], ctypes.byref(s), out.ctypes.data, count)
    if s.error:                                  # ran past the end of the block's bits: the Java bitstream throws
        r = -1
    return r, out[:count].tobytes(), int(s.pos)


def encode_block(chain, entropy, data, checksum=0, block_size=4 * 1024 * 1024):
    """-> (stream bytes, W bits, skipFlags, postLen); checksum 0 / 32 / 64; block_size = the stream's (TEXT sizes its hash map by it)"""
    a = _u8(data)
    cap = len(a) + len(a) // 8 + 2048
    out = np.zeros(cap, dtype=np.uint8)
    sf = ctypes.c_uint8(0)
    pl = ctypes.c_int(0)
    w = lib().kzo_encode_block_y(ttype(chain), E[entropy.upper()], {0: 0, 32: 1, 64: 2}[checksum], int(block_size), a.ctypes.data, len(a), out.ctypes.data, cap, ctypes.byref(sf), ctypes.byref(pl))
    if w == -13:
        raise OracleError(13)
    if w < 0:
        raise RuntimeError("oracle encode_block failed")
    return out[:(w + 7) // 8].tobytes(), int(w), sf.value, pl.value


def decode_block(chain, entropy, block_size, stream, nbits, cap):
    a = _u8(bytes(stream) + b"\0" * 16)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    r = lib().kzo_decode_block(ttype(chain), E[entropy.upper()], block_size, a.ctypes.data, nbits, out.ctypes.data, cap)
    return r, out[:max(r, 0)].tobytes()


def compress(chain, entropy, block_size, data, jobs=1, checksum=0, skip_blocks=False):
    a = _u8(data)
    cap = 2 * len(a) + 65536                   # 1 KiB blocks of random bytes through SRT grow by 28 %: header of 256 frequencies per block
    out = np.zeros(cap, dtype=np.uint8)
    r = lib().kzo_compress_x(ttype(chain), E[entropy.upper()], block_size, {0: 0, 32: 1, 64: 2}[checksum] | (0x100 if skip_blocks else 0), a.ctypes.data if len(a) else out.ctypes.data, len(a), out.ctypes.data, cap, jobs)
    if r == -13:
        raise OracleError(13)
    if r < 0:
        raise RuntimeError("oracle compress failed %d" % r)
    return out[:r].tobytes()


class OracleError(RuntimeError):
    """kzo_decompress failure; code = the K/Error.java value the reference reader would throw."""

    def __init__(self, code):
        super().__init__("oracle decompress failed: Error code %d" % code)
        self.code = code


def decompress(data, cap, jobs=1):
    a = _u8(bytes(data) + b"\0" * 16)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    r = lib().kzo_decompress(a.ctypes.data, len(data), out.ctypes.data, cap, jobs)
    if r < 0:
        raise OracleError(int(-r))
    return out[:r].tobytes()


def xxhash32(data, seed=0x4B414E5A):
    a = _u8(data)
    return int(lib().kzo_xxhash32(a.ctypes.data if len(a) else None, len(a), seed))


def xxhash64(data, seed=0x4B414E5A):
    a = _u8(data)
    return int(lib().kzo_xxhash64(a.ctypes.data if len(a) else None, len(a), seed))


class JavaRandom:
    """java.util.Random for the reference's test-input generators."""

    def __init__(self, seed):
        self.s = (ctypes.c_uint64 * 1)()
        lib().kzo_jrandom_init(self.s, seed)

    def next_int(self, bound):
        return int(lib().kzo_jrandom_next_int(self.s, bound))
