"""ctypes binding of oracle/liboracle.so (the CPU oracle; test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.environ.get("IXO_LIB") or os.path.join(ORACLE_DIR, "liboracle.so")  # IXO_LIB: another build (sanitizers)

INT, LONG, FLOAT, DOUBLE, STRING = 1, 2, 3, 4, 5
PLAIN, DELTA, RLE, SIMPLE_DICT, DICT, DICT_COMPRESS = 1, 2, 3, 4, 5, 6
BUFFER_OVERFLOW, DATA_CORRUPTED, DATA_IMPROPER, BUG = -1, -2, -3, -100


def _build_if_needed():
    if not os.path.exists(LIB_PATH) and os.path.exists(os.path.join(ORACLE_DIR, "Makefile")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


_build_if_needed()
lib = C.CDLL(LIB_PATH)

_u8p = C.POINTER(C.c_uint8)


def _p(a):
    """pointer to the bytes of a numpy array / bytes-like"""
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(_u8p)
    if isinstance(a, (bytes, bytearray)):  # the caller keeps `a` alive for the duration of the call
        return np.frombuffer(a, dtype=np.uint8).ctypes.data_as(_u8p) if len(a) else C.cast(None, _u8p)
    raise TypeError(type(a))


for name, res, args in [
    ("ixo_pack_ints", C.c_int, [C.c_int, C.c_int, _u8p, _u8p]),
    ("ixo_unpack_ints", C.c_int, [C.c_int, C.c_int, _u8p, _u8p]),
    ("ixo_pack_longs", C.c_int, [C.c_int, C.c_int, _u8p, _u8p]),
    ("ixo_unpack_longs", C.c_int, [C.c_int, C.c_int, _u8p, _u8p]),
    ("ixo_delta_encode_ints", C.c_int, [C.c_int, _u8p, _u8p, C.c_int]),
    ("ixo_delta_decode_ints", C.c_int, [C.c_int, _u8p, C.c_int, _u8p]),
    ("ixo_delta_encode_longs", C.c_int, [C.c_int, _u8p, _u8p, C.c_int]),
    ("ixo_delta_decode_longs", C.c_int, [C.c_int, _u8p, C.c_int, _u8p]),
    ("ixo_delta_encode_strings", C.c_int, [C.c_int, _u8p, _u8p, C.c_int]),
    ("ixo_delta_decode_strings", C.c_int, [C.c_int, _u8p, C.c_int, _u8p, C.c_int]),
    ("ixo_rle_encode_ints", C.c_int, [C.c_int, C.c_int, _u8p, _u8p, C.c_int]),
    ("ixo_rle_decode_ints", C.c_int, [C.c_int, C.c_int, _u8p, C.c_int, _u8p]),
    ("ixo_dict_encode", C.c_int, [C.c_int, C.c_float, C.c_int, _u8p, _u8p, C.c_int]),
    ("ixo_data_from_dict_struct", C.c_int, [C.c_int, C.c_int, _u8p, C.c_int, _u8p, C.c_int]),
    ("ixo_dict_struct_compress", C.c_int, [C.c_int, C.c_int, _u8p, _u8p, C.c_int]),
    ("ixo_dict_struct_decompress", C.c_int, [C.c_int, C.c_int, _u8p, C.c_int, _u8p, C.c_int]),
    ("ixo_dict_struct_size", C.c_int, [C.c_int, _u8p]),
    ("ixo_simple_dict_encode", C.c_int, [C.c_int, C.c_int, _u8p, _u8p, C.c_int]),
    ("ixo_compress_pack", C.c_int, [C.c_int, C.c_int, C.c_int, _u8p, C.c_int, _u8p, C.c_int, C.POINTER(C.c_int)]),
    ("ixo_compress_pack_as", C.c_int, [C.c_int, C.c_int, C.c_int, _u8p, C.c_int, _u8p, C.c_int]),
    ("ixo_decompress_pack", C.c_int, [C.c_int, C.c_int, C.c_int, _u8p, C.c_int, _u8p, C.c_int]),
    ("ixo_get_dict_struct", C.c_int, [C.c_int, C.c_int, _u8p, C.c_int, _u8p, C.c_int]),
    ("ixo_filter", C.c_int, [C.c_int, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), _u8p, C.POINTER(C.c_uint64)]),
    ("ixo_scan_filter", C.c_int, [C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), _u8p, C.POINTER(C.c_uint64)]),
    ("ixo_like", C.c_int, [_u8p, C.c_int, _u8p, C.c_int]),
    ("ixo_pack_stats", C.c_int, [C.c_int, C.c_int, _u8p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("ixo_select", C.c_int, [C.c_int, C.c_int, _u8p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_int32), _u8p,
                             C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("ixo_handback", C.c_int, [C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int]),
]:
    fn = getattr(lib, name)
    fn.restype = res
    fn.argtypes = args


class PackDesc(C.Structure):
    _fields_ = [("in_off", C.c_uint64), ("out_off", C.c_uint64), ("pack_size", C.c_int32), ("data_size", C.c_int32),
                ("obj_count", C.c_int32), ("encode_type", C.c_int32), ("data_type", C.c_int32), ("status", C.c_int32)]


lib.ixo_decode_batch.restype = C.c_int
lib.ixo_decode_batch.argtypes = [C.POINTER(PackDesc), C.c_int, _u8p, _u8p, C.c_int]


# ---------------------------------------------------------------- convenience wrappers (numpy in / numpy out)
def rows_to_bytes(rows, data_type):
    """rows: numpy array (numbers) or list of bytes (strings) -> decoded-pack bytes (np.uint8)."""
    if data_type == STRING:
        offs = np.zeros(len(rows) + 1, dtype="<i4")
        if len(rows):
            offs[1:] = np.cumsum([len(s) for s in rows])
        return np.concatenate([offs.view(np.uint8), np.frombuffer(b"".join(rows), dtype=np.uint8)])
    dt = {INT: "<i4", LONG: "<i8", FLOAT: "<f4", DOUBLE: "<f8"}[data_type]
    return np.ascontiguousarray(np.asarray(rows, dtype=dt)).view(np.uint8).copy()


def bytes_to_strings(buf, n):
    buf = np.asarray(buf, dtype=np.uint8)
    offs = buf[:4 * (n + 1)].view("<i4")
    body = buf[4 * (n + 1):].tobytes()
    return [body[offs[i]:offs[i + 1]] for i in range(n)]


def compress_pack(data, data_type, n, is_indexed=False):
    """-> (pack bytes np.uint8, encode_type)"""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    limit = len(data) + 4 * n + 4096
    out = np.zeros(limit, dtype=np.uint8)
    et = C.c_int(0)
    sz = lib.ixo_compress_pack(data_type, int(is_indexed), n, _p(data), len(data), _p(out), limit, C.byref(et))
    if sz < 0:
        raise RuntimeError("oracle compress failed: %d" % sz)
    return out[:sz].copy(), et.value


def compress_pack_as(data, data_type, n, encode_type):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    limit = 2 * len(data) + 4 * n + 4096
    out = np.zeros(limit, dtype=np.uint8)
    sz = lib.ixo_compress_pack_as(data_type, encode_type, n, _p(data), len(data), _p(out), limit)
    if sz < 0:
        raise RuntimeError("oracle compress_as failed: %d" % sz)
    return out[:sz].copy()


def decompress_pack(pack, data_type, encode_type, n, data_size):
    """-> (decoded bytes np.uint8, consumed-or-error)"""
    pack = np.ascontiguousarray(pack, dtype=np.uint8)
    out = np.zeros(max(data_size, 1), dtype=np.uint8)
    r = lib.ixo_decompress_pack(data_type, encode_type, n, _p(pack), len(pack), _p(out), data_size)
    return out[:data_size], r


def get_dict_struct(pack, data_type, n):
    pack = np.ascontiguousarray(pack, dtype=np.uint8)
    size = int(pack[:4].view("<i4")[0])
    out = np.zeros(max(size, 8), dtype=np.uint8)
    r = lib.ixo_get_dict_struct(data_type, n, _p(pack), len(pack), _p(out), size)
    return out[:size], r


def data_from_dict_struct(ds, data_type, n, data_size):
    ds = np.ascontiguousarray(ds, dtype=np.uint8)
    out = np.zeros(max(data_size, 1), dtype=np.uint8)
    r = lib.ixo_data_from_dict_struct(data_type, n, _p(ds), len(ds), _p(out), data_size)
    return out[:data_size], r


def handback(sql_type, n, decoded, row_offset, row_count, out_limit=None):
    """DefaultPackReader hand-back of rows [row_offset, row_offset+row_count) -> (bytes np.uint8, bytes-written-or-error)"""
    decoded = np.ascontiguousarray(decoded, dtype=np.uint8)
    if out_limit is None:
        out_limit = len(decoded) + 8 * row_count + 16
    out = np.zeros(max(out_limit, 1), dtype=np.uint8)
    r = lib.ixo_handback(sql_type, n, _p(decoded), len(decoded), row_offset, row_count, _p(out), out_limit)
    return out[:max(r, 0)], r


def filter_pack(pack, data_type, encode_type, n, op, keys):
    """ExtIndex_DictBits on one pack -> (result kind or error, bitmap np.uint64[1024]).  keys: ints (numValue) or bytes."""
    pack = np.ascontiguousarray(pack, dtype=np.uint8)
    bitmap = np.zeros(1024, dtype=np.uint64)
    if data_type == STRING:
        sv = rows_to_bytes(list(keys), STRING)
        r = lib.ixo_filter(data_type, encode_type, n, _p(pack), len(pack), op, len(keys), None, _p(sv),
                           bitmap.ctypes.data_as(C.POINTER(C.c_uint64)))
    else:
        nv = np.asarray(keys, dtype=np.int64)
        r = lib.ixo_filter(data_type, encode_type, n, _p(pack), len(pack), op, len(keys), nv.ctypes.data_as(C.POINTER(C.c_int64)),
                           None, bitmap.ctypes.data_as(C.POINTER(C.c_uint64)))
    return r, bitmap


def scan_filter_pack(data, data_type, n, op, keys):
    """ExtIndex_SimpleBits on one decoded pack -> (0 / 2 or error, bitmap np.uint64[1024]).  keys: ints (numValue) or bytes."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    bitmap = np.zeros(1024, dtype=np.uint64)
    bp = bitmap.ctypes.data_as(C.POINTER(C.c_uint64))
    if keys and isinstance(keys[0], (bytes, bytearray)):
        sv = rows_to_bytes(list(keys), STRING)
        r = lib.ixo_scan_filter(data_type, n, _p(data), len(data), op, len(keys), None, _p(sv), bp)
    else:
        nv = np.asarray(keys, dtype=np.int64)
        r = lib.ixo_scan_filter(data_type, n, _p(data), len(data), op, len(keys), nv.ctypes.data_as(C.POINTER(C.c_int64)), None, bp)
    return r, bitmap


def pack_stats(data, data_type, n):
    """DataPackCreator_VLT's dpn statistics -> (minValue, maxValue, sum) as int64"""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    r = lib.ixo_pack_stats(data_type, n, _p(data), C.byref(a), C.byref(b), C.byref(c))
    if r:
        raise RuntimeError("ixo_pack_stats: %d" % r)
    return a.value, b.value, c.value


def select_pack(data, data_type, n, bitmap=None):
    """DataPack.foreach(BitMap, setter) on one decoded pack; bitmap None = BitMap.ALL.
    -> (count or error, row ids int32[count], values: np array of the pack's type, or list of bytes for strings)"""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    ids = np.zeros(max(n, 1), dtype=np.int32)
    vals = np.zeros(len(data) + 8, dtype=np.uint8)
    so = np.zeros(max(n, 1), dtype=np.int32)
    sl = np.zeros(max(n, 1), dtype=np.int32)
    i32p = C.POINTER(C.c_int32)
    bp = None
    if bitmap is not None:
        bitmap = np.ascontiguousarray(bitmap, dtype=np.uint64)
        assert len(bitmap) == 1024
        bp = bitmap.ctypes.data_as(C.POINTER(C.c_uint64))
    r = lib.ixo_select(data_type, n, _p(data), len(data), bp, ids.ctypes.data_as(i32p), _p(vals), so.ctypes.data_as(i32p),
                       sl.ctypes.data_as(i32p))
    if r < 0:
        return r, None, None
    if data_type == STRING:
        body = vals.tobytes()
        return r, ids[:r].copy(), [body[so[k]:so[k] + sl[k]] for k in range(r)]
    dt = {INT: "<i4", LONG: "<i8", FLOAT: "<f4", DOUBLE: "<f8"}[data_type]
    return r, ids[:r].copy(), vals[:r * np.dtype(dt).itemsize].view(dt).copy()
