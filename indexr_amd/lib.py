"""ctypes binding of libixrgpu.so (C ABI: include/ixr_gpu.h, include/ixr_synth.h).

There is no fallback: if the shared library is missing, importing this module raises; if no gfx950 device is usable the
decode entry points return IXR_ERR_NO_DEVICE / IXR_ERR_HIP and the wrappers raise IxrError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IXR_LIB") or os.path.join(_HERE, "libixrgpu.so")  # IXR_LIB: A/B builds of the same library

INT, LONG, FLOAT, DOUBLE, STRING = 1, 2, 3, 4, 5
ENC_RAW, ENC_DELTA, ENC_SIMPLE_DICT, ENC_DICT_COMPRESS = 0, 2, 4, 6
ERR_BUFFER_OVERFLOW, ERR_DATA_CORRUPTED, ERR_DATA_IMPROPER, ERR_BUG = -1, -2, -3, -100
ERR_ILLEGAL_ENCODE_TYPE, ERR_UNSUPPORTED_TYPE = -101, -102
ERR_INVALID_ARGUMENT, ERR_NO_DEVICE, ERR_HIP, ERR_OUT_OF_MEMORY = -1000, -1001, -1002, -1003
PACK_MAX_ROWS = 65536
IN_SLACK = 16
STATIC_MAX_PACKS = 64
STATIC_SCRATCH_BYTES = 32 << 20

(COL_I_SEQ, COL_I_RAND, COL_I_CAT, COL_I_RUNS, COL_L_TS, COL_L_CARD, COL_L_RAND, COL_F_CARD, COL_D_CARD, COL_S_CAT,
 COL_S_URL, COL_S_UNIQ, COL_S_EMPTY) = range(13)
COL_NAMES = ["i_seq", "i_rand", "i_cat", "i_runs", "l_ts", "l_card", "l_rand", "f_card", "d_card", "s_cat", "s_url",
             "s_uniq", "s_empty"]

if not os.path.exists(LIB_PATH):
    raise ImportError("%s is missing: build it with `make -C indexr_amd/csrc` (or __graft_entry__.build())" % LIB_PATH)
lib = C.CDLL(LIB_PATH)


class IxrError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = lib.ixr_last_error().decode(errors="replace")
        super().__init__("%s: code %d: %s" % (what, code, msg))


class PackDesc(C.Structure):
    _fields_ = [("in_off", C.c_uint64), ("out_off", C.c_uint64), ("pack_size", C.c_int32), ("data_size", C.c_int32),
                ("obj_count", C.c_int32), ("encode_type", C.c_int32), ("data_type", C.c_int32), ("reserved", C.c_int32)]


class HandbackDesc(C.Structure):
    """ixr_handback_desc: one decoded pack to turn into column vector contents (DefaultPackReader.java:164-241)"""
    _fields_ = [("src_off", C.c_uint64), ("dst_off", C.c_uint64), ("obj_count", C.c_int32), ("data_size", C.c_int32),
                ("sql_type", C.c_int32), ("row_offset", C.c_int32), ("row_count", C.c_int32), ("reserved", C.c_int32)]


SQL_INT, SQL_BIGINT, SQL_FLOAT, SQL_DOUBLE, SQL_VARCHAR, SQL_DATE, SQL_TIME, SQL_DATETIME = range(1, 9)


class FilterDesc(C.Structure):
    """ixr_filter_desc: a predicate on one pack's dictionary ids (ExtIndex_DictBits.java)"""
    _fields_ = [("op", C.c_int32), ("n_values", C.c_int32), ("values_off", C.c_uint64)]


FILTER_EQUAL, FILTER_IN, FILTER_GREATER, FILTER_GREATER_EQUAL, FILTER_BETWEEN, FILTER_LIKE = 1, 2, 3, 4, 5, 6
BITMAP_NONE, BITMAP_ALL, BITMAP_SOME = 0, 1, 2
BITMAP_BYTES = 8192


class ScanFilterDesc(C.Structure):
    """ixr_scan_filter_desc: a predicate on the rows of one decoded pack (ExtIndex_SimpleBits.java)"""
    _fields_ = [("src_off", C.c_uint64), ("values_off", C.c_uint64), ("data_type", C.c_int32), ("obj_count", C.c_int32),
                ("data_size", C.c_int32), ("op", C.c_int32), ("n_values", C.c_int32), ("reserved", C.c_int32)]


class SelectDesc(C.Structure):
    """ixr_select_desc: the rows of one decoded pack that a bitmap picks (DataPack.foreach(BitMap, XxxSetter), DataPack.java:273-306)"""
    _fields_ = [("src_off", C.c_uint64), ("dst_off", C.c_uint64), ("bitmap_off", C.c_uint64), ("data_type", C.c_int32),
                ("obj_count", C.c_int32), ("data_size", C.c_int32), ("all", C.c_int32)]


class EncodeDesc(C.Structure):
    """ixr_encode_desc: one decoded numeric pack to compress on the device (PackCompressor_VLT.compress)"""
    _fields_ = [("src_off", C.c_uint64), ("dst_off", C.c_uint64), ("data_type", C.c_int32), ("is_indexed", C.c_int32),
                ("obj_count", C.c_int32), ("data_size", C.c_int32), ("pack_size", C.c_int32), ("encode_type", C.c_int32),
                ("min_value", C.c_int64), ("max_value", C.c_int64), ("sum", C.c_int64)]


class SynthReq(C.Structure):
    _fields_ = [("kind", C.c_int32), ("pack_id", C.c_uint32), ("rows", C.c_int32), ("is_indexed", C.c_int32)]


_vp, _u64, _i32 = C.c_void_p, C.c_uint64, C.c_int32
_sig = {
    "ixr_init": (_i32, [_i32]),
    "ixr_last_error": (C.c_char_p, []),
    "ixr_device_count": (_i32, []),
    "ixr_debug_switch": (_i32, [C.c_char_p, _i32]),
    "ixr_dev_alloc": (_vp, [_i32, _u64]),
    "ixr_dev_free": (None, [_i32, _vp]),
    "ixr_host_alloc_pinned": (_vp, [_u64]),
    "ixr_host_free_pinned": (None, [_vp]),
    "ixr_memcpy_h2d": (_i32, [_i32, _vp, _vp, _u64]),
    "ixr_memcpy_d2h": (_i32, [_i32, _vp, _vp, _u64]),
    "ixr_dev_synchronize": (_i32, [_i32]),
    "ixr_stream_create": (C.c_void_p, [_i32]),
    "ixr_stream_destroy": (None, [_i32, C.c_void_p]),
    "ixr_stream_synchronize": (_i32, [_i32, C.c_void_p]),
    "ixr_plan_create": (_vp, [_i32, C.POINTER(PackDesc), _i32, _u64]),
    "ixr_plan_destroy": (None, [_vp]),
    "ixr_plan_in_bytes": (_u64, [_vp]),
    "ixr_plan_out_bytes": (_u64, [_vp]),
    "ixr_plan_scratch_bytes": (_u64, [_vp]),
    "ixr_plan_run": (_i32, [_vp, _vp, _u64, _vp, _u64, _vp]),
    "ixr_plan_run_host": (_i32, [_vp, _vp, _u64, _vp, _u64]),
    "ixr_plan_status": (_i32, [_vp, C.POINTER(_i32), _vp]),
    "ixr_plan_totals": (None, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    "ixr_plan_set_profiling": (None, [_vp, _i32]),
    "ixr_plan_set_overlap": (None, [_vp, _i32]),
    "ixr_plan_kernel_times": (_i32, [_vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(_i32), _i32]),
    "ixr_plan_kernel_bytes": (_i32, [_vp, C.POINTER(_u64), _i32]),
    "ixr_plan_run_dict_structs": (_i32, [_vp, _vp, _u64, _vp, _u64, C.POINTER(_u64), _vp]),
    "ixr_decode_batch_static": (_i32, [_i32, C.POINTER(PackDesc), _i32, _vp, _u64, _vp, _u64, _vp, _vp]),
    "ixr_decode_pack": (_i32, [_i32, _i32, _i32, _i32, _vp, _i32, _vp, _i32]),
    "ixr_get_dict_struct": (_i32, [_i32, _i32, _i32, _vp, _i32, _vp, _i32]),
    "ixr_get_data_from_dict_struct": (_i32, [_i32, _i32, _i32, _vp, _i32, _vp, _i32]),
    "ixr_compress_pack": (_i32, [_i32, _i32, _i32, _vp, _i32, _vp, _i32, C.POINTER(_i32)]),
    "ixr_compress_bound": (_i32, [_i32, _i32]),
    "ixr_plan_run_filter": (_i32, [_vp, _vp, _u64, C.POINTER(FilterDesc), _vp, _u64, _vp, _u64, C.POINTER(_i32), _vp]),
    "ixr_scan_filter_run": (_i32, [_i32, C.POINTER(ScanFilterDesc), _i32, _vp, _u64, _vp, _u64, _vp, _u64, C.POINTER(_i32), _vp]),
    "ixr_select_bound": (_u64, [C.POINTER(SelectDesc)]),
    "ixr_select_run": (_i32, [_i32, C.POINTER(SelectDesc), _i32, _vp, _u64, _vp, _u64, _vp, _u64, C.POINTER(_i32), _vp]),
    "ixr_encode_run": (_i32, [_i32, C.POINTER(EncodeDesc), _i32, _vp, _u64, _vp, _u64, _u64, _vp]),
    "ixr_handback_bound": (_u64, [C.POINTER(HandbackDesc)]),
    "ixr_handback_run": (_i32, [_i32, C.POINTER(HandbackDesc), _i32, _vp, _u64, _vp, _u64, C.POINTER(_i32), _vp]),
    "ixr_synth_data_type": (_i32, [_i32]),
    "ixr_synth_column": (_i32, [_i32, _u64, C.c_uint32, _i32, _vp, _i32]),
    "ixr_synth_build": (_i32, [C.POINTER(SynthReq), _i32, _u64, _i32, C.POINTER(PackDesc), C.POINTER(C.POINTER(C.c_uint8)),
                               C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    "ixr_synth_free": (None, [C.POINTER(C.c_uint8)]),
    "ixr_checksum": (_u64, [_vp, _u64]),
    "ixr_plan_checksums": (_i32, [_vp, _vp, _u64, C.POINTER(_u64), _vp]),
}
for _name, (_res, _args) in _sig.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library does not export what include/*.h declares
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTED_SYMBOLS = sorted(_sig)


def debug_switch(name, value):
    """ixr_debug_switch: A/B switches of the library ("split_str", "slice", "wave_walk_from"); they apply to plans created afterwards"""
    rc = lib.ixr_debug_switch(name.encode(), int(value))
    if rc:
        raise IxrError(rc, "ixr_debug_switch(%s)" % name)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def width_of(data_type):
    return {INT: 4, FLOAT: 4, LONG: 8, DOUBLE: 8}.get(data_type, 0)


# ---------------------------------------------------------------------------------------------------- host write path
def compress_pack(data, data_type, obj_count, is_indexed=False):
    """PackCompressor_VLT.compress: decoded pack bytes -> (pack bytes, encode_type)."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    bound = lib.ixr_compress_bound(obj_count, len(data))
    out = np.empty(bound, dtype=np.uint8)
    et = _i32(0)
    sz = lib.ixr_compress_pack(data_type, int(is_indexed), obj_count, _ptr(data), len(data), _ptr(out), bound, C.byref(et))
    if sz < 0:
        raise IxrError(sz, "ixr_compress_pack")
    return out[:sz].copy(), et.value


def synth_column(kind, seed, pack_id, rows):
    cap = 4 * (rows + 1) + 96 * rows
    out = np.empty(cap, dtype=np.uint8)
    sz = lib.ixr_synth_column(kind, seed, pack_id, rows, _ptr(out), cap)
    if sz < 0:
        raise IxrError(sz, "ixr_synth_column")
    return out[:sz].copy()


def synth_build(reqs, seed, threads=8, want_checksums=True):
    """reqs: list of (kind, pack_id, rows, is_indexed) -> (descs ctypes array, arena np.uint8 (owned copy), out_bytes, checksums)"""
    n = len(reqs)
    arr = (SynthReq * n)(*[SynthReq(*r) for r in reqs])
    descs = (PackDesc * n)()
    arena = C.POINTER(C.c_uint8)()
    nbytes, out_bytes = _u64(0), _u64(0)
    sums = (_u64 * n)() if want_checksums else None
    rc = lib.ixr_synth_build(arr, n, seed, threads, descs, C.byref(arena), C.byref(nbytes), C.byref(out_bytes), sums)
    if rc < 0:
        raise IxrError(rc, "ixr_synth_build")
    try:
        host = np.ctypeslib.as_array(arena, shape=(nbytes.value,)).copy()
    finally:
        lib.ixr_synth_free(arena)
    return descs, host, out_bytes.value, (np.frombuffer(sums, dtype=np.uint64).copy() if want_checksums else None)


def checksum(buf):
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    return int(lib.ixr_checksum(_ptr(buf), len(buf)))


# ---------------------------------------------------------------------------------------------------- device side
class DeviceBuffer:
    def __init__(self, device, nbytes):
        self.device, self.nbytes = device, int(nbytes)
        self.ptr = lib.ixr_dev_alloc(device, self.nbytes)
        if not self.ptr:
            raise IxrError(ERR_OUT_OF_MEMORY, "ixr_dev_alloc(%d)" % nbytes)

    def upload(self, host, offset=0):
        host = np.ascontiguousarray(host, dtype=np.uint8)
        assert offset + len(host) <= self.nbytes
        rc = lib.ixr_memcpy_h2d(self.device, C.c_void_p(self.ptr + offset), _ptr(host), len(host))
        if rc:
            raise IxrError(rc, "ixr_memcpy_h2d")

    def download(self, offset=0, nbytes=None):
        nbytes = self.nbytes - offset if nbytes is None else nbytes
        out = np.empty(nbytes, dtype=np.uint8)
        rc = lib.ixr_memcpy_d2h(self.device, _ptr(out), C.c_void_p(self.ptr + offset), nbytes)
        if rc:
            raise IxrError(rc, "ixr_memcpy_d2h")
        return out

    def free(self):
        if self.ptr:
            lib.ixr_dev_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedBuffer:
    """page-locked host memory (ixr_host_alloc_pinned) with a numpy view"""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = lib.ixr_host_alloc_pinned(self.nbytes)
        if not self.ptr:
            raise IxrError(ERR_OUT_OF_MEMORY, "ixr_host_alloc_pinned")
        self.array = np.ctypeslib.as_array((C.c_uint8 * max(self.nbytes, 1)).from_address(self.ptr))[:self.nbytes]

    def free(self):
        if self.ptr:
            self.array = None
            lib.ixr_host_free_pinned(self.ptr)
            self.ptr = None


class Plan:
    """Batch decode plan (ixr_plan_*)."""

    def __init__(self, device, descs, scratch_budget=0):
        self.device = device
        self.n = len(descs)
        if not isinstance(descs, C.Array):
            descs = (PackDesc * self.n)(*descs)
        self.descs = descs
        self.handle = lib.ixr_plan_create(device, descs, self.n, scratch_budget)
        if not self.handle:
            raise IxrError(ERR_HIP, "ixr_plan_create")
        self.in_bytes = lib.ixr_plan_in_bytes(self.handle)
        self.out_bytes = lib.ixr_plan_out_bytes(self.handle)
        self.scratch_bytes = lib.ixr_plan_scratch_bytes(self.handle)
        a, b, c = _u64(0), _u64(0), _u64(0)
        lib.ixr_plan_totals(self.handle, C.byref(a), C.byref(b), C.byref(c))
        self.total_in, self.total_out, self.total_rows = a.value, b.value, c.value

    def run(self, d_in, in_bytes, d_out, out_bytes, stream=None):
        rc = lib.ixr_plan_run(self.handle, d_in, in_bytes, d_out, out_bytes, stream)
        if rc:
            raise IxrError(rc, "ixr_plan_run")

    def run_host(self, h_in, in_bytes, h_out, out_bytes):
        """ixr_plan_run_host: arenas in host memory (addresses), copies overlapped with the decode"""
        rc = lib.ixr_plan_run_host(self.handle, h_in, in_bytes, h_out, out_bytes)
        if rc:
            raise IxrError(rc, "ixr_plan_run_host")

    def run_dict_structs(self, d_in, in_bytes, d_out, out_bytes, ds_off, stream=None):
        arr = (_u64 * self.n)(*[int(x) for x in ds_off])
        rc = lib.ixr_plan_run_dict_structs(self.handle, d_in, in_bytes, d_out, out_bytes, arr, stream)
        if rc:
            raise IxrError(rc, "ixr_plan_run_dict_structs")

    def status(self, stream=None):
        st = (_i32 * self.n)()
        bad = lib.ixr_plan_status(self.handle, st, stream)
        if bad < 0:
            raise IxrError(bad, "ixr_plan_status")
        return np.frombuffer(st, dtype=np.int32).copy(), bad

    def checksums(self, d_out, out_bytes, stream=None):
        sums = (_u64 * self.n)()
        rc = lib.ixr_plan_checksums(self.handle, d_out, out_bytes, sums, stream)
        if rc:
            raise IxrError(rc, "ixr_plan_checksums")
        return np.frombuffer(sums, dtype=np.uint64).copy()

    def set_profiling(self, on):
        lib.ixr_plan_set_profiling(self.handle, int(on))

    def set_overlap(self, on):
        lib.ixr_plan_set_overlap(self.handle, int(on))

    def kernel_times(self):
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        launches = (_i32 * cap)()
        k = lib.ixr_plan_kernel_times(self.handle, names, ms, launches, cap)
        alg = (_u64 * cap)()
        lib.ixr_plan_kernel_bytes(self.handle, alg, cap)
        return [(names[i].decode(), float(ms[i]), int(launches[i]), int(alg[i])) for i in range(k)]

    def destroy(self):
        if self.handle:
            lib.ixr_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def decode_pack(pack, data_type, encode_type, obj_count, data_size, device=0):
    """ixr_decode_pack: -> (decoded bytes np.uint8[data_size], consumed-or-error)"""
    pack = np.ascontiguousarray(pack, dtype=np.uint8)
    out = np.zeros(max(data_size, 1), dtype=np.uint8)
    r = lib.ixr_decode_pack(device, data_type, encode_type, obj_count, _ptr(pack), len(pack), _ptr(out), data_size)
    return out[:data_size], r


def get_dict_struct(pack, data_type, obj_count, device=0):
    pack = np.ascontiguousarray(pack, dtype=np.uint8)
    size = int(pack[:4].view("<i4")[0]) if len(pack) >= 4 else 0
    out = np.zeros(max(size, 8), dtype=np.uint8)
    r = lib.ixr_get_dict_struct(device, data_type, obj_count, _ptr(pack), len(pack), _ptr(out), max(size, 0))
    return out[:max(size, 0)], r


def get_data_from_dict_struct(ds, data_type, obj_count, data_size, device=0):
    ds = np.ascontiguousarray(ds, dtype=np.uint8)
    out = np.zeros(max(data_size, 1), dtype=np.uint8)
    r = lib.ixr_get_data_from_dict_struct(device, data_type, obj_count, _ptr(ds), len(ds), _ptr(out), data_size)
    return out[:data_size], r


def handback_run(descs, d_decoded, decoded_bytes, d_out, out_bytes, device=0, stream=None):
    """ixr_handback_run: -> (status per descriptor, number of failed descriptors)"""
    n = len(descs)
    arr = (HandbackDesc * n)(*descs)
    st = (_i32 * n)()
    bad = lib.ixr_handback_run(device, arr, n, d_decoded, decoded_bytes, d_out, out_bytes, st, stream)
    if bad < 0:
        raise IxrError(bad, "ixr_handback_run")
    return np.frombuffer(st, dtype=np.int32).copy(), bad


def scan_filter_run(descs, d_decoded, decoded_bytes, values, d_bitmaps, bitmap_bytes, device=0, stream=None):
    """ixr_scan_filter_run: -> (results per pack, number of packs with a negative result)"""
    n = len(descs)
    arr = (ScanFilterDesc * n)(*descs)
    res = (_i32 * n)()
    values = np.ascontiguousarray(values, dtype=np.uint8)
    bad = lib.ixr_scan_filter_run(device, arr, n, d_decoded, decoded_bytes, _ptr(values), len(values), d_bitmaps, bitmap_bytes, res, stream)
    if bad < 0:
        raise IxrError(bad, "ixr_scan_filter_run")
    return np.frombuffer(res, dtype=np.int32).copy(), bad


def select_run(descs, d_decoded, decoded_bytes, d_bitmaps, bitmap_bytes, d_out, out_bytes, device=0, stream=None):
    """ixr_select_run: -> (picked rows per descriptor, number of descriptors with a negative count)"""
    n = len(descs)
    arr = (SelectDesc * n)(*descs)
    cnt = (_i32 * n)()
    bad = lib.ixr_select_run(device, arr, n, d_decoded, decoded_bytes, d_bitmaps, bitmap_bytes, d_out, out_bytes, cnt, stream)
    if bad < 0:
        raise IxrError(bad, "ixr_select_run")
    return np.frombuffer(cnt, dtype=np.int32).copy(), bad


def encode_run(descs, d_data, data_bytes, d_out, out_bytes, device=0, scratch_budget=0, stream=None):
    """ixr_encode_run: fills pack_size / encode_type of every descriptor -> (ctypes array of descriptors, failed packs)"""
    n = len(descs)
    arr = descs if isinstance(descs, C.Array) else (EncodeDesc * n)(*descs)
    bad = lib.ixr_encode_run(device, arr, n, d_data, data_bytes, d_out, out_bytes, scratch_budget, stream)
    if bad < 0:
        raise IxrError(bad, "ixr_encode_run")
    return arr, bad


def filter_keys(keys_per_pack, is_string_per_pack):
    """Packs the keys of every pack's predicate into one value buffer -> (np.uint8 buffer, values_off per pack).
    Numeric keys are the reference's long numValue (ints as they are, float / double keys as the raw bits of the double)."""
    parts, offs, pos = [], [], 0
    for keys, is_str in zip(keys_per_pack, is_string_per_pack):
        offs.append(pos)
        if is_str:
            o = np.zeros(len(keys) + 1, dtype="<i4")
            o[1:] = np.cumsum([len(k) for k in keys])
            b = o.tobytes() + b"".join(keys)
        else:
            b = np.asarray(keys, dtype="<i8").tobytes()
        b += b"\0" * (-len(b) % 8)
        parts.append(b)
        pos += len(b)
    return np.frombuffer(b"".join(parts) + b"\0" * 8, dtype=np.uint8).copy(), offs


def plan_run_filter(plan, d_in, in_bytes, filters, values, d_bitmaps, bitmap_bytes, stream=None):
    """ixr_plan_run_filter: -> (results per pack, number of packs with a negative result)"""
    arr = (FilterDesc * plan.n)(*filters)
    res = (_i32 * plan.n)()
    values = np.ascontiguousarray(values, dtype=np.uint8)
    bad = lib.ixr_plan_run_filter(plan.handle, d_in, in_bytes, arr, _ptr(values), len(values), d_bitmaps, bitmap_bytes, res, stream)
    if bad < 0:
        raise IxrError(bad, "ixr_plan_run_filter")
    return np.frombuffer(res, dtype=np.int32).copy(), bad
