"""Pins the CPU oracle (oracle/ixr_oracle.c).

The reference's tests for this path are round-trip property tests with no golden bytes
(indexr-vlt-segment/src/test/java/io/indexr/vlt/codec/{pack/PackingTest,delta/DeltaCodecTest,rle/RLEPackingHybridCodecTest,
dict/DictCodecTest}.java and indexr-segment/src/test/java/io/indexr/segment/pack/DataPackTest.java).  Here those invariants
are restated and, in addition, every encoded byte string is compared against an independent spec-level codec (tests/pyref.py)
and hand-derived known-answer bytes.
"""
import struct

import numpy as np
import pytest

import oracle_lib as O
import pyref as R
from oracle_lib import _p, lib


def rng(seed):
    return np.random.default_rng(seed)


# ------------------------------------------------------------------------------------------ bit packing
def test_packing_hand_kat():
    # bw=3, values 0..7: bit string (LSB first) 000 100 010 110 001 101 011 111 -> 0x88 0xC6 0xFA
    vals = np.arange(8, dtype="<i4")
    out = np.zeros(3, dtype=np.uint8)
    assert lib.ixo_pack_ints(3, 8, _p(vals), _p(out)) == 3
    assert out.tobytes() == bytes([0x88, 0xC6, 0xFA])
    back = np.zeros(8, dtype="<i4")
    assert lib.ixo_unpack_ints(3, 8, _p(out), _p(back)) == 3
    assert (back == vals).all()
    # bw=1: value i is bit i ; 0b10110001 = 0xB1
    vals = np.array([1, 0, 0, 0, 1, 1, 0, 1], dtype="<i4")
    out = np.zeros(1, dtype=np.uint8)
    lib.ixo_pack_ints(1, 8, _p(vals), _p(out))
    assert out[0] == 0xB1
    # bw=12, two values 0xABC, 0x123 (+6 zero) -> bytes BC 3A 12 then zeros
    vals = np.array([0xABC, 0x123, 0, 0, 0, 0, 0, 0], dtype="<i4")
    out = np.zeros(12, dtype=np.uint8)
    lib.ixo_pack_ints(12, 8, _p(vals), _p(out))
    assert out.tobytes()[:3] == bytes([0xBC, 0x3A, 0x12]) and not out[3:].any()


@pytest.mark.parametrize("bw", list(range(0, 33)))
def test_int_packing_all_widths(bw):
    # PackingTest.java:31-59: widths 1..31, counts 0..99 (tails < 8). Here 0..32 incl. sign bit.
    r = rng(bw)
    for count in list(range(0, 20)) + [31, 32, 33, 63, 64, 65, 99]:
        hi = (1 << bw)
        vals = r.integers(0, hi, size=count, dtype=np.uint64).astype(np.uint32).view("<i4") if bw else np.zeros(count, "<i4")
        nbytes = ((count + 7) // 8) * bw
        out = np.zeros(nbytes + 8, dtype=np.uint8)
        assert lib.ixo_pack_ints(bw, count, _p(vals), _p(out)) == nbytes
        expect = R.pack_bits([int(v) & 0xFFFFFFFF for v in vals], bw)
        assert out[:nbytes].tobytes() == expect
        back = np.full(count + 8, -7, dtype="<i4")
        assert lib.ixo_unpack_ints(bw, count, _p(out), _p(back)) == nbytes
        assert (back[:count] == vals).all() and (back[count:] == -7).all()


@pytest.mark.parametrize("bw", list(range(0, 65)))
def test_long_packing_all_widths(bw):
    # PackingTest.java:61-115: widths 1..63 for longs. Here 0..64.
    r = rng(100 + bw)
    for count in [0, 1, 7, 8, 9, 31, 32, 33, 99]:
        if bw == 0:
            vals = np.zeros(count, "<i8")
        elif bw == 64:
            vals = r.integers(-2**63, 2**63 - 1, size=count, dtype=np.int64).astype("<i8")
        else:
            vals = r.integers(0, 1 << bw, size=count, dtype=np.uint64).view("<i8")
        nbytes = ((count + 7) // 8) * bw
        out = np.zeros(nbytes + 8, dtype=np.uint8)
        assert lib.ixo_pack_longs(bw, count, _p(vals), _p(out)) == nbytes
        expect = R.pack_bits([int(v) & 0xFFFFFFFFFFFFFFFF for v in vals], bw)
        assert out[:nbytes].tobytes() == expect
        back = np.full(count + 8, -7, dtype="<i8")
        assert lib.ixo_unpack_longs(bw, count, _p(out), _p(back)) == nbytes
        assert (back[:count] == vals).all() and (back[count:] == -7).all()


# ------------------------------------------------------------------------------------------ DELTA numbers
def _delta_roundtrip_ints(vals):
    vals = np.ascontiguousarray(vals, dtype="<i4")
    n = len(vals)
    limit = 8 * n + 64
    out = np.zeros(limit, dtype=np.uint8)
    sz = lib.ixo_delta_encode_ints(n, _p(vals), _p(out), limit)
    assert sz >= 0
    assert out[:sz].tobytes() == R.delta_encode([int(v) for v in vals], 32)
    back = np.zeros(n + 1, dtype="<i4")
    consumed = lib.ixo_delta_decode_ints(n, _p(out), sz, _p(back))
    assert consumed == sz
    assert (back[:n] == vals).all()
    pv, pc = R.delta_decode(out[:sz].tobytes(), 0, n, 32)
    assert pc == sz and pv == [int(v) for v in vals]
    return out[:sz]


def _delta_roundtrip_longs(vals):
    vals = np.ascontiguousarray(vals, dtype="<i8")
    n = len(vals)
    limit = 16 * n + 64
    out = np.zeros(limit, dtype=np.uint8)
    sz = lib.ixo_delta_encode_longs(n, _p(vals), _p(out), limit)
    assert sz >= 0
    assert out[:sz].tobytes() == R.delta_encode([int(v) for v in vals], 64)
    back = np.zeros(n + 1, dtype="<i8")
    consumed = lib.ixo_delta_decode_longs(n, _p(out), sz, _p(back))
    assert consumed == sz
    assert (back[:n] == vals).all()
    return out[:sz]


def test_delta_hand_kat():
    # DataPackTest.java:108 literal input. first=1 -> zz 0x02; deltas 0,1,1,2,1,1,1,1,-1120,1444,-333; min=-1120 -> zz 2239
    # = BF 11; rel = 1120,1121,1121,1122,1121,1121,1121,1121,0,2564,787 -> bw=12; 11 values -> 2 groups -> 24 bytes.
    vals = [1, 2, 3, 5, 6, 7, 8, 9, -1111, 333, 0]
    enc = _delta_roundtrip_ints(vals).tobytes()
    rel = [1120, 1121, 1121, 1122, 1121, 1121, 1121, 1121, 0, 2564, 787]
    acc = 0
    for i, v in enumerate(rel):
        acc |= v << (12 * i)
    body = bytes([0x02, 0xBF, 0x11, 12]) + acc.to_bytes(24, "little")
    assert enc == struct.pack("<i", len(body)) + body
    assert len(enc) == 32


@pytest.mark.parametrize("n", [0, 1, 2, 7, 8, 9, 31, 32, 33, 127, 128, 129, 255, 256, 257, 1000, 4097, 65535, 65536])
def test_delta_ints_sizes(n):
    r = rng(n)
    walk = np.cumsum(r.integers(-100000, 10000000, size=n)).astype(np.int64)
    _delta_roundtrip_ints((walk & 0xFFFFFFFF).astype(np.uint32).view("<i4"))


def test_delta_ints_edge_values():
    i32 = np.iinfo(np.int32)
    _delta_roundtrip_ints([0])
    _delta_roundtrip_ints([10, 10, 10, 10])  # DataPackTest.java:122
    _delta_roundtrip_ints([i32.min, i32.max] * 100)  # wrapping deltas, bw 32
    _delta_roundtrip_ints([i32.max, i32.min, 0, -1, 1] * 77)
    _delta_roundtrip_ints(np.zeros(300, dtype=np.int32))  # bw 0 everywhere
    _delta_roundtrip_ints(rng(5).integers(i32.min, i32.max, size=5000, dtype=np.int64))
    # every bit width 0..32 shows up as a mini-block width
    vals = [0]
    for bw in range(33):
        for k in range(32):
            vals.append((vals[-1] + ((1 << bw) - 1 if k == 3 and bw else 0)) & 0xFFFFFFFF)
    _delta_roundtrip_ints(np.array(vals, dtype=np.uint32).view("<i4"))


@pytest.mark.parametrize("n", [0, 1, 31, 32, 33, 128, 129, 1000, 65536])
def test_delta_longs_sizes(n):
    r = rng(1000 + n)
    walk = np.cumsum(r.integers(-10**12, 10**15, size=n)).astype(np.int64)
    _delta_roundtrip_longs(walk)


def test_delta_longs_edge_values():
    i64 = np.iinfo(np.int64)
    _delta_roundtrip_longs([1, i64.max, 3, i64.min, 6, 7, 8, 9, 1111, 333, 0])  # DataPackTest.java:129
    _delta_roundtrip_longs([i64.min, i64.max] * 70)
    _delta_roundtrip_longs(rng(6).integers(i64.min, i64.max, size=3000, dtype=np.int64))
    vals = [0]
    for bw in range(65):
        for k in range(32):
            vals.append((vals[-1] + ((1 << bw) - 1 if k == 5 and bw else 0)) & 0xFFFFFFFFFFFFFFFF)
    _delta_roundtrip_longs(np.array(vals, dtype=np.uint64).view("<i8"))


def test_delta_floats_doubles_are_bit_patterns():
    # DeltaCodec.java:248-265: floats reuse the int routine, doubles the long routine
    f = np.array([1, 2, 3, 5, 6, 7, 8, 9, 1111.4666, 333, 0, np.nan, -0.0, np.inf, -np.inf], dtype="<f4")
    _delta_roundtrip_ints(f.view("<i4"))
    d = np.array([1, 2, 3, 5, np.finfo(np.float64).tiny, 7, np.finfo(np.float64).max, 9, 1111, -333.4444444, 0, np.nan, -0.0],
                 dtype="<f8")
    _delta_roundtrip_longs(d.view("<i8"))


def test_delta_corrupt_inputs():
    vals = np.arange(1000, dtype="<i4")
    enc = _delta_roundtrip_ints(vals).copy()
    back = np.zeros(1000, dtype="<i4")
    assert lib.ixo_delta_decode_ints(1000, _p(enc), len(enc) - 5, _p(back)) == O.DATA_CORRUPTED  # truncated
    bad = enc.copy()
    bad[0:4] = np.array([len(enc) + 50], dtype="<i4").view(np.uint8)  # length beyond buffer
    assert lib.ixo_delta_decode_ints(1000, _p(bad), len(bad), _p(back)) == O.DATA_CORRUPTED
    bad = enc.copy()
    bad[4 + 1 + 1] = 77  # first mini-block width > 32
    assert lib.ixo_delta_decode_ints(1000, _p(bad), len(bad), _p(back)) == O.DATA_CORRUPTED


# ------------------------------------------------------------------------------------------ DELTA strings
def _strings(r, n, prefixes=(b"", b"www.", b"www.example.com/", b"\xe4\xb8\xad\xe6\x96\x87")):
    out = []
    for _ in range(n):
        p = prefixes[int(r.integers(0, len(prefixes)))]
        ln = int(r.integers(0, 24))
        out.append(p + bytes(r.integers(97, 123, size=ln, dtype=np.uint8)))
    return out


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 1000, 20000])
def test_delta_strings(n):
    # DeltaCodecTest.java:115-169: strings with 3 shared prefixes
    r = rng(n)
    strs = sorted(_strings(r, n)) if n % 2 else _strings(r, n)
    ss = O.rows_to_bytes(strs, O.STRING)
    limit = len(ss) * 2 + 1024
    out = np.zeros(limit, dtype=np.uint8)
    sz = lib.ixo_delta_encode_strings(n, _p(ss), _p(out), limit)
    assert sz >= 0
    assert out[:sz].tobytes() == R.delta_encode_strings(strs)
    back = np.zeros(len(ss) + 8, dtype=np.uint8)
    consumed = lib.ixo_delta_decode_strings(n, _p(out), sz, _p(back), len(ss))
    assert consumed == sz
    assert (back[:len(ss)] == ss).all()
    if n:
        ps, pc = R.delta_decode_strings(out[:sz].tobytes(), 0, n)
        assert pc == sz and ps == strs


# ------------------------------------------------------------------------------------------ RLE hybrid
def _rle_roundtrip(ids, bw):
    ids = np.ascontiguousarray(ids, dtype="<i4")
    n = len(ids)
    limit = 8 * n + 64
    out = np.zeros(limit, dtype=np.uint8)
    sz = lib.ixo_rle_encode_ints(bw, n, _p(ids), _p(out), limit)
    assert sz >= 0
    assert out[:sz].tobytes() == R.rle_encode([int(v) for v in ids], bw)
    back = np.zeros(n + 1, dtype="<i4")
    assert lib.ixo_rle_decode_ints(bw, n, _p(out), sz, _p(back)) == sz
    assert (back[:n] == ids).all()
    if n:
        pv, pc = R.rle_decode(out[:sz].tobytes(), 0, n, bw)
        assert pc == sz and pv == [int(v) for v in ids]
    return out[:sz]


def test_rle_hand_kat():
    # 10 x 5 then 3,4 with bw=3: run header 10<<1 = 0x14, value 1 byte 0x05; tail literal: header (2<<1)|1 = 5, packed 3|4<<3 = 0x23, 2 zero bytes
    enc = _rle_roundtrip([5] * 10 + [3, 4], 3).tobytes()
    body = bytes([0x14, 0x05, 0x05, 0x23, 0x00, 0x00])
    assert enc == struct.pack("<i", len(body)) + body
    # a trailing run is never RLE-coded (RLEPackingHybridCodec.java:47-67): 12 x 1, bw=1 -> one literal chunk of 12
    enc = _rle_roundtrip([1] * 12, 1).tobytes()
    body = bytes([(12 << 1) | 1, 0xFF, 0x0F])
    assert enc == struct.pack("<i", len(body)) + body
    # 3-byte fixed-width value (bw 17): byte + short<<8 (VarInteger.java:37-40)
    enc = _rle_roundtrip([0x1ABCD] * 9 + [0], 17).tobytes()
    assert enc[4:4 + 4] == bytes([0x12, 0xCD, 0xAB, 0x01])


@pytest.mark.parametrize("seed", range(6))
def test_rle_random_runs(seed):
    # RLEPackingHybridCodecTest.java:27-69: random runs of length 0..15
    r = rng(seed)
    card = [1, 2, 7, 300, 40000, 65536][seed]
    bw = max(card - 1, 0).bit_length()
    ids = []
    while len(ids) < 20000:
        ids += [int(r.integers(0, card))] * int(r.integers(0, 16))
    for n in [0, 1, 7, 8, 9, 100, len(ids)]:
        _rle_roundtrip(ids[:n], bw)


def test_rle_rejects_negative_and_corrupt():
    ids = np.array([1, -2, 3], dtype="<i4")
    out = np.zeros(64, dtype=np.uint8)
    assert lib.ixo_rle_encode_ints(2, 3, _p(ids), _p(out), 64) == O.DATA_IMPROPER
    enc = _rle_roundtrip([3] * 100 + [1, 2, 3], 2).copy()
    back = np.zeros(200, dtype="<i4")
    assert lib.ixo_rle_decode_ints(2, 102, _p(enc), len(enc), _p(back)) == O.DATA_CORRUPTED  # count mismatch
    assert lib.ixo_rle_decode_ints(2, 104, _p(enc), len(enc), _p(back)) == O.DATA_CORRUPTED
    assert lib.ixo_rle_decode_ints(2, 103, _p(enc), len(enc) - 1, _p(back)) == O.DATA_CORRUPTED  # truncated


# ------------------------------------------------------------------------------------------ dictionaries / packs
def _column(r, data_type, n, card):
    """values with `card` distinct candidates (DictCodecTest.java:41-274 style: k/10 for fp)"""
    k = r.integers(0, card, size=n)
    if data_type == O.INT:
        return (k * 7919 - 1000).astype("<i4")
    if data_type == O.LONG:
        return (k.astype(np.int64) * 1000003 * 65537 - 5).astype("<i8")
    if data_type == O.FLOAT:
        return (k / 10.0).astype("<f4")
    if data_type == O.DOUBLE:
        return (k / 10.0).astype("<f8")
    return [b"str_%d" % (v * 31) if v % 5 else b"" for v in k]


def _bits(rows, data_type):
    if data_type == O.STRING:
        return list(rows)
    it = {O.INT: "<i4", O.LONG: "<i8", O.FLOAT: "<i4", O.DOUBLE: "<i8"}[data_type]
    return [int(v) for v in np.asarray(rows).view(it)]


ALL_TYPES = [O.INT, O.LONG, O.FLOAT, O.DOUBLE, O.STRING]


@pytest.mark.parametrize("data_type", ALL_TYPES)
@pytest.mark.parametrize("card", [1, 2, 10, 63, 64, 65, 1010, 5010])
def test_dict_codecs_roundtrip(data_type, card):
    r = rng(card * 10 + data_type)
    n = 6000 if card < 2000 else 20000
    rows = _column(r, data_type, n, card)
    data = O.rows_to_bytes(rows, data_type)
    # DictCodecTest.java:48-52: ratio 0 => DATA_IMPROPER
    tmp = np.zeros(len(data) + 4 * n + 2048, dtype=np.uint8)
    assert lib.ixo_dict_encode(data_type, 0.0, n, _p(data), _p(tmp), len(tmp)) == O.DATA_IMPROPER
    # Dict (uncompressed DictStruct), ratio 1
    ds_size = lib.ixo_dict_encode(data_type, 1.0, n, _p(data), _p(tmp), len(tmp))
    assert ds_size > 0 and lib.ixo_dict_struct_size(data_type, _p(tmp)) == ds_size
    back, w = O.data_from_dict_struct(tmp[:ds_size], data_type, n, len(data))
    assert w == len(data) and (back == data).all()
    # SimpleDict and DictCompress through the pack-level entry points; consumed == encoded size
    for et in (O.SIMPLE_DICT, O.DICT_COMPRESS, O.DELTA):
        pack = O.compress_pack_as(data, data_type, n, et)
        dec, consumed = O.decompress_pack(pack, data_type, et, n, len(data))
        assert consumed == len(pack), (et, consumed, len(pack))
        assert (dec == data).all()
        pdec, pcons = R.decode_pack(pack, n, data_type, et)
        assert pcons == len(pack) and pdec == data.tobytes()
        if et == O.DICT_COMPRESS:
            assert pack.tobytes() == R.encode_dict_compress(_bits(rows, data_type), data_type)
            ds, c2 = O.get_dict_struct(pack, data_type, n)
            assert c2 == len(pack) and ds.tobytes() == tmp[:ds_size].tobytes()


def test_dict_ratio_threshold_is_float_ceil():
    # PackCompressor_VLT.java:35 uses 0.2f: maxDictEntryCount = (int) ceil(65536 * 0.2f) = 13108
    n = 65536
    for card, expect in [(13108, O.DICT_COMPRESS), (13109, O.DELTA)]:
        vals = (np.arange(n) % card).astype("<i4")
        pack, et = O.compress_pack(vals.view(np.uint8), O.INT, n, is_indexed=False)
        assert et == expect
        dec, consumed = O.decompress_pack(pack, O.INT, et, n, 4 * n)
        assert consumed == len(pack) and (dec.view("<i4") == vals).all()
        pack, et = O.compress_pack(vals.view(np.uint8), O.INT, n, is_indexed=True)
        assert et == O.DICT_COMPRESS


# literal inputs of DataPackTest.java:108-169,196 (expected output = the input)
F32 = np.float32
DATAPACK_CASES = [
    (O.INT, np.array([1, 2, 3, 5, 6, 7, 8, 9, -1111, 333, 0], dtype="<i4")),
    (O.INT, np.array([0], dtype="<i4")),
    (O.INT, np.array([10, 10, 10, 10], dtype="<i4")),
    (O.LONG, np.array([1, 2**63 - 1, 3, -2**63, 6, 7, 8, 9, 1111, 333, 0], dtype="<i8")),
    (O.FLOAT, np.array([1, 2, 3, 5, 6, 7, 8, 9, F32(1111.4666), 333, 0], dtype="<f4")),
    (O.FLOAT, np.array([F32(1.1), F32(1.1)], dtype="<f4")),
    (O.DOUBLE, np.array([1, 2, 3, 5, 4.9e-324, 7, 1.7976931348623157e308, 9, 1111, -333.4444444, 0], dtype="<f8")),
    (O.STRING, [b"qwfwrfwe", b"121343%$#22342342", b"2211", b"sdfdsgkwkek00-\t\t\x05\x03\x01\x08sdaasds", b""]),
    (O.STRING, [b"", b"", b"", b""]),
    (O.STRING, [s.encode() for s in ["121343%$#22342342", "#$%#%242rwef23r23dfsdf", "", "2211",
                                     "sdfdsgkwkek00-\t\t\x05\x03\x01\x08sdaasds", "", "323weewew", "[][][][[]]", "windows", "mac",
                                     "linux", "android", "windows", "mac", "linux", "android", "windows", "mac", "linux",
                                     "android", "qwfwrfwe"]]),
]


@pytest.mark.parametrize("case", range(len(DATAPACK_CASES)))
@pytest.mark.parametrize("is_indexed", [False, True])
def test_datapack_literal_inputs(case, is_indexed):
    data_type, rows = DATAPACK_CASES[case]
    n = len(rows)
    data = O.rows_to_bytes(rows, data_type)
    pack, et = O.compress_pack(data, data_type, n, is_indexed)
    dec, consumed = O.decompress_pack(pack, data_type, et, n, len(data))
    assert consumed == len(pack) and (dec == data).all()
    pdec, pcons = R.decode_pack(pack, n, data_type, et)
    assert pcons == len(pack) and pdec == data.tobytes()
    if et == O.DICT_COMPRESS:
        assert pack.tobytes() == R.encode_dict_compress(_bits(rows, data_type), data_type)


def test_datapack_hand_kat_dict_pack():
    # int {10,10,10,10}: DictStruct = count 1 | 10 | 16 | ids 0,0,0,0 -> 28 bytes; compressed form:
    # size(28) | entryCount 1 | compressed 0 | entry 0A000000 | RLE: len 1 | header (4<<1)|1 = 9 with bw 0 (no payload)
    data = np.array([10, 10, 10, 10], dtype="<i4").view(np.uint8)
    pack, et = O.compress_pack(data, O.INT, 4, False)
    assert et == O.DICT_COMPRESS
    expect = struct.pack("<iiBi", 28, 1, 0, 10) + struct.pack("<i", 1) + bytes([9])
    assert pack.tobytes() == expect
    ds, c = O.get_dict_struct(pack, O.INT, 4)
    assert c == len(pack)
    assert ds.tobytes() == struct.pack("<iii4i", 1, 10, 16, 0, 0, 0, 0)


def test_float_special_values_decode_only():
    # NaN payloads / -0.0 in a hand-made dictionary pack: decode is a pure function of the bytes
    entries = np.array([0x80000000, 0x00000000, 0x7FC00001, 0xFFC00000, 0x7F800000], dtype="<u4")
    ids = [4, 0, 1, 2, 3, 3, 0]
    ds = struct.pack("<i", 5) + entries.tobytes() + struct.pack("<i", 28) + struct.pack("<7i", *ids)
    pack = struct.pack("<iiB", len(ds), 5, 0) + entries.tobytes() + R.rle_encode(ids, 3)
    dec, consumed = O.decompress_pack(np.frombuffer(pack, np.uint8), O.FLOAT, O.DICT_COMPRESS, 7, 28)
    assert consumed == len(pack)
    assert (dec.view("<u4") == entries[ids]).all()


def test_uncompressed_pack_is_identity():
    data = np.arange(100, dtype="<i8").view(np.uint8)
    dec, consumed = O.decompress_pack(data, O.LONG, 0, 100, len(data))
    assert consumed == len(data) and (dec == data).all()


def test_decode_batch_threads():
    import ctypes as C
    r = rng(7)
    packs, descs, rows_all = [], [], []
    in_off = out_off = 0
    for i in range(24):
        dt = ALL_TYPES[i % 5]
        n = int(r.integers(1, 3000))
        rows = _column(r, dt, n, int(r.integers(1, 500)))
        data = O.rows_to_bytes(rows, dt)
        pack, et = O.compress_pack(data, dt, n, bool(i & 1))
        packs.append(pack)
        rows_all.append(data)
        descs.append((in_off, out_off, len(pack), len(data), n, et, dt))
        in_off += (len(pack) + 15) & ~15
        out_off += (len(data) + 15) & ~15
    arena = np.zeros(in_off + 16, dtype=np.uint8)
    for p, d in zip(packs, descs):
        arena[d[0]:d[0] + len(p)] = p
    out = np.zeros(out_off + 16, dtype=np.uint8)
    arr = (O.PackDesc * len(descs))()
    for k, d in enumerate(descs):
        arr[k] = O.PackDesc(d[0], d[1], d[2], d[3], d[4], d[5], d[6], 0)
    for threads in (1, 4):
        out[:] = 0
        assert lib.ixo_decode_batch(arr, len(descs), _p(arena), _p(out), threads) == 0
        for d, data in zip(descs, rows_all):
            assert (out[d[1]:d[1] + d[3]] == data).all()
