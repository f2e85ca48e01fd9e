"""GPU parity tests: the HIP decoder, called through the C ABI (libixrgpu.so), against the CPU oracle on the same inputs.

Bar: bit-exact (integer / byte work; floats and doubles are moved as bit patterns, never computed on).
The inputs restate the reference's own tests: DataPackTest literal inputs, DeltaCodecTest random walks,
RLEPackingHybridCodecTest random runs, DictCodecTest cardinality sweeps over all five types, plus edge cases (tails < 8,
every bit width, wrap-around, empty strings, maximum pack size, corrupted streams).
"""
import os
import struct

import numpy as np
import pytest

import oracle_lib as O
import pyref as R
from gpu_util import Case, check_cases, run_batch
from indexr_amd import lib as L
from test_oracle import ALL_TYPES, DATAPACK_CASES, _column, _strings, rng

pytestmark = pytest.mark.gpu


def oracle_case(rows, data_type, encode_type=None, is_indexed=False, name=""):
    """Encode with the oracle (choice of codec as the reference, or forced), expected output = oracle decode."""
    n = len(rows)
    data = O.rows_to_bytes(rows, data_type)
    if encode_type is None:
        pack, et = O.compress_pack(data, data_type, n, is_indexed)
    else:
        pack, et = O.compress_pack_as(data, data_type, n, encode_type), encode_type
    dec, consumed = O.decompress_pack(pack, data_type, et, n, len(data))
    assert consumed == len(pack) and (dec == data).all()
    return Case(pack, data_type, et, n, len(data), dec, name or "type%d enc%d n%d" % (data_type, et, n))


def test_device_is_gfx950():
    assert L.lib.ixr_device_count() >= 1
    assert L.lib.ixr_init(0) == 0, L.lib.ixr_last_error()


def test_datapack_literal_inputs():
    cases = []
    for data_type, rows in DATAPACK_CASES:
        for is_indexed in (False, True):
            cases.append(oracle_case(rows, data_type, None, is_indexed))
        for et in (O.DELTA, O.SIMPLE_DICT, O.DICT_COMPRESS, 0):
            cases.append(oracle_case(rows, data_type, et))
    check_cases(cases)


def test_delta_ints_sizes_and_widths():
    cases = []
    i32 = np.iinfo(np.int32)
    for n in [1, 2, 3, 4, 5, 7, 8, 9, 31, 32, 33, 127, 128, 129, 255, 256, 257, 1023, 1024, 1025, 4097, 65535, 65536]:
        r = rng(n)
        walk = np.cumsum(r.integers(-100000, 10000000, size=n)).astype(np.int64)
        cases.append(oracle_case((walk & 0xFFFFFFFF).astype(np.uint32).view("<i4"), O.INT, O.DELTA))
    cases.append(oracle_case(np.array([i32.min, i32.max] * 500, dtype="<i4"), O.INT, O.DELTA, name="wrap"))
    cases.append(oracle_case(np.zeros(3000, dtype="<i4"), O.INT, O.DELTA, name="bw0"))
    cases.append(oracle_case(rng(5).integers(i32.min, i32.max, size=65536, dtype=np.int64).astype("<i4"), O.INT, O.DELTA, name="rand32"))
    vals = [0]
    for bw in range(33):
        for k in range(32):
            vals.append((vals[-1] + ((1 << bw) - 1 if k == 3 and bw else 0)) & 0xFFFFFFFF)
    cases.append(oracle_case(np.array(vals, dtype=np.uint32).view("<i4"), O.INT, O.DELTA, name="every width"))
    f = np.array([1, 2, 3, 5, 6, 7, 8, 9, 1111.4666, 333, 0, np.nan, -0.0, np.inf, -np.inf] * 300, dtype="<f4")
    cases.append(oracle_case(f, O.FLOAT, O.DELTA, name="float bits"))
    check_cases(cases)


def test_delta_longs_sizes_and_widths():
    cases = []
    i64 = np.iinfo(np.int64)
    for n in [1, 2, 5, 31, 32, 33, 127, 128, 129, 1000, 1024, 1025, 65535, 65536]:
        r = rng(1000 + n)
        cases.append(oracle_case(np.cumsum(r.integers(-10**12, 10**15, size=n)).astype("<i8"), O.LONG, O.DELTA))
    cases.append(oracle_case(np.array([i64.min, i64.max] * 300, dtype="<i8"), O.LONG, O.DELTA, name="wrap"))
    cases.append(oracle_case(rng(6).integers(i64.min, i64.max, size=65536, dtype=np.int64).astype("<i8"), O.LONG, O.DELTA, name="rand64"))
    vals = [0]
    for bw in range(65):
        for k in range(32):
            vals.append((vals[-1] + ((1 << bw) - 1 if k == 5 and bw else 0)) & 0xFFFFFFFFFFFFFFFF)
    cases.append(oracle_case(np.array(vals, dtype=np.uint64).view("<i8"), O.LONG, O.DELTA, name="every width"))
    d = np.array([1, 2, 3, 5, 4.9e-324, 7, 1.7976931348623157e308, 9, 1111, -333.4444444, 0, np.nan, -0.0] * 200, dtype="<f8")
    cases.append(oracle_case(d, O.DOUBLE, O.DELTA, name="double bits"))
    check_cases(cases)


@pytest.mark.parametrize("data_type", ALL_TYPES)
def test_dict_codecs_cardinality_sweep(data_type):
    # DictCodecTest.java:41-274: all types x {SimpleDict, DictCompress}, cardinality sweep, 65536 values
    cases = []
    for card in [1, 2, 3, 10, 63, 64, 65, 255, 256, 257, 1010, 5010, 20010, 65536]:
        r = rng(card * 10 + data_type)
        n = 65536 if card in (1, 1010, 65536) else int(r.integers(1000, 30000))
        rows = _column(r, data_type, n, card)
        for et in (O.DICT_COMPRESS, O.SIMPLE_DICT):
            cases.append(oracle_case(rows, data_type, et, name="type%d enc%d card%d n%d" % (data_type, et, card, n)))
    check_cases(cases)


def test_rle_runs():
    # RLEPackingHybridCodecTest.java:27-69 : random runs of length 0..15, plus long runs crossing tiles
    cases = []
    for seed, card in enumerate([1, 2, 7, 300, 40000]):
        r = rng(seed)
        vals = []
        while len(vals) < 65536:
            vals += [int(r.integers(0, card))] * int(r.integers(0, 16))
        for n in [1, 7, 8, 9, 1023, 1024, 1025, 65536]:
            cases.append(oracle_case(np.array(vals[:n], dtype="<i4") * 3 - 7, O.INT, O.DICT_COMPRESS, name="runs card%d n%d" % (card, n)))
    r = rng(99)
    vals = []
    while len(vals) < 65536:
        vals += [int(r.integers(0, 5))] * int(r.integers(1, 5000))
    cases.append(oracle_case(np.array(vals[:65536], dtype="<i8") * 1000003, O.LONG, O.DICT_COMPRESS, name="long runs"))
    cases.append(oracle_case(np.full(65536, 7.25, dtype="<f8"), O.DOUBLE, O.DICT_COMPRESS, name="one value"))
    check_cases(cases)


def test_strings():
    cases = []
    r = rng(1)
    for n in [1, 2, 63, 64, 65, 1000, 65536]:
        cases.append(oracle_case(_strings(r, n), O.STRING, O.DICT_COMPRESS, name="str dict n%d" % n))
        cases.append(oracle_case(_strings(r, n), O.STRING, O.SIMPLE_DICT, name="str simple n%d" % n))
        cases.append(oracle_case(_strings(r, n), O.STRING, O.DELTA, name="str delta n%d" % n))
    cases.append(oracle_case([b""] * 5000, O.STRING, name="all empty"))
    cases.append(oracle_case([b"x" * 300, b"x" * 299 + b"y", b"", b"x" * 65535, b"abc"] * 11, O.STRING, O.DICT_COMPRESS, name="long strings"))
    cases.append(oracle_case([b"x" * 70000][0:0] + [bytes([i % 251]) * (i % 400) for i in range(3000)], O.STRING, O.DICT_COMPRESS, name="ragged"))
    cases.append(oracle_case([bytes([i % 251]) * (i % 400) for i in range(3000)], O.STRING, O.DELTA, name="ragged delta"))
    uniq = [b"%08d" % int(v) for v in rng(3).integers(0, 10**8, size=65536)]
    cases.append(oracle_case(uniq, O.STRING, O.DICT_COMPRESS, name="unique strings"))
    cases.append(oracle_case(sorted(uniq), O.STRING, O.DELTA, name="sorted unique delta"))
    check_cases(cases)


def test_synthetic_columns_all_kinds_with_small_scratch_budget():
    reqs = []
    for kind in range(13):
        for pack_id, rows in ((0, 65536), (1, 65536), (2, 16960)):
            reqs.append((kind, pack_id, rows, 0))
    descs, arena, out_bytes, sums = L.synth_build(reqs, 0xC0FFEE, threads=8)
    for budget in (0, 24 << 20):  # second run forces several scratch groups
        plan = L.Plan(0, descs, budget)
        d_in = L.DeviceBuffer(0, len(arena))
        d_out = L.DeviceBuffer(0, out_bytes + 16)
        d_in.upload(arena)
        plan.run(d_in.ptr, len(arena), d_out.ptr, out_bytes + 16)
        status, bad = plan.status()
        assert bad == 0, status
        got = plan.checksums(d_out.ptr, out_bytes + 16)
        assert (got == sums).all()
        host = d_out.download()
        for (kind, pack_id, rows, _), d in zip(reqs, descs):
            data = L.synth_column(kind, 0xC0FFEE, pack_id, rows)
            assert host[d.out_off:d.out_off + d.data_size].tobytes() == data.tobytes(), (L.COL_NAMES[kind], pack_id)
            dec, consumed = O.decompress_pack(arena[d.in_off:d.in_off + d.pack_size], d.data_type, d.encode_type, rows, d.data_size)
            assert consumed == d.pack_size and dec.tobytes() == data.tobytes()
        plan.destroy(); d_in.free(); d_out.free()


def test_single_pack_entry_points():
    r = rng(11)
    for data_type in ALL_TYPES:
        rows = _column(r, data_type, 5000, 300)
        data = O.rows_to_bytes(rows, data_type)
        pack, et = O.compress_pack(data, data_type, 5000, True)
        assert et == O.DICT_COMPRESS
        out, consumed = L.decode_pack(pack, data_type, et, 5000, len(data))
        assert consumed == len(pack), L.lib.ixr_last_error()
        assert out.tobytes() == data.tobytes()
        # getDictStruct == oracle's DictStruct, then getDataFromDictStruct == rows
        ds_o, c_o = O.get_dict_struct(pack, data_type, 5000)
        ds_g, c_g = L.get_dict_struct(pack, data_type, 5000)
        assert c_g == c_o == len(pack), L.lib.ixr_last_error()
        assert ds_g.tobytes() == ds_o.tobytes()
        rows_g, w = L.get_data_from_dict_struct(ds_g, data_type, 5000, len(data))
        assert w == len(data), L.lib.ixr_last_error()
        assert rows_g.tobytes() == data.tobytes()
    # valCount == 0 => 0 bytes consumed (every reference codec: `if (valCount == 0) return 0`)
    out, consumed = L.decode_pack(np.zeros(4, np.uint8), L.INT, L.ENC_DELTA, 0, 0)
    assert consumed == 0
    # illegal encode type
    out, consumed = L.decode_pack(np.zeros(64, np.uint8), L.INT, 3, 10, 40)
    assert consumed == L.ERR_ILLEGAL_ENCODE_TYPE


def test_dict_structs_batch():
    r = rng(12)
    cases = []
    for data_type in ALL_TYPES:
        for card, n in ((1, 100), (40, 3000), (70, 65536), (3000, 20000)):
            cases.append(oracle_case(_column(r, data_type, n, card), data_type, O.DICT_COMPRESS))
    cases.append(oracle_case(np.arange(1000, dtype="<i4"), O.INT, O.DELTA))  # skipped by the DictStruct run
    from gpu_util import build_arena, align16
    descs, arena, _ = build_arena(cases)
    ds_off, off = [], 0
    for c in cases:
        ds_off.append(off)
        off += align16(8 + c.data_size + 4 * c.n)
    plan = L.Plan(0, descs)
    d_in = L.DeviceBuffer(0, len(arena)); d_out = L.DeviceBuffer(0, off + 16)
    d_in.upload(arena)
    plan.run_dict_structs(d_in.ptr, len(arena), d_out.ptr, off + 16, ds_off)
    status, _ = plan.status()
    host = d_out.download()
    for c, o, st in zip(cases, ds_off, status):
        assert st == len(c.pack)
        if c.encode_type != O.DICT_COMPRESS:
            continue
        ds_o, c_o = O.get_dict_struct(c.pack, c.data_type, c.n)
        assert c_o == len(c.pack)
        assert host[o:o + len(ds_o)].tobytes() == ds_o.tobytes(), c.name
    plan.destroy(); d_in.free(); d_out.free()


def test_dict_structs_mixed_plan_arena_holds_dictionary_packs_only():
    """Packs without a DictStruct (DELTA numbers, front-coded string packs, SIMPLE_DICT) are skipped: they get no room in
    the arena and ds_off 0, and nothing of them may be written anywhere.  A small scratch budget splits the plan into
    several groups, so a stray decode of a later group would land on DictStructs an earlier group produced."""
    from gpu_util import build_arena, align16
    r = rng(77)
    cases = []
    for rep in range(6):
        cases.append(oracle_case(np.cumsum(r.integers(0, 9, size=65536)).astype("<i4"), O.INT, O.DELTA, name="int delta %d" % rep))
        cases.append(oracle_case(_strings(r, 3000), O.STRING, O.DELTA, name="str delta %d" % rep))
        cases.append(oracle_case(_column(r, O.LONG, 30000, 500), O.LONG, O.DICT_COMPRESS, name="long dict %d" % rep))
        cases.append(oracle_case(_strings(r, 20000), O.STRING, O.DICT_COMPRESS, name="str dict %d" % rep))
        cases.append(oracle_case(_column(r, O.FLOAT, 2000, 50), O.FLOAT, O.SIMPLE_DICT, name="float simple %d" % rep))
        cases.append(oracle_case(_column(r, O.DOUBLE, 65536, 40000), O.DOUBLE, O.DICT_COMPRESS, name="double dict %d" % rep))
    descs, arena, _ = build_arena(cases)
    ds_off, off = [], 0
    for c in cases:
        if c.encode_type == O.DICT_COMPRESS:
            ds_off.append(off)
            off += align16(8 + c.data_size + 4 * c.n)
        else:
            ds_off.append(0)
    guard = 4096
    fill = 0xA7
    for budget in (0, 3 << 20):
        plan = L.Plan(0, descs, budget)
        d_in = L.DeviceBuffer(0, len(arena)); d_out = L.DeviceBuffer(0, off + guard)
        d_in.upload(arena)
        d_out.upload(np.full(off + guard, fill, dtype=np.uint8))
        plan.run_dict_structs(d_in.ptr, len(arena), d_out.ptr, off, ds_off)
        status, _ = plan.status()
        host = d_out.download()
        assert (host[off:] == fill).all(), "bytes after the arena were written"
        for c, o, st in zip(cases, ds_off, status):
            assert st == len(c.pack), c.name
            if c.encode_type != O.DICT_COMPRESS:
                continue
            ds_o, c_o = O.get_dict_struct(c.pack, c.data_type, c.n)
            assert c_o == len(c.pack)
            assert host[o:o + len(ds_o)].tobytes() == ds_o.tobytes(), "%s (scratch budget %d)" % (c.name, budget)
        plan.destroy(); d_in.free(); d_out.free()


def test_rows_from_dict_struct_with_an_output_too_small_for_the_index():
    """DictStruct.decodeStrings into a buffer that cannot even hold the row index: BUFFER_OVERFLOW, nothing written
    past the caller's buffer (PackCompressor_VLT.getDataFromDictStruct :188-213)."""
    r = rng(78)
    rows = _strings(r, 65536)
    data = O.rows_to_bytes(rows, O.STRING)
    pack, et = O.compress_pack(data, O.STRING, 65536, False)
    ds, consumed = O.get_dict_struct(pack, O.STRING, 65536)
    assert consumed == len(pack)
    for limit in (16, 4 * 65536, 4 * 65537 - 1):
        out, w = L.get_data_from_dict_struct(ds, O.STRING, 65536, limit)
        assert w == L.ERR_BUFFER_OVERFLOW, (limit, w)
    # the index fits, the body does not
    out, w = L.get_data_from_dict_struct(ds, O.STRING, 65536, 4 * 65537 + 10)
    assert w == L.ERR_BUFFER_OVERFLOW
    out, w = L.get_data_from_dict_struct(ds, O.STRING, 65536, len(data))
    assert w == len(data) and out.tobytes() == data.tobytes()


def _corrupt_variants(case, r):
    out = []
    p = case.pack
    # truncated pack
    if len(p) > 8:
        out.append(Case(p[:len(p) - 3], case.data_type, case.encode_type, case.n, case.data_size, None, case.name + " truncated"))
    # a few random byte flips in the framing area; outcome must match the oracle: same bytes, or both fail
    for k in range(6):
        q = p.copy()
        pos = int(r.integers(0, min(len(q), 64)))
        q[pos] ^= 1 << int(r.integers(0, 8))
        dec, consumed = O.decompress_pack(q, case.data_type, case.encode_type, case.n, case.data_size)
        ok = consumed == len(q)
        out.append(Case(q, case.data_type, case.encode_type, case.n, case.data_size, dec if ok else None, case.name + " flip%d@%d" % (k, pos)))
    return out


def test_corrupted_packs_fail_cleanly():
    """Errors instead of faults. For every mutated pack the GPU must either report an error (when the oracle does) or
    produce the oracle's bytes.  The GPU validates more strictly than the reference in two documented places
    (stream length words must match the walked length; dictionary ids must be < entryCount), so a mutation the oracle
    happens to accept may still be rejected by the GPU -- that is allowed, the converse is not."""
    r = rng(77)
    base = []
    for data_type in ALL_TYPES:
        base.append(oracle_case(_column(r, data_type, 3000, 100), data_type, O.DICT_COMPRESS))
        base.append(oracle_case(_column(r, data_type, 3000, 100), data_type, O.DELTA))
        base.append(oracle_case(_column(r, data_type, 3000, 100), data_type, O.SIMPLE_DICT))
    cases = []
    for c in base:
        cases += _corrupt_variants(c, r)
    outs, status, plan = run_batch(cases)
    plan.destroy()
    for c, out, st in zip(cases, outs, status):
        if c.expect is None:
            assert st != len(c.pack) or st < 0, c.name
            assert st < 0 or st != len(c.pack), c.name
        elif st == len(c.pack):
            assert out.tobytes() == c.expect.tobytes(), c.name
        else:
            assert st < 0, c.name
    # ids beyond the dictionary (bw allows 0..3, entryCount 3, id 3 used)
    entries = struct.pack("<3i", 5, 6, 7)
    ids = [0, 1, 2, 3, 1]
    ds_size = 4 + 12 + 4 + 20
    pack = struct.pack("<iiB", ds_size, 3, 0) + entries + R.rle_encode(ids, 2)
    outs, status, plan = run_batch([Case(np.frombuffer(pack, np.uint8), L.INT, L.ENC_DICT_COMPRESS, 5, 20, None, "bad id")])
    plan.destroy()
    assert status[0] == L.ERR_DATA_CORRUPTED


def test_large_launch_walk_agrees_with_small_launch_walk_on_corrupted_packs():
    """Large batches (2048 packs and more) take the framing walk with a one-wave workgroup per stream, whose record is written
    by one kernel and completed by another.  Mutated packs (flips anywhere in the first kilobyte: stream lengths, varints,
    mini-block width bytes, segment headers) must get the same status and the same bytes from it as from the
    workgroup-per-pack walk of a small batch."""
    r = rng(91)
    base = []
    for data_type in ALL_TYPES:
        base.append(oracle_case(_column(r, data_type, 5000, 300), data_type, O.DICT_COMPRESS))
        base.append(oracle_case(_column(r, data_type, 5000, 3000), data_type, O.DELTA))
    seq = (np.cumsum(r.integers(0, 2001, 20000)) + 1500000000000).astype("<i8")  # one width per block: the walk's guess holds
    base.append(oracle_case(seq, L.LONG, O.DICT_COMPRESS))
    base.append(oracle_case(seq.astype("<i4"), L.INT, O.DELTA))
    cases = list(base)
    for c in base:
        p = c.pack
        cases.append(Case(p[:len(p) - 5], c.data_type, c.encode_type, c.n, c.data_size, None, c.name + " truncated"))
        for k in range(12):
            q = p.copy()
            pos = int(r.integers(0, min(len(q), 1024)))
            q[pos] ^= 1 << int(r.integers(0, 8))
            cases.append(Case(q, c.data_type, c.encode_type, c.n, c.data_size, None, c.name + " flip%d@%d" % (k, pos)))
    outs_a, status_a, plan = run_batch(cases)
    plan.destroy()
    filler = base[0]
    big = cases + [filler] * 2000
    assert len(big) >= 2048  # the built-in choice, no switch
    outs_b, status_b, plan = run_batch(big)
    plan.destroy()
    assert len({int(x) for x in status_a}) > 3  # clean packs and more than one kind of failure
    for i, c in enumerate(cases):
        assert status_a[i] == status_b[i], (c.name, status_a[i], status_b[i])
        if status_a[i] >= 0:
            assert outs_a[i].tobytes() == outs_b[i].tobytes(), c.name
    for i, c in enumerate(base):
        assert status_b[i] == len(c.pack) and outs_b[i].tobytes() == c.expect.tobytes(), c.name
    assert all(int(x) == len(filler.pack) for x in status_b[len(cases):])


def test_string_packs_of_mixed_dictionary_sizes_in_a_large_launch():
    """String packs with dictionaries from 40 to 30000 entries side by side in one launch of more than 1024 packs (the tile
    bases come from a table of string lengths in LDS, whatever the dictionary's size; the two-kernel form of the string rows
    switches from workgroups of 16 waves to 4 at that launch size)."""
    r = rng(5)

    def uniq(m, n):
        words = [b"k%07d" % i + bytes(r.integers(97, 123, int(r.integers(0, 9))).astype(np.uint8)) for i in range(m)]
        rows = [words[i] for i in r.integers(0, m, n)]
        rows[:m] = words  # every entry is used
        return rows
    small = oracle_case(uniq(40, 300), L.STRING, O.DICT_COMPRESS, name="40 entries")
    edge_in = oracle_case(uniq(12287, 14000), L.STRING, O.DICT_COMPRESS, name="12287 entries")
    edge_out = oracle_case(uniq(12288, 14000), L.STRING, O.DICT_COMPRESS, name="12288 entries")
    big = oracle_case(uniq(30000, 40000), L.STRING, O.DICT_COMPRESS, name="30000 entries")
    cases = []
    for i in range(1040):
        cases.append((small, edge_in, small, edge_out, small, big)[i % 6] if i % 40 < 6 else small)
    check_cases(cases)


def test_string_pack_that_announces_a_wrong_dict_struct_size():
    """The reference decodes the DictStruct into a buffer of the size the pack's first word announces: a smaller figure
    fails (BUFFER_OVERFLOW), a larger one is accepted.  Found by tools/soak.py as a one-bit flip the decoder let through."""
    r = rng(3)
    few = [b"www.s%d.com/%d" % (i % 5, i % 4) for i in range(200)]                 # plain entries
    many = [b"www.s%d.com/%d" % (i % 37, i % 500) for i in r.integers(0, 5000, 9000)]  # front-coded entries
    cases = []
    for rows in (few, many):
        c = oracle_case(rows, L.STRING, O.DICT_COMPRESS)
        size = int(c.pack[:4].view("<i4")[0])
        for delta in (-8, -1, 1, 64):
            q = c.pack.copy()
            q[:4] = np.frombuffer(np.int32(size + delta).tobytes(), dtype=np.uint8)
            dec, consumed = O.decompress_pack(q, c.data_type, c.encode_type, c.n, c.data_size)
            assert (consumed == len(q)) == (delta > 0)
            cases.append(Case(q, c.data_type, c.encode_type, c.n, c.data_size, dec if delta > 0 else None, "%s size%+d" % (c.name, delta)))
    check_cases(cases)


def test_bad_arguments():
    d = L.PackDesc(8, 0, 10, 40, 10, L.ENC_DELTA, L.INT, 0)  # misaligned in_off
    with pytest.raises(L.IxrError):
        L.Plan(0, [d])
    d = L.PackDesc(0, 0, 10, 40, 70000, L.ENC_DELTA, L.INT, 0)  # too many rows
    with pytest.raises(L.IxrError):
        L.Plan(0, [d])
    good = L.PackDesc(0, 0, 100, 400, 100, L.ENC_DELTA, L.INT, 0)
    plan = L.Plan(0, [good])
    buf = L.DeviceBuffer(0, 1024)
    with pytest.raises(L.IxrError):
        plan.run(buf.ptr, 100, buf.ptr, 1024)  # arena smaller than pack + slack
    plan.destroy(); buf.free()


def test_config1_one_million_row_segment():
    """BASELINE config 1: 1 000 000 rows = 15 full packs + 16 960 rows, columns {INT, BIGINT, DOUBLE, VARCHAR}; every decoded
    pack compared byte for byte with the CPU reader (oracle)."""
    kinds = [L.COL_I_CAT, L.COL_L_TS, L.COL_D_CARD, L.COL_S_URL]
    reqs = []
    for kind in kinds:
        for p in range(16):
            reqs.append((kind, p, 65536 if p < 15 else 1000000 - 15 * 65536, 0))
    descs, arena, out_bytes, sums = L.synth_build(reqs, 0xC0FFEE, threads=8)
    plan = L.Plan(0, descs)
    d_in = L.DeviceBuffer(0, len(arena)); d_out = L.DeviceBuffer(0, out_bytes + 16)
    d_in.upload(arena)
    plan.run(d_in.ptr, len(arena), d_out.ptr, out_bytes + 16)
    status, bad = plan.status()
    assert bad == 0
    assert plan.total_rows == 4 * 1000000
    host = d_out.download()
    for d in descs:
        dec, consumed = O.decompress_pack(arena[d.in_off:d.in_off + d.pack_size], d.data_type, d.encode_type, d.obj_count, d.data_size)
        assert consumed == d.pack_size
        assert host[d.out_off:d.out_off + d.data_size].tobytes() == dec.tobytes()
    assert (plan.checksums(d_out.ptr, out_bytes + 16) == sums).all()
    plan.destroy(); d_in.free(); d_out.free()


def test_coexists_with_torch_when_torch_is_imported_first():
    """bench.py needs torch.distributed for N > 1.  torch bundles its own libamdhip64 (same soname): importing torch first
    makes libixrgpu bind to that copy, so one process holds one HIP runtime.  Run in a child process to keep this test
    process on the system runtime."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import torch; assert torch.cuda.is_available(); torch.zeros(4, device='cuda').sum().item(); "
            "import __graft_entry__ as g; g.smoke()")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "smoke ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_explicit_stream_repeated_runs_multiple_groups():
    """The batch path on a caller-created stream: numeric and string packs in one plan (the two kinds of kernels run on two
    internal streams that fork from and join the caller's), a scratch budget small enough for several groups, three
    runs back to back into the same output arena; every run's checksums must match the host rows'."""
    reqs = []
    for pack_id in range(6):
        for kind in (2, 4, 8, 10, 0, 9):
            reqs.append((kind, pack_id, 65536 if pack_id < 5 else 777, 0))
    descs, arena, out_bytes, sums = L.synth_build(reqs, 0xBEEF, threads=4)
    plan = L.Plan(0, descs, 24 << 20)
    stream = L.lib.ixr_stream_create(0)
    assert stream
    d_in = L.DeviceBuffer(0, len(arena))
    d_out = L.DeviceBuffer(0, out_bytes + 16)
    try:
        d_in.upload(arena)
        for rep in range(3):
            d_out.upload(np.full(out_bytes + 16, 0x5A + rep, dtype=np.uint8))
            plan.run(d_in.ptr, len(arena), d_out.ptr, out_bytes + 16, stream)
            status, bad = plan.status(stream)
            assert bad == 0 and all(int(s) == d.pack_size for s, d in zip(status, descs))
            got = plan.checksums(d_out.ptr, out_bytes + 16, stream)
            assert (got == sums).all(), "run %d" % rep
        assert L.lib.ixr_stream_synchronize(0, stream) == 0
    finally:
        plan.destroy()
        d_in.free()
        d_out.free()
        L.lib.ixr_stream_destroy(0, stream)


@pytest.mark.gpu
def test_host_arenas_pipelined_run_matches_the_device_run():
    """ixr_plan_run_host: both arenas in pinned host memory, the plan cut into several groups (small scratch budget) that
    go through two staging buffers per direction.  Every pack's rows must equal the source rows (and therefore what
    ixr_plan_run leaves on the device); two runs back to back reuse the staging buffers and streams; a pack corrupted in
    the host arena is reported by ixr_plan_status and spoils no other pack."""
    reqs = []
    for pack_id in range(7):
        for kind in (2, 4, 8, 10, 0, 9, 12):
            reqs.append((kind, pack_id, 65536 if pack_id < 6 else 4321, 0))
    descs, arena, out_bytes, sums = L.synth_build(reqs, 0xFACADE, threads=4)
    plan = L.Plan(0, descs, 24 << 20)  # ~ a dozen groups
    h_in, h_out = L.PinnedBuffer(len(arena)), L.PinnedBuffer(out_bytes + 16)
    try:
        h_in.array[:] = arena
        for rep in range(2):
            h_out.array[:] = 0xA5
            plan.run_host(h_in.ptr, len(arena), h_out.ptr, out_bytes + 16)
            status, bad = plan.status()
            assert bad == 0 and all(int(s) == d.pack_size for s, d in zip(status, descs))
            for (kind, pack_id, rows, _), d in zip(reqs, descs):
                data = L.synth_column(kind, 0xFACADE, pack_id, rows)
                assert h_out.array[d.out_off:d.out_off + d.data_size].tobytes() == data.tobytes(), (rep, L.COL_NAMES[kind], pack_id)
        # a corrupted pack in the middle: its status is negative, every other pack still decodes to its rows
        victim = len(descs) // 2
        dv = descs[victim]
        h_in.array[dv.in_off:dv.in_off + 4] = 0xFF
        h_out.array[:] = 0
        plan.run_host(h_in.ptr, len(arena), h_out.ptr, out_bytes + 16)
        status, bad = plan.status()
        assert bad >= 1 and int(status[victim]) != dv.pack_size
        for i, ((kind, pack_id, rows, _), d) in enumerate(zip(reqs, descs)):
            if int(status[i]) == d.pack_size:
                data = L.synth_column(kind, 0xFACADE, pack_id, rows)
                assert h_out.array[d.out_off:d.out_off + d.data_size].tobytes() == data.tobytes(), (L.COL_NAMES[kind], pack_id)
        assert sum(int(status[i]) != d.pack_size for i, d in enumerate(descs)) == bad
        # arenas that are too small are refused before anything runs
        with pytest.raises(L.IxrError):
            plan.run_host(h_in.ptr, len(arena) - 64, h_out.ptr, out_bytes + 16)
    finally:
        plan.destroy()
        h_in.free()
        h_out.free()
