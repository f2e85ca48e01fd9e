"""The rows a bitmap picks: DataPack.foreach(BitMap position, XxxSetter) (DataPack.java:273-306, 324-357, 375-408, 426-459,
708-762).  CPU: the oracle against numpy indexing.  GPU: ixr_select_run on decoded packs and on the bitmaps the filter
calls leave, against the oracle, bit for bit."""
import numpy as np
import pytest

import oracle_lib as O
from indexr_amd import lib as L
from test_filter import bitmap_of, make_columns

NP_T = {O.INT: "<i4", O.LONG: "<i8", O.FLOAT: "<f4", O.DOUBLE: "<f8"}


def masks(n, rng):
    """None stands for BitMap.ALL"""
    m_none = np.zeros(n, dtype=bool)
    m_every = np.ones(n, dtype=bool)
    m_sparse = rng.random(n) < 0.01
    m_half = rng.random(n) < 0.5
    m_runs = (np.arange(n) // 700) % 3 == 1
    m_ends = np.zeros(n, dtype=bool)
    m_ends[[0, n - 1]] = True
    return [None, m_none, m_every, m_sparse, m_half, m_runs, m_ends]


def select_cases(n, seed):
    """(data type, rows, decoded bytes, bitmap words or None, mask or None)"""
    rng = np.random.default_rng(seed)
    cols = make_columns(n, rng)
    out = []
    for dt in (O.INT, O.LONG, O.FLOAT, O.DOUBLE, O.STRING):
        rows = list(cols[dt]) if dt == O.STRING else cols[dt]
        data = O.rows_to_bytes(rows, dt)
        for m in masks(n, rng):
            out.append((dt, rows, data, None if m is None else bitmap_of(m), m))
    return out


def expected(dt, rows, mask):
    n = len(rows)
    idx = np.arange(n) if mask is None else np.flatnonzero(mask[:n])
    if dt == O.STRING:
        return idx.astype(np.int32), [rows[i] for i in idx]
    return idx.astype(np.int32), np.asarray(rows, dtype=NP_T[dt])[idx]


def same_values(dt, got, want):
    if dt == O.STRING:
        return got == want
    w = np.dtype(NP_T[dt]).itemsize
    return np.array_equal(np.asarray(got).view("u%d" % w), np.asarray(want).view("u%d" % w))  # bits, so NaNs compare


def test_oracle_select_visits_the_set_bits_in_row_order():
    for n in (1, 63, 64, 65, 5000):
        for dt, rows, data, bm, mask in select_cases(n, 100 + n):
            cnt, ids, vals = O.select_pack(data, dt, n, bm)
            want_ids, want_vals = expected(dt, rows, mask)
            assert cnt == len(want_ids), (n, dt)
            assert np.array_equal(ids, want_ids), (n, dt)
            assert same_values(dt, vals, want_vals), (n, dt)


def test_oracle_select_stops_at_obj_count_and_checks_string_offsets():
    ints = np.arange(100, dtype="<i4") * 3
    bm = np.zeros(1024, dtype=np.uint64)
    bm[0] = np.uint64(0b1011)
    bm[1] = np.uint64(1 << 35)  # row 99
    bm[1] |= np.uint64(1 << 36)  # row 100: not a row of the pack
    bm[500] = np.uint64(7)
    cnt, ids, vals = O.select_pack(O.rows_to_bytes(ints, O.INT), O.INT, 100, bm)
    assert cnt == 4 and ids.tolist() == [0, 1, 3, 99] and vals.tolist() == [0, 3, 9, 297]
    assert O.select_pack(O.rows_to_bytes(ints, O.INT)[:-4], O.INT, 100, bm)[0] == -1000
    s = O.rows_to_bytes([b"ab", b"c", b"def"], O.STRING).copy()
    s[8:12] = np.frombuffer(np.int32(99).tobytes(), dtype=np.uint8)
    assert O.select_pack(s, O.STRING, 3, None)[0] == O.DATA_CORRUPTED
    pick02 = np.zeros(1024, dtype=np.uint64)
    pick02[0] = np.uint64(0b101)
    assert O.select_pack(s, O.STRING, 3, pick02)[0] == O.DATA_CORRUPTED  # row 2 starts at the bad offset
    pick0 = np.zeros(1024, dtype=np.uint64)
    pick0[0] = np.uint64(1)
    cnt, ids, vals = O.select_pack(s, O.STRING, 3, pick0)  # only the rows visited are read, as in the reference
    assert cnt == 1 and vals == [b"ab"]


def overlapping_strings_pack():
    """offsets that step backwards between picked rows: rows 0, 2, 4 ... all read bytes [0, 6) of a 6-byte body"""
    n = 8
    offs = np.array([0, 6] * (n // 2) + [0], dtype="<i4")
    data = np.concatenate([offs.view(np.uint8), np.frombuffer(b"abcdef", dtype=np.uint8)])
    bm = np.zeros(1024, dtype=np.uint64)
    bm[0] = np.uint64(0b01010101)
    return n, data, bm


def test_oracle_select_reports_picked_bytes_beyond_the_body():
    n, data, bm = overlapping_strings_pack()
    assert O.select_pack(data, O.STRING, n, bm)[0] == O.DATA_CORRUPTED
    one = np.zeros(1024, dtype=np.uint64)
    one[0] = np.uint64(0b100)
    cnt, ids, vals = O.select_pack(data, O.STRING, n, one)
    assert cnt == 1 and ids.tolist() == [2] and vals == [b"abcdef"]


def read_select_output(host, desc, cnt, dt):
    """the arrays ixr_select_run wrote for one descriptor -> (row ids, values)"""
    n, base = desc.obj_count, desc.dst_off
    ids = host[base:base + 4 * n].view("<i4")[:cnt]
    if dt == O.STRING:
        ao = host[base + 4 * n:base + 8 * n].view("<i4")[:cnt]
        al = host[base + 8 * n:base + 12 * n].view("<i4")[:cnt]
        body = host[base + 12 * n:base + int(L.lib.ixr_select_bound(desc))].tobytes()
        if cnt:  # the bytes lie one after the other
            assert ao[0] == 0 and np.array_equal(ao[1:], np.cumsum(al)[:-1])
        return ids, [body[ao[k]:ao[k] + al[k]] for k in range(cnt)]
    w = np.dtype(NP_T[dt]).itemsize
    vo = base + ((4 * n + 7) & ~7)
    return ids, host[vo:vo + w * cnt].view(NP_T[dt])


def test_select_bound_covers_the_layout():
    assert L.lib.ixr_select_bound(L.SelectDesc(0, 0, 0, O.INT, 100, 400, 0)) == 800
    assert L.lib.ixr_select_bound(L.SelectDesc(0, 0, 0, O.LONG, 101, 808, 1)) == (408 + 808 + 15) // 16 * 16
    assert L.lib.ixr_select_bound(L.SelectDesc(0, 0, 0, O.DOUBLE, 3, 24, 0)) == 48
    assert L.lib.ixr_select_bound(L.SelectDesc(0, 0, 0, O.STRING, 10, 44 + 123, 0)) == (120 + 123 + 15) // 16 * 16


@pytest.mark.gpu
def test_gpu_select_matches_oracle_for_every_type_and_bitmap():
    cases = select_cases(65536, 5) + select_cases(12345, 6) + select_cases(1, 7) + select_cases(1025, 8)
    # bits at and above objCount are ignored
    ints = (np.arange(100, dtype="<i4") * 3).astype("<i4")
    bm = np.full(1024, np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    cases.append((O.INT, ints, O.rows_to_bytes(ints, O.INT), bm, np.ones(100, dtype=bool)))
    n_ov, ov_data, ov_bm = overlapping_strings_pack()
    cases.append((O.STRING, None, ov_data, ov_bm, None))
    bad = O.rows_to_bytes([b"ab", b"c"] * 700, O.STRING).copy()
    bad[4 * 1300:4 * 1300 + 4] = np.frombuffer(np.int32(1 << 30).tobytes(), dtype=np.uint8)
    cases.append((O.STRING, None, bad, None, None))
    arena_parts, src, pos, seen = [], [], 0, {}
    for dt, rows, data, bm, mask in cases:
        if id(data) not in seen:
            seen[id(data)] = pos
            arena_parts.append(np.asarray(data, dtype=np.uint8))
            pad = -len(data) % 256
            arena_parts.append(np.zeros(pad, dtype=np.uint8))
            pos += len(data) + pad
        src.append(seen[id(data)])
    arena = np.concatenate(arena_parts)
    bitmaps = np.concatenate([np.zeros(1024, dtype=np.uint64) if bm is None else bm for _, _, _, bm, _ in cases])
    descs, dst = [], 0
    for i, ((dt, rows, data, bm, mask), so) in enumerate(zip(cases, src)):
        n = n_ov if data is ov_data else (1400 if data is bad else len(rows))
        d = L.SelectDesc(so, dst, i * L.BITMAP_BYTES, dt, n, len(data), 1 if bm is None else 0)
        descs.append(d)
        dst += (int(L.lib.ixr_select_bound(d)) + 255) // 256 * 256
    d_dec, d_bm, d_out = L.DeviceBuffer(0, len(arena)), L.DeviceBuffer(0, bitmaps.nbytes), L.DeviceBuffer(0, dst)
    try:
        d_dec.upload(arena)
        d_bm.upload(bitmaps.view(np.uint8))
        d_out.upload(np.full(dst, 0xEE, dtype=np.uint8))
        cnt, nbad = L.select_run(descs, d_dec.ptr, len(arena), d_bm.ptr, bitmaps.nbytes, d_out.ptr, dst)
        host = d_out.download()
        for i, ((dt, rows, data, bm, mask), d) in enumerate(zip(cases, descs)):
            want, ids, vals = O.select_pack(data, dt, d.obj_count, bm)
            assert cnt[i] == want, (i, dt, cnt[i], want)
            if want < 0:
                continue
            got_ids, got_vals = read_select_output(host, d, want, dt)
            assert np.array_equal(got_ids, ids), (i, dt)
            assert same_values(dt, got_vals, vals), (i, dt)
            # nothing written past the descriptor's bound
            end = d.dst_off + int(L.lib.ixr_select_bound(d))
            nxt = descs[i + 1].dst_off if i + 1 < len(descs) else dst
            assert (host[end:nxt] == 0xEE).all(), i
        assert nbad == 2 and cnt[-1] == O.DATA_CORRUPTED and cnt[-2] == O.DATA_CORRUPTED
    finally:
        d_dec.free(); d_bm.free(); d_out.free()


@pytest.mark.gpu
def test_gpu_select_after_decode_and_filter():
    """decode packs, filter them on the device, hand out the rows the bitmaps pick: nothing leaves the device in between"""
    import gpu_util as G
    rng = np.random.default_rng(19)
    n = 50000
    cols = make_columns(n, rng)
    seq = (np.cumsum(rng.integers(0, 4, n)) + 1000).astype("<i4")
    specs = [(O.INT, seq, O.DELTA, L.FILTER_BETWEEN, [20000, 50000]),
             (O.DOUBLE, cols[O.DOUBLE], None, L.FILTER_GREATER, [int(np.asarray([np.median(cols[O.DOUBLE])]).view("<i8")[0])]),
             (O.STRING, list(cols[O.STRING]), None, L.FILTER_LIKE, [b"www.s1%/2%"]),
             (O.STRING, list(cols[O.STRING]), O.DELTA, L.FILTER_GREATER_EQUAL, [b"www.s2"])]
    cases = []
    for dt, rows, enc, _, _ in specs:
        data = O.rows_to_bytes(rows, dt)
        if enc is None:
            pack, et = O.compress_pack(data, dt, n, True)
        else:
            pack, et = O.compress_pack_as(data, dt, n, enc), enc
        cases.append(G.Case(pack, dt, et, n, len(data)))
    descs, arena, out_bytes = G.build_arena(cases)
    plan = L.Plan(0, descs)
    bm_bytes = L.BITMAP_BYTES * len(cases)
    sel, dst = [], 0
    for i, d in enumerate(descs):
        s = L.SelectDesc(d.out_off, dst, i * L.BITMAP_BYTES, d.data_type, d.obj_count, d.data_size, 0)
        sel.append(s)
        dst += (int(L.lib.ixr_select_bound(s)) + 255) // 256 * 256
    d_in, d_out, d_bm, d_sel = L.DeviceBuffer(0, len(arena)), L.DeviceBuffer(0, out_bytes), L.DeviceBuffer(0, bm_bytes), L.DeviceBuffer(0, dst)
    try:
        d_in.upload(arena)
        plan.run(d_in.ptr, len(arena), d_out.ptr, out_bytes)
        values, voffs = L.filter_keys([k for *_, k in specs], [dt == O.STRING for dt, *_ in specs])
        fd = [L.ScanFilterDesc(d.out_off, vo, d.data_type, d.obj_count, d.data_size, op, len(keys), 0)
              for d, (_, _, _, op, keys), vo in zip(descs, specs, voffs)]
        res, nbad = L.scan_filter_run(fd, d_out.ptr, out_bytes, values, d_bm.ptr, bm_bytes)
        assert nbad == 0 and (res == L.BITMAP_SOME).all()
        cnt, nbad = L.select_run(sel, d_out.ptr, out_bytes, d_bm.ptr, bm_bytes, d_sel.ptr, dst)
        assert nbad == 0
        host = d_sel.download()
        for i, (dt, rows, _, op, keys) in enumerate(specs):
            data = O.rows_to_bytes(rows, dt)
            _, bm = O.scan_filter_pack(data, dt, n, op, keys)
            want, ids, vals = O.select_pack(data, dt, n, bm)
            assert 0 < want < n and cnt[i] == want, i
            got_ids, got_vals = read_select_output(host, sel[i], want, dt)
            assert np.array_equal(got_ids, ids) and same_values(dt, got_vals, vals), i
    finally:
        plan.destroy()
        d_in.free(); d_out.free(); d_bm.free(); d_sel.free()


@pytest.mark.gpu
def test_gpu_select_rejects_bad_descriptors():
    data = O.rows_to_bytes(np.arange(100, dtype="<i4"), O.INT)
    d_dec, d_bm, d_out = L.DeviceBuffer(0, 1024), L.DeviceBuffer(0, L.BITMAP_BYTES), L.DeviceBuffer(0, 1024)
    try:
        d_dec.upload(np.concatenate([data, np.zeros(1024 - 400, dtype=np.uint8)]))
        d_bm.upload(np.full(L.BITMAP_BYTES, 0x01, dtype=np.uint8))
        good = (0, 0, 0, O.INT, 100, 400, 0)

        def run(f, bm_bytes=L.BITMAP_BYTES, out_bytes=1024):
            return L.select_run([L.SelectDesc(*f)], d_dec.ptr, 1024, d_bm.ptr, bm_bytes, d_out.ptr, out_bytes)
        cnt, bad = run(good)
        assert bad == 0 and cnt[0] == 13  # rows 0, 8, ... 96
        for change in ({0: 1024}, {0: 4}, {1: 8}, {1: 512}, {2: 8}, {2: 4}, {3: 0}, {3: 6}, {4: 0}, {4: 70000}, {5: -1}, {6: 2}):
            f = list(good)
            for k, v in change.items():
                f[k] = v
            with pytest.raises(L.IxrError) as e:
                run(f)
            assert e.value.code == L.ERR_INVALID_ARGUMENT, change
        with pytest.raises(L.IxrError):
            run(good, bm_bytes=L.BITMAP_BYTES - 8)
        with pytest.raises(L.IxrError):
            run(good, out_bytes=784)
        # BitMap.ALL reads no bitmap
        cnt, bad = L.select_run([L.SelectDesc(0, 0, 1 << 40, O.INT, 100, 400, 1)], d_dec.ptr, 1024, None, 0, d_out.ptr, 1024)
        assert bad == 0 and cnt[0] == 100
        # a data_size too small for the rows is reported per pack
        cnt, bad = run((0, 0, 0, O.INT, 100, 396, 0))
        assert bad == 1 and cnt[0] == L.ERR_INVALID_ARGUMENT
    finally:
        d_dec.free(); d_bm.free(); d_out.free()
