"""Predicates on the rows of decoded packs (ExtIndex_SimpleBits.java:25-331, what packs without a dictionary get).
CPU: the oracle against a direct numpy statement of the predicate.  GPU: ixr_scan_filter_run on the arena a decode leaves,
against the oracle, bit for bit, for every op and type."""
import numpy as np
import pytest

import oracle_lib as O
from indexr_amd import lib as L
from test_filter import LIKE_PATTERNS, bitmap_of, dbits, like_column, like_regex, make_columns

NUMERIC = (O.INT, O.LONG, O.FLOAT, O.DOUBLE)


def numeric_special_columns():
    """rows where the language's comparisons and a bit-wise reading differ: NaN, -0.0, infinities, extreme integers"""
    f = np.array([0.0, -0.0, 1.5, np.nan, np.inf, -np.inf, 2.5, 1.5, -1.5, 3.4e38], dtype="<f4")
    d = np.array([0.0, -0.0, 1.5, np.nan, np.inf, -np.inf, 2.5, 1.5, -1.5, 1e308, 5e-324], dtype="<f8")
    i = np.array([0, -1, 2**31 - 1, -2**31, 7, 7, 8], dtype="<i4")
    q = np.array([0, -1, 2**63 - 1, -2**63, 7, 2**32 + 7, 8], dtype="<i8")
    return {O.INT: i, O.LONG: q, O.FLOAT: f, O.DOUBLE: d}


def special_predicates(dt):
    if dt == O.INT:  # (int) numValue: the key's low 32 bits
        return [(L.FILTER_EQUAL, [2**32 + 7]), (L.FILTER_IN, [-1, 2**31 - 1, 2**33]), (L.FILTER_GREATER, [-2**31]), (L.FILTER_GREATER_EQUAL, [2**31 - 1]),
                (L.FILTER_BETWEEN, [-1, 7]), (L.FILTER_BETWEEN, [8, 7]), (L.FILTER_GREATER, [2**31 - 1])]
    if dt == O.LONG:
        return [(L.FILTER_EQUAL, [2**32 + 7]), (L.FILTER_IN, [-2**63, 2**63 - 1]), (L.FILTER_GREATER, [-2**63]), (L.FILTER_GREATER_EQUAL, [2**63 - 1]),
                (L.FILTER_BETWEEN, [-1, 8]), (L.FILTER_GREATER, [2**63 - 1])]
    return [(L.FILTER_EQUAL, [dbits(0.0)]), (L.FILTER_EQUAL, [dbits(-0.0)]), (L.FILTER_EQUAL, [dbits(float("nan"))]), (L.FILTER_IN, [dbits(float("nan")), dbits(1.5)]),
            (L.FILTER_GREATER, [dbits(float("-inf"))]), (L.FILTER_GREATER_EQUAL, [dbits(float("-inf"))]), (L.FILTER_GREATER_EQUAL, [dbits(float("nan"))]),
            (L.FILTER_BETWEEN, [dbits(-0.0), dbits(2.5)]), (L.FILTER_BETWEEN, [dbits(float("-inf")), dbits(float("inf"))]), (L.FILTER_GREATER, [dbits(float("inf"))]),
            (L.FILTER_EQUAL, [dbits(1.1)])]  # 1.1 narrows to a float no row holds


def numpy_mask(dt, rows, op, keys):
    """the predicate on the rows with numpy's IEEE / integer comparisons"""
    np_t = {O.INT: np.int32, O.LONG: np.int64, O.FLOAT: np.float32, O.DOUBLE: np.float64}[dt]

    def key(k):
        if dt == O.INT:
            return np.array([k], dtype=np.int64).astype(np.uint32).view(np.int32)[0]
        if dt == O.LONG:
            return np.int64(k)
        return np_t(np.array([k], dtype=np.int64).view(np.float64)[0])
    a = np.asarray(rows, dtype=np_t)
    with np.errstate(invalid="ignore", over="ignore"):
        ks = [key(k) for k in keys]
        if op == L.FILTER_EQUAL:
            return a == ks[0]
        if op == L.FILTER_IN:
            m = np.zeros(len(a), dtype=bool)
            for k in ks:
                m |= a == k
            return m
        if op == L.FILTER_GREATER:
            return a > ks[0]
        if op == L.FILTER_GREATER_EQUAL:
            return a >= ks[0]
        return (a >= ks[0]) & (a <= ks[1])


def string_mask(rows, op, keys):
    if op in (L.FILTER_EQUAL, L.FILTER_IN):
        ks = set(keys)
        return np.array([v in ks for v in rows])
    if op == L.FILTER_GREATER:
        return np.array([v > keys[0] for v in rows])
    if op == L.FILTER_GREATER_EQUAL:
        return np.array([v >= keys[0] for v in rows])
    if op == L.FILTER_BETWEEN:
        return np.array([keys[0] <= v <= keys[1] for v in rows])
    rx = like_regex(keys[0])
    return np.array([rx.fullmatch(v) is not None for v in rows])


def general_predicates(dt, rows):
    vals = sorted(set(rows))
    lo, hi, mid = vals[0], vals[-1], vals[len(vals) // 2]
    if dt == O.STRING:
        K = lambda v: v
        absent, below, above = mid + b"~absent", b"", hi + b"\xff"
    else:
        K = (lambda v: dbits(v)) if dt in (O.FLOAT, O.DOUBLE) else (lambda v: int(v))
        absent, below, above = (mid + 0.03125 if dt in (O.FLOAT, O.DOUBLE) else int(mid) + 1), lo - 5, hi + 5
    out = [(L.FILTER_EQUAL, [K(mid)]), (L.FILTER_EQUAL, [K(absent)]), (L.FILTER_IN, [K(lo), K(absent), K(hi), K(mid)]),
           (L.FILTER_IN, [K(absent), K(above)]), (L.FILTER_IN, [K(v) for v in vals[::5]] + [K(absent)] * 70),
           (L.FILTER_GREATER, [K(mid)]), (L.FILTER_GREATER_EQUAL, [K(mid)]), (L.FILTER_GREATER, [K(hi)]), (L.FILTER_GREATER_EQUAL, [K(below)]),
           (L.FILTER_BETWEEN, [K(lo), K(mid)]), (L.FILTER_BETWEEN, [K(absent), K(hi)]), (L.FILTER_BETWEEN, [K(mid), K(lo)]),
           (L.FILTER_BETWEEN, [K(above), K(above)])]
    if dt == O.STRING:
        out += [(L.FILTER_LIKE, [p]) for p in LIKE_PATTERNS]
    return out


def all_cases(n, seed):
    """(data type, rows, decoded bytes, op, keys)"""
    rng = np.random.default_rng(seed)
    cols = make_columns(n, rng)
    out = []
    for dt in NUMERIC + (O.STRING,):
        rows = list(cols[dt]) if dt == O.STRING else cols[dt]
        data = O.rows_to_bytes(rows, dt)
        for op, keys in general_predicates(dt, rows if dt == O.STRING else rows.tolist()):
            out.append((dt, rows, data, op, keys))
    for dt, rows in numeric_special_columns().items():
        data = O.rows_to_bytes(rows, dt)
        for op, keys in special_predicates(dt):
            out.append((dt, rows, data, op, keys))
    wild, _ = like_column(n, rng)
    wdata = O.rows_to_bytes(wild, O.STRING)
    for pat in LIKE_PATTERNS + [b"a%", b"%b", b"\\%%", b"\\\\%", b"_\\__"]:
        out.append((O.STRING, wild, wdata, L.FILTER_LIKE, [pat]))
    out.append((O.STRING, wild, wdata, L.FILTER_EQUAL, [b""]))
    out.append((O.STRING, wild, wdata, L.FILTER_GREATER, [b""]))
    return out


def test_oracle_scan_filter_selects_the_rows_that_satisfy_the_predicate():
    kinds = set()
    for dt, rows, data, op, keys in all_cases(5000, 23):
        mask = string_mask(rows, op, keys) if dt == O.STRING else numpy_mask(dt, rows, op, keys)
        r, bm = O.scan_filter_pack(data, dt, len(rows), op, keys)
        kinds.add(r)
        assert r == (L.BITMAP_SOME if mask.any() else L.BITMAP_NONE), (dt, op, keys[:3])  # never ALL
        assert np.array_equal(bm, bitmap_of(mask)), (dt, op, keys[:3])
    assert kinds == {L.BITMAP_NONE, L.BITMAP_SOME}


def test_oracle_scan_filter_rejects_like_on_numbers_and_short_packs():
    ints = np.arange(100, dtype="<i4")
    data = O.rows_to_bytes(ints, O.INT)
    assert O.scan_filter_pack(data, O.INT, 100, L.FILTER_LIKE, [b"1%"])[0] == -102
    assert O.scan_filter_pack(data[:-4], O.INT, 100, L.FILTER_EQUAL, [3])[0] == -1000
    s = O.rows_to_bytes([b"ab", b"c"], O.STRING).copy()
    s[8:12] = np.frombuffer(np.int32(99).tobytes(), dtype=np.uint8)  # the last offset leaves the pack
    assert O.scan_filter_pack(s, O.STRING, 2, L.FILTER_EQUAL, [b"c"])[0] == O.DATA_CORRUPTED


@pytest.mark.gpu
def test_gpu_scan_filter_matches_oracle_for_every_op_and_type():
    cases = all_cases(65536, 77) + all_cases(12345, 78)
    ints = np.arange(100, dtype="<i4")
    cases.append((O.INT, ints, O.rows_to_bytes(ints, O.INT), L.FILTER_LIKE, [b"1%"]))
    bad = O.rows_to_bytes([b"ab", b"c"] * 700, O.STRING).copy()
    bad[4 * 1300:4 * 1300 + 4] = np.frombuffer(np.int32(1 << 30).tobytes(), dtype=np.uint8)
    cases.append((O.STRING, [b"ab", b"c"] * 700, bad, L.FILTER_EQUAL, [b"c"]))
    # the decoded arena: every distinct pack once
    arena_parts, offs, pos, seen = [], [], 0, {}
    for dt, rows, data, op, keys in cases:
        key = id(data)
        if key not in seen:
            seen[key] = pos
            arena_parts.append(np.asarray(data, dtype=np.uint8))
            pad = -len(data) % 256
            arena_parts.append(np.zeros(pad, dtype=np.uint8))
            pos += len(data) + pad
        offs.append(seen[key])
    arena = np.concatenate(arena_parts)
    values, voffs = L.filter_keys([k for *_, k in cases], [dt == O.STRING or op == L.FILTER_LIKE for dt, _, _, op, _ in cases])
    descs = [L.ScanFilterDesc(o, vo, dt, len(rows), len(data), op, len(keys), 0)
             for (dt, rows, data, op, keys), o, vo in zip(cases, offs, voffs)]
    d_dec = L.DeviceBuffer(0, len(arena))
    d_bm = L.DeviceBuffer(0, L.BITMAP_BYTES * len(cases))
    try:
        d_dec.upload(arena)
        d_bm.upload(np.full(L.BITMAP_BYTES * len(cases), 0xEE, dtype=np.uint8))
        res, nbad = L.scan_filter_run(descs, d_dec.ptr, len(arena), values, d_bm.ptr, L.BITMAP_BYTES * len(cases))
        host = d_bm.download().view("<u8").reshape(len(cases), 1024)
        kinds = set()
        for i, (dt, rows, data, op, keys) in enumerate(cases):
            okeys = [0] if (op == L.FILTER_LIKE and dt != O.STRING) else keys
            want, bm = O.scan_filter_pack(data, dt, len(rows), op, okeys)
            kinds.add(int(want))
            assert res[i] == want, (i, dt, op, keys[:3], res[i], want)
            if want >= 0:
                assert np.array_equal(host[i], bm), (i, dt, op, keys[:3])
        assert kinds == {L.BITMAP_NONE, L.BITMAP_SOME, -102, O.DATA_CORRUPTED}
        assert nbad == 2
    finally:
        d_dec.free()
        d_bm.free()


@pytest.mark.gpu
def test_gpu_scan_filter_on_the_output_of_a_decode():
    """decode DELTA and dictionary packs, then filter the decoded arena in place"""
    import gpu_util as G
    rng = np.random.default_rng(9)
    n = 40000
    seq = (np.cumsum(rng.integers(0, 4, n)) + 1000).astype("<i4")
    cols = make_columns(n, rng)
    specs = [(O.INT, seq, O.DELTA, L.FILTER_BETWEEN, [20000, 50000]),
             (O.LONG, cols[O.LONG], None, L.FILTER_GREATER, [int(np.median(cols[O.LONG]))]),
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
    d_in, d_out = L.DeviceBuffer(0, len(arena)), L.DeviceBuffer(0, out_bytes)
    d_bm = L.DeviceBuffer(0, L.BITMAP_BYTES * len(cases))
    try:
        d_in.upload(arena)
        plan.run(d_in.ptr, len(arena), d_out.ptr, out_bytes)  # the filter below is ordered after it on the same stream
        values, voffs = L.filter_keys([k for *_, k in specs], [dt == O.STRING for dt, *_ in specs])
        fd = [L.ScanFilterDesc(d.out_off, vo, d.data_type, d.obj_count, d.data_size, op, len(keys), 0)
              for d, (_, _, _, op, keys), vo in zip(descs, specs, voffs)]
        res, nbad = L.scan_filter_run(fd, d_out.ptr, out_bytes, values, d_bm.ptr, L.BITMAP_BYTES * len(cases))
        assert nbad == 0
        host = d_bm.download().view("<u8").reshape(len(cases), 1024)
        for i, (dt, rows, _, op, keys) in enumerate(specs):
            want, bm = O.scan_filter_pack(O.rows_to_bytes(rows, dt), dt, n, op, keys)
            assert want == L.BITMAP_SOME and res[i] == want and np.array_equal(host[i], bm), i
    finally:
        plan.destroy()
        d_in.free(); d_out.free(); d_bm.free()


@pytest.mark.gpu
def test_gpu_scan_filter_rejects_bad_descriptors():
    data = O.rows_to_bytes(np.arange(100, dtype="<i4"), O.INT)
    d_dec, d_bm = L.DeviceBuffer(0, 1024), L.DeviceBuffer(0, L.BITMAP_BYTES)
    try:
        d_dec.upload(np.concatenate([data, np.zeros(1024 - 400, dtype=np.uint8)]))
        values = np.zeros(24, dtype=np.uint8)
        good = (0, 0, O.INT, 100, 400, L.FILTER_EQUAL, 1, 0)
        res, bad = L.scan_filter_run([L.ScanFilterDesc(*good)], d_dec.ptr, 1024, values, d_bm.ptr, L.BITMAP_BYTES)
        assert bad == 0 and res[0] == L.BITMAP_SOME
        for change in ({0: 1024}, {0: 4}, {1: 24}, {1: 4}, {2: 9}, {3: 0}, {3: 70000}, {5: 0}, {5: 7}, {6: 0}, {6: 4}):
            f = list(good)
            for k, v in change.items():
                f[k] = v
            with pytest.raises(L.IxrError) as e:
                L.scan_filter_run([L.ScanFilterDesc(*f)], d_dec.ptr, 1024, values, d_bm.ptr, L.BITMAP_BYTES)
            assert e.value.code == L.ERR_INVALID_ARGUMENT, change
        with pytest.raises(L.IxrError):
            L.scan_filter_run([L.ScanFilterDesc(*good)], d_dec.ptr, 1024, values, d_bm.ptr, L.BITMAP_BYTES - 8)
        # a data_size too small for the rows is reported per pack
        res, bad = L.scan_filter_run([L.ScanFilterDesc(0, 0, O.INT, 100, 396, L.FILTER_EQUAL, 1, 0)], d_dec.ptr, 1024, values, d_bm.ptr, L.BITMAP_BYTES)
        assert bad == 1 and res[0] == L.ERR_INVALID_ARGUMENT
    finally:
        d_dec.free()
        d_bm.free()
