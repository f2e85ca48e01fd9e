"""Predicate evaluation on dictionary ids (ExtIndex_DictBits.java:33-200).
CPU: the oracle against a direct numpy statement of "which rows satisfy the predicate" (for keys where dictionary search
and value comparison agree: no NaN / -0.0 keys).  GPU: the device path against the oracle, bit for bit, for every op."""
import re
import struct

import numpy as np
import pytest

import oracle_lib as O
from indexr_amd import lib as L


def dbits(x):
    return struct.unpack("<q", struct.pack("<d", float(x)))[0]


def bitmap_of(mask):
    m = np.zeros(65536, dtype=bool)
    m[:len(mask)] = mask
    return np.packbits(m, bitorder="little").view("<u8")


def make_columns(n, rng):
    urls = sorted(set(b"www.s%d.com/%d" % (i % 31, i * 7 % 257) for i in range(600)))
    cols = {
        O.INT: rng.integers(-300, 300, n).astype("<i4") * 7,
        O.LONG: (rng.integers(0, 4000, n).astype("<i8") << 20) - (1 << 30),
        O.FLOAT: (rng.integers(-200, 800, n) / 8).astype("<f4"),
        O.DOUBLE: (rng.integers(-2000, 9000, n) / 16).astype("<f8"),
        O.STRING: [urls[i] for i in rng.integers(0, len(urls), n)],
    }
    return cols


def key_of(dt, v):
    if dt in (O.FLOAT, O.DOUBLE):
        return dbits(v)
    return v if dt == O.STRING else int(v)


def predicates_for(dt, rows):
    """(op, keys, expected row mask or None) for present keys, absent keys, out-of-range keys"""
    vals = sorted(set(rows))
    lo, hi, mid = vals[0], vals[-1], vals[len(vals) // 2]
    arr = rows if dt == O.STRING else np.asarray(rows)

    def mask(f):
        return np.array([f(v) for v in rows]) if dt == O.STRING else f(arr)
    if dt == O.STRING:
        absent, below, above = mid + b"~absent", b"", hi + b"\xff"
    else:
        absent, below, above = mid + 0.03125 if dt in (O.FLOAT, O.DOUBLE) else mid + 1, lo - 5, hi + 5
        if dt in (O.INT, O.LONG):
            absent = int(mid) + 1  # values are multiples of 7 / 2^20
    K = lambda v: key_of(dt, v)
    eq = lambda k: (lambda a: a == k) if dt != O.STRING else (lambda v: v == k)
    ge = lambda k: (lambda a: a >= k) if dt != O.STRING else (lambda v: v >= k)
    gt = lambda k: (lambda a: a > k) if dt != O.STRING else (lambda v: v > k)
    btw = lambda a_, b_: (lambda a: (a >= a_) & (a <= b_)) if dt != O.STRING else (lambda v: a_ <= v <= b_)
    out = [
        (L.FILTER_EQUAL, [K(mid)], mask(eq(mid))),
        (L.FILTER_EQUAL, [K(absent)], mask(eq(absent))),
        (L.FILTER_IN, [K(lo), K(absent), K(hi), K(mid)], mask(eq(lo)) | mask(eq(hi)) | mask(eq(mid))),
        (L.FILTER_IN, [K(absent), K(above)], mask(eq(absent))),
        (L.FILTER_IN, [K(v) for v in vals[::3]] + [K(absent)] * 70, None),
        (L.FILTER_GREATER, [K(mid)], mask(gt(mid))),
        (L.FILTER_GREATER_EQUAL, [K(mid)], mask(ge(mid))),
        (L.FILTER_GREATER, [K(absent)], mask(gt(absent))),
        (L.FILTER_GREATER, [K(hi)], mask(gt(hi))),
        (L.FILTER_GREATER_EQUAL, [K(below)], mask(ge(below))),
        (L.FILTER_GREATER, [K(above)], mask(gt(above))),
        (L.FILTER_BETWEEN, [K(lo), K(mid)], mask(btw(lo, mid))),
        (L.FILTER_BETWEEN, [K(absent), K(hi)], mask(btw(absent, hi))),
        (L.FILTER_BETWEEN, [K(below), K(above)], mask(btw(below, above))),
        (L.FILTER_BETWEEN, [K(mid), K(lo)], mask(btw(mid, lo))),
        (L.FILTER_BETWEEN, [K(above), K(above)], None),  # from == entryCount: the reference's checkState fails
    ]
    return out


def like_regex(pattern):
    """SQLLike.match (SQLLike.java:12-94) said as a regular expression over bytes: % -> .*, _ -> ., backslash takes the next
    byte literally; a pattern ending in a lone backslash matches nothing."""
    out, i = [], 0
    while i < len(pattern):
        c = pattern[i:i + 1]
        if c == b"\\":
            if i + 1 >= len(pattern):
                return re.compile(b"(?!)")
            out.append(re.escape(pattern[i + 1:i + 2]))
            i += 2
            continue
        out.append(b".*" if c == b"%" else b"." if c == b"_" else re.escape(c))
        i += 1
    return re.compile(b"".join(out), re.DOTALL)


def oracle_like(s, p):
    return bool(O.lib.ixo_like(O._p(s), len(s), O._p(p), len(p)))


LIKE_PATTERNS = [b"www.s1%", b"%/25_", b"%", b"%%", b"nomatch%", b"www.s_.com/%", b"%s2%.com%7", b"www.s0.com/0", b"", b"_",
                 b"%w%w%w%.%c%m%/%1%", b"www.s3.com/1__", b"%\\%%", b"%\\_%", b"www%\\", b"%.com/%%%5", b"_ww.s%"]


def like_column(n, rng):
    """strings around the wildcard and escape bytes themselves, the empty string included"""
    alphabet = [b"a", b"b", b"%", b"_", b"\\", b"ab", b""]
    pool = sorted(set(b"".join(alphabet[j] for j in rng.integers(0, len(alphabet), rng.integers(0, 7))) for _ in range(400)))
    return [pool[i] for i in rng.integers(0, len(pool), n)], pool


def test_oracle_like_agrees_with_the_regular_expression_reading():
    rng = np.random.default_rng(99)
    atoms = [b"a", b"b", b"%", b"_", b"\\", b"c"]
    fixed = [(b"windows", b"win%", True), (b"windows", b"%dow_", True), (b"mac", b"win%", False), (b"", b"%", True), (b"", b"", True),
             (b"a%b", b"a\\%b", True), (b"axb", b"a\\%b", False), (b"abc", b"____", False), (b"abc", b"abc\\", False), (b"a_c", b"a\\_c", True),
             (b"a\\c", b"a\\\\c", True), (b"aaaaaaaaaaaaaaaaaaaab", b"%a%a%a%a%a%a%a%a%a%c", False)]
    for s_, p_, want in fixed:
        assert oracle_like(s_, p_) == want, (s_, p_)
        assert bool(like_regex(p_).fullmatch(s_)) == want, (s_, p_)
    for _ in range(4000):
        s_ = b"".join(atoms[j] for j in rng.integers(0, len(atoms), rng.integers(0, 9)))
        p_ = b"".join(atoms[j] for j in rng.integers(0, len(atoms), rng.integers(0, 7)))
        assert oracle_like(s_, p_) == bool(like_regex(p_).fullmatch(s_)), (s_, p_)


def test_oracle_like_filter_marks_the_rows_whose_string_matches():
    rng = np.random.default_rng(5)
    n = 6000
    for rows in (make_columns(n, rng)[O.STRING], like_column(n, rng)[0]):
        pack, et = O.compress_pack(O.rows_to_bytes(rows, O.STRING), O.STRING, n, True)
        assert et == O.DICT_COMPRESS
        kinds = set()
        for pat in LIKE_PATTERNS + [b"a%", b"%b", b"\\%%", b"\\\\%", b"_\\__"]:
            rx = like_regex(pat)
            mask = np.array([rx.fullmatch(v) is not None for v in rows])
            r, bm = O.filter_pack(pack, O.STRING, et, n, L.FILTER_LIKE, [pat])
            kinds.add(r)
            if not mask.any():
                assert r == L.BITMAP_NONE, pat
            else:  # never ALL: the bitmap comes back as it is (fixBitmapInPack)
                assert r == L.BITMAP_SOME and np.array_equal(bm, bitmap_of(mask)), pat
        assert kinds == {L.BITMAP_NONE, L.BITMAP_SOME}
    ints = np.arange(500, dtype="<i4") % 7
    p1, e1 = O.compress_pack(O.rows_to_bytes(ints, O.INT), O.INT, 500, True)
    assert O.filter_pack(p1, O.INT, e1, 500, L.FILTER_LIKE, [0])[0] == -102  # "column type ... is illegal in LIKE"


def expected_kind(mask):
    return L.BITMAP_NONE if not mask.any() else None  # ALL / SOME depends on the dictionary shortcut, not only on the rows


@pytest.mark.parametrize("dt", [O.INT, O.LONG, O.FLOAT, O.DOUBLE, O.STRING])
def test_oracle_filter_selects_the_rows_that_satisfy_the_predicate(dt):
    rng = np.random.default_rng(17 + dt)
    n = 5000
    rows = make_columns(n, rng)[dt]
    rows = list(rows) if dt == O.STRING else rows
    data = O.rows_to_bytes(rows, dt)
    pack, et = O.compress_pack(data, dt, n, True)
    assert et == O.DICT_COMPRESS
    for op, keys, mask in predicates_for(dt, rows.tolist() if dt != O.STRING else rows):
        r, bm = O.filter_pack(pack, dt, et, n, op, keys)
        if mask is None:
            assert r in (L.BITMAP_SOME, -100)
            continue
        if not mask.any():
            assert r == L.BITMAP_NONE, (op, keys)
        elif r == L.BITMAP_ALL:
            assert mask.all(), (op, keys)
        else:
            assert r == L.BITMAP_SOME and np.array_equal(bm, bitmap_of(mask)), (op, keys)
    # a pack that is not dictionary encoded has no such index
    pack2 = O.compress_pack_as(data, dt, n, O.DELTA)
    assert O.filter_pack(pack2, dt, O.DELTA, n, L.FILTER_EQUAL, [key_of(dt, rows[0])])[0] == O.DATA_IMPROPER


def test_oracle_filter_float_specials_follow_plain_comparisons():
    vals = np.array([0.0, -0.0, 1.5, np.inf, -np.inf, 2.5, 1.5, 0.0], dtype="<f4")
    data = O.rows_to_bytes(vals, O.FLOAT)
    pack, et = O.compress_pack(data, O.FLOAT, len(vals), True)
    # NaN key: neither < nor > holds, so the binary search "finds" the middle entry (BinarySearch.java:162-181)
    r, bm = O.filter_pack(pack, O.FLOAT, et, len(vals), L.FILTER_EQUAL, [dbits(float("nan"))])
    assert r == L.BITMAP_SOME and int(bm[0]) != 0
    r, _ = O.filter_pack(pack, O.FLOAT, et, len(vals), L.FILTER_GREATER, [dbits(np.inf)])
    assert r == L.BITMAP_NONE
    r, _ = O.filter_pack(pack, O.FLOAT, et, len(vals), L.FILTER_GREATER_EQUAL, [dbits(-np.inf)])
    assert r == L.BITMAP_ALL


@pytest.mark.gpu
def test_gpu_filter_matches_oracle_for_every_op_and_type():
    import gpu_util as G
    rng = np.random.default_rng(4242)
    cases, preds = [], []
    for n in (65536, 12345):
        cols = make_columns(n, rng)
        for dt in (O.INT, O.LONG, O.FLOAT, O.DOUBLE, O.STRING):
            rows = cols[dt]
            data = O.rows_to_bytes(rows, dt)
            pack, et = O.compress_pack(data, dt, n, True)
            plist = predicates_for(dt, list(rows) if dt == O.STRING else rows.tolist())
            if dt in (O.FLOAT, O.DOUBLE):
                plist.append((L.FILTER_EQUAL, [dbits(float("nan"))], None))
                plist.append((L.FILTER_GREATER_EQUAL, [dbits(-0.0)], None))
            if dt == O.STRING:
                plist += [(L.FILTER_LIKE, [pat], None) for pat in LIKE_PATTERNS]
            for op, keys, _ in plist:
                cases.append(G.Case(pack, dt, et, n, len(data)))
                preds.append((op, keys))
        wild, _ = like_column(n, rng)
        wdata = O.rows_to_bytes(wild, O.STRING)
        wpack, wet = O.compress_pack(wdata, O.STRING, n, True)
        for pat in LIKE_PATTERNS + [b"a%", b"%b", b"\\%%", b"\\\\%", b"_\\__", b"%a%b%a%b%\\%%_"]:
            cases.append(G.Case(wpack, O.STRING, wet, n, len(wdata)))
            preds.append((L.FILTER_LIKE, [pat]))
    # one entry dictionaries (ALL shortcut of equal), a DELTA pack and a corrupted pack
    one = np.full(300, 42, dtype="<i4")
    p1, e1 = O.compress_pack(O.rows_to_bytes(one, O.INT), O.INT, 300, True)
    cases.append(G.Case(p1, O.INT, e1, 300, 1200)); preds.append((L.FILTER_EQUAL, [42]))
    cases.append(G.Case(p1, O.INT, e1, 300, 1200)); preds.append((L.FILTER_IN, [42, 43]))
    cases.append(G.Case(p1, O.INT, e1, 300, 1200)); preds.append((L.FILTER_LIKE, [b"4%"]))  # numeric column: rejected
    seq = np.arange(70000, 70000 + 40000, dtype="<i4")
    p2 = O.compress_pack_as(O.rows_to_bytes(seq, O.INT), O.INT, 40000, O.DELTA)
    cases.append(G.Case(p2, O.INT, O.DELTA, 40000, 160000)); preds.append((L.FILTER_EQUAL, [70001]))
    bad = cases[0].pack.copy(); bad[4:8] = 0xFF
    cases.append(G.Case(bad, cases[0].data_type, cases[0].encode_type, cases[0].n, cases[0].data_size)); preds.append(preds[0])

    descs, arena, _ = G.build_arena(cases)
    plan = L.Plan(0, descs, 64 << 20)  # small scratch budget: several groups
    values, offs = L.filter_keys([k for _, k in preds], [c.data_type == O.STRING or op == L.FILTER_LIKE for c, (op, _) in zip(cases, preds)])
    filters = [L.FilterDesc(op, len(k), o) for (op, k), o in zip(preds, offs)]
    d_in = L.DeviceBuffer(0, len(arena))
    d_bm = L.DeviceBuffer(0, L.BITMAP_BYTES * len(cases))
    try:
        d_in.upload(arena)
        d_bm.upload(np.full(L.BITMAP_BYTES * len(cases), 0xEE, dtype=np.uint8))
        res, nbad = L.plan_run_filter(plan, d_in.ptr, len(arena), filters, values, d_bm.ptr, L.BITMAP_BYTES * len(cases))
        host = d_bm.download().view("<u8").reshape(len(cases), 1024)
        kinds = set()
        for i, (c, (op, keys)) in enumerate(zip(cases, preds)):
            okeys = [0] if (op == L.FILTER_LIKE and c.data_type != O.STRING) else keys
            want, bm = O.filter_pack(c.pack, c.data_type, c.encode_type, c.n, op, okeys)
            if i == len(cases) - 1:
                assert res[i] < 0  # the corrupted pack: some decode error
                continue
            assert res[i] == want, (i, c.data_type, op, keys[:3], res[i], want)
            kinds.add(int(want))
            if want == L.BITMAP_SOME:
                assert np.array_equal(host[i], bm), (i, c.data_type, op)
            elif want == L.BITMAP_ALL:
                assert np.array_equal(host[i], bitmap_of(np.ones(c.n, dtype=bool))), i
            else:
                assert not host[i].any(), i
        assert kinds >= {L.BITMAP_NONE, L.BITMAP_ALL, L.BITMAP_SOME, -100, -102, O.DATA_IMPROPER}
        assert nbad == sum(1 for r in res if r < 0)
    finally:
        plan.destroy()
        d_in.free()
        d_bm.free()
