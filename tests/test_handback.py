"""Engine hand-back (decoded pack -> Spark column vector contents, DefaultPackReader.java:164-241).
CPU: the oracle against an independent numpy / big-int statement of the Java expressions.  GPU: the device kernel against
the oracle on packs the GPU decoded itself."""
import numpy as np
import pytest

import oracle_lib as O
from indexr_amd import lib as L

MS_PER_DAY = 86400000
I64 = np.iinfo(np.int64)


def _java_div(a, b):  # Java long division truncates toward zero
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def _wrap(v, bits):
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


DATE_VALUES = [0, 1, -1, MS_PER_DAY - 1, MS_PER_DAY, -MS_PER_DAY, -MS_PER_DAY - 1, 1418083200000, -2170000000000,
               253370764800000, int(I64.max), int(I64.min), int(I64.max) // 1000 + 7, 5 * MS_PER_DAY + 12345]


def test_oracle_date_and_datetime_follow_the_java_expressions():
    v = np.array(DATE_VALUES, dtype="<i8")
    data = v.view(np.uint8)
    out, r = O.handback(6, len(v), data, 0, len(v))
    assert r == 4 * len(v)
    assert out.view("<i4").tolist() == [_wrap(_java_div(x, MS_PER_DAY), 32) for x in DATE_VALUES]
    out, r = O.handback(8, len(v), data, 2, 5)
    assert r == 40
    assert out.view("<i8").tolist() == [_wrap(x * 1000, 64) for x in DATE_VALUES[2:7]]


def test_oracle_numbers_strings_and_errors():
    ints = np.arange(-5, 20, dtype="<i4")
    out, r = O.handback(1, len(ints), ints.view(np.uint8), 3, 10)
    assert r == 40 and out.view("<i4").tolist() == ints[3:13].tolist()
    dbl = np.linspace(-1, 1, 9).astype("<f8")
    out, r = O.handback(4, 9, dbl.view(np.uint8), 0, 9)
    assert r == 72 and out.tobytes() == dbl.tobytes()
    rows = [b"windows", b"", b"mac", b"linux\x00x", b"", b"android"]
    data = O.rows_to_bytes(rows, O.STRING)
    out, r = O.handback(5, 6, data, 1, 4)
    sel = rows[1:5]
    assert r == 8 * 4 + sum(len(s) for s in sel)
    offs, lens = out[:16].view("<i4").tolist(), out[16:32].view("<i4").tolist()
    assert lens == [len(s) for s in sel] and offs == [0, 0, 3, 10]  # "", "mac", 7 bytes, ""
    assert out[32:].tobytes() == b"".join(sel)
    assert O.handback(7, len(ints), ints.view(np.uint8), 0, 1)[1] == -102       # TIME: "Unsupported date type"
    assert O.handback(1, len(ints), ints.view(np.uint8), 20, 10)[1] == -1000    # rows outside the pack
    assert O.handback(2, 9, dbl.view(np.uint8), 0, 9, out_limit=71)[1] == O.BUFFER_OVERFLOW


def test_handback_bound_and_argument_checks_need_no_gpu():
    d = L.HandbackDesc(0, 0, 100, 404 + 1234, L.SQL_VARCHAR, 10, 50, 0)
    assert L.lib.ixr_handback_bound(d) == (8 * 50 + 1234 + 15) // 16 * 16
    assert L.lib.ixr_handback_bound(L.HandbackDesc(0, 0, 100, 800, L.SQL_DATE, 0, 100, 0)) == 400
    assert L.lib.ixr_handback_bound(L.HandbackDesc(0, 0, 100, 800, L.SQL_DATETIME, 0, 33, 0)) == 272


@pytest.mark.gpu
def test_gpu_handback_matches_oracle_after_gpu_decode():
    rng = np.random.default_rng(99)
    n = 65536
    ms = (1_400_000_000_000 + np.cumsum(rng.integers(0, 5_000_000, n))).astype("<i8")
    ms[:len(DATE_VALUES)] = DATE_VALUES
    urls = [b"www.site%d.com/p/%d" % (i % 97, i * 31 % 1009) for i in range(3000)]
    cols = [  # (sql type, data type, rows)
        (L.SQL_INT, O.INT, rng.integers(-1000, 1000, n).astype("<i4")),
        (L.SQL_BIGINT, O.LONG, ms),
        (L.SQL_FLOAT, O.FLOAT, (rng.integers(0, 500, n) / 10).astype("<f4")),
        (L.SQL_DOUBLE, O.DOUBLE, (rng.integers(0, 5000, n) / 10).astype("<f8")),
        (L.SQL_DATE, O.LONG, ms),
        (L.SQL_DATETIME, O.LONG, ms),
        (L.SQL_VARCHAR, O.STRING, [urls[i] for i in rng.integers(0, len(urls), n)]),
        (L.SQL_TIME, O.INT, rng.integers(0, MS_PER_DAY, n).astype("<i4")),
    ]
    import gpu_util as G
    cases = []
    for sql, dt, rows in cols:
        data = O.rows_to_bytes(rows, dt)
        pack, et = O.compress_pack(data, dt, n, False)
        cases.append(G.Case(pack, dt, et, n, len(data), expect=data))
    descs, arena, out_bytes = G.build_arena(cases)
    plan = L.Plan(0, descs)
    d_in, d_dec = L.DeviceBuffer(0, len(arena)), L.DeviceBuffer(0, out_bytes)
    ranges = [(0, n), (0, n), (100, 1000), (n - 7, 7), (0, n), (5, n - 5), (12345, 40000), (0, 1)]
    hb, off = [], 0
    for (sql, dt, rows), d, (r0, rc) in zip(cols, descs, ranges):
        h = L.HandbackDesc(d.out_off, off, n, d.data_size, sql, r0, rc, 0)
        hb.append(h)
        off += int(L.lib.ixr_handback_bound(h))
    d_hb = L.DeviceBuffer(0, off + 16)
    try:
        d_in.upload(arena)
        plan.run(d_in.ptr, len(arena), d_dec.ptr, out_bytes)
        _, bad = plan.status()
        assert bad == 0
        d_hb.upload(np.full(off + 16, 0xCD, dtype=np.uint8))
        status, nbad = L.handback_run(hb, d_dec.ptr, out_bytes, d_hb.ptr, off + 16)
        host = d_hb.download()
        assert nbad == 1 and status[-1] == -102  # the TIME column is rejected like the reference rejects it
        for (sql, dt, rows), c, h, st in zip(cols[:-1], cases, hb, status):
            want, r = O.handback(sql, n, c.expect, h.row_offset, h.row_count)
            assert st == r == len(want), sql
            assert host[h.dst_off:h.dst_off + r].tobytes() == want.tobytes(), sql
            bound = int(L.lib.ixr_handback_bound(h))
            assert (host[h.dst_off + r:h.dst_off + bound] == 0xCD).all(), sql  # nothing written past the result
    finally:
        plan.destroy()
        d_in.free()
        d_dec.free()
        d_hb.free()
