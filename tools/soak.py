#!/usr/bin/env python3
"""Differential soak run, GPU decoder against the CPU oracle, on random packs of random shapes (not part of the test suite:
the suite's seeds are fixed, this one walks through fresh seeds for as long as it is given).

usage: python tools/soak.py [--seconds 240] [--seed 1] [--batch 600]
Every batch: random data types, row counts (1..65536, biased towards the edges), value distributions (few / many distinct
values, runs of random lengths, monotone sequences, extreme values, NaN payloads; strings with shared prefixes, empty strings,
long strings, every byte value), encoded by the oracle with the reference's codec choice or a forced codec, decoded in one plan
through the C ABI and compared byte for byte; a share of the packs is mutated (truncation, byte flips) and must either fail or
decode to what the oracle decodes.  Prints one line per batch and a summary; exits non-zero on the first difference.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from gpu_util import Case, run_batch  # noqa: E402

NUM = {O.INT: "<i4", O.LONG: "<i8", O.FLOAT: "<f4", O.DOUBLE: "<f8"}


def pick_n(r):
    k = int(r.integers(0, 10))
    if k == 0:
        return int(r.integers(1, 10))
    if k == 1:
        return int(r.choice([63, 64, 65, 127, 128, 129, 1023, 1024, 1025, 65535, 65536]))
    if k == 2:
        return 65536
    return int(r.integers(1, 65537))


def runs_of(r, values, n):
    """values repeated in runs of random lengths (short, exactly 7 / 8 / 9, long)"""
    out = np.empty(n, dtype=values.dtype)
    i = 0
    while i < n:
        ln = int(r.choice([1, 2, 7, 8, 9, 30, 100, 1000, 5000], p=[.2, .1, .1, .15, .1, .15, .1, .07, .03]))
        out[i:i + ln] = values[int(r.integers(0, len(values)))]
        i += ln
    return out


def numbers(r, dt, n):
    t = NUM[dt]
    bits = 32 if dt in (O.INT, O.FLOAT) else 64
    it = "<i4" if bits == 32 else "<i8"
    kind = int(r.integers(0, 8))
    if kind == 0:  # few distinct values
        card = int(r.choice([1, 2, 3, 16, 64, 65, 300, 1000]))
        vals = r.integers(-(1 << (bits - 2)), 1 << (bits - 2), card).astype(it)
        rows = vals[r.integers(0, card, n)]
    elif kind == 1:  # runs
        card = int(r.choice([1, 2, 16, 200]))
        rows = runs_of(r, r.integers(-1000, 1000, card).astype(it), n)
    elif kind == 2:  # monotone with a random step range
        step = int(r.choice([1, 2, 4, 100, 2001, 1 << 20]))
        rows = (np.cumsum(r.integers(0, step, n)) + int(r.integers(-(1 << 30), 1 << 30))).astype(it)
    elif kind == 3:  # full range
        rows = r.integers(-(1 << (bits - 1)), (1 << (bits - 1)) - 1, n, endpoint=True).astype(it)
    elif kind == 4:  # extremes mixed in
        ext = np.array([-(1 << (bits - 1)), (1 << (bits - 1)) - 1, 0, -1, 1], dtype=it)
        rows = ext[r.integers(0, len(ext), n)]
    elif kind == 5:  # many distinct values of moderate width
        rows = r.integers(0, 1 << int(r.integers(1, bits - 1)), n).astype(it)
    elif kind == 6:  # k / 10 as in DictCodecTest
        card = int(r.choice([10, 1000, 10000, 50000]))
        k = r.integers(0, card, n)
        rows = (k / 10.0).astype(t).view(it) if dt in (O.FLOAT, O.DOUBLE) else (k * 7).astype(it)
    else:  # floats: special bit patterns
        if dt in (O.FLOAT, O.DOUBLE):
            sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.5, -2.25], dtype=t).view(it)
            rows = sp[r.integers(0, len(sp), n)].copy()
            nan_payload = r.integers(1, 1 << 20, n).astype(it)
            mask = r.random(n) < 0.05
            base = np.array([np.nan], dtype=t).view(it)[0]
            rows[mask] = base | nan_payload[mask]
        else:
            rows = r.integers(-5, 5, n).astype(it)
    return np.ascontiguousarray(rows).view(t)


def strings(r, n):
    kind = int(r.integers(0, 6))
    if kind == 0:
        words = [b"", b"a", b"ab", b"abc", b"www.example.com/", b"www.example.com/x", bytes(range(256))]
        return [words[i] for i in r.integers(0, len(words), n)]
    if kind == 1:  # many distinct, shared prefixes
        card = int(r.choice([10, 70, 3000, 20000]))
        dom = [b"www.s%d.com/%d" % (i % 37, i) for i in range(card)]
        return [dom[i] for i in r.integers(0, card, n)]
    def sliced(lens, lo, hi):
        buf = r.integers(lo, hi, int(lens.sum())).astype(np.uint8).tobytes()
        ends = np.cumsum(lens)
        return [buf[e - l:e] for e, l in zip(ends.tolist(), lens.tolist())]
    if kind == 2:  # unique short strings
        return sliced(r.integers(0, 25, n), 48, 123)
    if kind == 3:  # mostly empty
        return sliced(np.where(r.random(n) < 0.9, 0, r.integers(1, 17, n)), 97, 123)
    if kind == 4:  # runs of few values, one long string
        words = np.array([0, 1, 2, 3])
        ids = runs_of(r, words, n)
        table = [b"x" * int(r.integers(200, 3000)), b"", b"yy", b"x" * 15 + b"z"]
        return [table[i] for i in ids]
    n = min(n, 4000)  # arbitrary bytes, longer strings
    return sliced(r.integers(0, 200, n), 0, 256)


def make_case(r, idx):
    dt = int(r.choice([O.INT, O.LONG, O.FLOAT, O.DOUBLE, O.STRING], p=[.25, .2, .15, .15, .25]))
    n = pick_n(r)
    rows = strings(r, n) if dt == O.STRING else numbers(r, dt, n)
    n = len(rows)
    data = O.rows_to_bytes(rows, dt)
    forced = int(r.integers(0, 4))
    try:
        if forced == 0:
            pack, et = O.compress_pack(data, dt, n, bool(r.integers(0, 2)))
        else:
            et = [O.DICT_COMPRESS, O.DELTA, O.SIMPLE_DICT][forced - 1]
            pack = O.compress_pack_as(data, dt, n, et)
    except RuntimeError:  # the forced codec does not fit its buffer for this data: take the reference's choice
        pack, et = O.compress_pack(data, dt, n, True)
    dec, consumed = O.decompress_pack(pack, dt, et, n, len(data))
    # the expectation is what the oracle decodes, not the input: a dictionary keeps one NaN for all NaN payloads
    # (Float.compare / Double.compare order the entries), so such rows do not survive the reference's own round trip
    if consumed != len(pack):
        raise SystemExit("oracle could not decode its own pack: case %d (type %d enc %d n %d)" % (idx, dt, et, n))
    case = Case(pack, dt, et, n, len(data), dec, "case %d type %d enc %d n %d" % (idx, dt, et, n))
    if r.random() < 0.15 and len(pack) > 12:  # a mutated twin: must fail, or decode to what the oracle decodes
        q = pack.copy()
        if r.random() < 0.3:
            q = q[:len(q) - int(r.integers(1, min(len(q) - 4, 40)))]
        else:
            for _ in range(int(r.integers(1, 4))):
                q[int(r.integers(0, min(len(q), 2048)))] ^= 1 << int(r.integers(0, 8))
        d2, c2 = O.decompress_pack(q, dt, et, n, len(data))
        return [case, Case(q, dt, et, n, len(data), d2 if c2 == len(q) else None, case.name + " mutated")]
    return [case]


def encode_soak(args):
    """the device encoder against the oracle's encoder: same bytes, same codec, same DataPackNode statistics"""
    from indexr_amd import lib as L
    t0 = time.time()
    total = rows_total = 0
    batch = 0
    while time.time() - t0 < args.seconds:
        r = np.random.default_rng([args.seed, 7777, batch])
        cases, descs, parts, src, dst = [], [], [], 0, 0
        for i in range(args.batch):
            dt = int(r.choice([O.INT, O.LONG, O.FLOAT, O.DOUBLE, O.STRING], p=[.25, .2, .15, .15, .25]))
            n = pick_n(r)
            rows = strings(r, n) if dt == O.STRING else numbers(r, dt, n)
            idx = bool(r.integers(0, 2))
            data = O.rows_to_bytes(rows, dt)
            cases.append((dt, idx, len(rows), data))
            parts.append(data)
            parts.append(np.zeros(-len(data) % 16, dtype=np.uint8))
            descs.append(L.EncodeDesc(src, dst, dt, int(idx), len(rows), len(data), 0, 0, 0, 0, 0))
            src += len(data) + (-len(data) % 16)
            dst += (L.lib.ixr_compress_bound(len(rows), len(data)) + 15) & ~15
        arena = np.concatenate(parts)
        d_data, d_out = L.DeviceBuffer(0, len(arena)), L.DeviceBuffer(0, dst)
        try:
            d_data.upload(arena)
            arr, bad = L.encode_run(descs, d_data.ptr, len(arena), d_out.ptr, dst)
            host = d_out.download()
        finally:
            d_data.free()
            d_out.free()
        for k, ((dt, idx, n, data), a) in enumerate(zip(cases, arr)):
            want, et = O.compress_pack(data, dt, n, idx)
            got = host[a.dst_off:a.dst_off + max(a.pack_size, 0)]
            if a.pack_size != len(want) or a.encode_type != et or got.tobytes() != want.tobytes():
                raise SystemExit("ENCODER DIFFERENCE seed %d batch %d case %d type %d indexed %d n %d: size %d vs %d, codec %d vs %d"
                                 % (args.seed, batch, k, dt, idx, n, a.pack_size, len(want), a.encode_type, et))
            if dt != O.STRING:
                mn, mx, sm = O.pack_stats(data, dt, n)
                nan_sum = dt in (O.FLOAT, O.DOUBLE) and np.isnan(np.array([sm, a.sum], dtype="<i8").view("<f8")).all()
                if (a.min_value, a.max_value) != (mn, mx) or not (a.sum == sm or nan_sum):
                    raise SystemExit("STATISTICS DIFFERENCE seed %d batch %d case %d type %d n %d" % (args.seed, batch, k, dt, n))
            rows_total += n
        total += len(cases)
        batch += 1
        print("encode batch %d: %d packs so far, %d rows, %.0f s" % (batch, total, rows_total, time.time() - t0), flush=True)
    print("encode soak ok: seed %d, %d batches, %d packs, %d rows, no difference" % (args.seed, batch, total, rows_total))


def random_predicate(r, dt, rows):
    """(op, keys): keys taken from the rows, next to them, or from nowhere; LIKE patterns cut out of a row"""
    from indexr_amd import lib as L
    n = len(rows)

    def key():
        v = rows[int(r.integers(0, n))]
        if dt == O.STRING:
            k = int(r.integers(0, 4))
            return v if k else (v + b"~" if k == 1 else v[:max(0, len(v) - 1)])
        if dt in (O.FLOAT, O.DOUBLE):
            f = float(v) + (0.0 if r.random() < 0.7 else 0.03125)
            return int(np.array([f], dtype="<f8").view("<i8")[0])
        return max(-(1 << 63), min((1 << 63) - 1, int(v) + (0 if r.random() < 0.7 else int(r.integers(-3, 4)))))
    ops = [L.FILTER_EQUAL, L.FILTER_IN, L.FILTER_GREATER, L.FILTER_GREATER_EQUAL, L.FILTER_BETWEEN] + ([L.FILTER_LIKE] if dt == O.STRING else [])
    op = int(r.choice(ops))
    if op == L.FILTER_IN:
        return op, [key() for _ in range(int(r.choice([1, 2, 5, 70])))]
    if op == L.FILTER_BETWEEN:
        a, b = key(), key()
        if dt in (O.FLOAT, O.DOUBLE):  # raw bits do not order like the values: order by value
            fa, fb = np.array([a, b], dtype="<i8").view("<f8")
            if fa > fb:
                a, b = b, a
        elif a > b and r.random() < 0.9:
            a, b = b, a
        return op, [a, b]
    if op == L.FILTER_LIKE:
        v = rows[int(r.integers(0, n))]
        cut = int(r.integers(0, len(v) + 1))
        pat = [v[:cut] + b"%", b"%" + v[cut:], v[:cut] + b"_" + v[cut + 1:], b"%" + v[cut:cut + 3] + b"%", v][int(r.integers(0, 5))]
        return op, [pat]
    return op, [key()]


def filter_soak(args):
    """predicates on dictionary ids, on decoded rows, and the rows the bitmaps pick, against the oracle"""
    from indexr_amd import lib as L
    from gpu_util import build_arena
    t0 = time.time()
    total = rows_total = 0
    batch = 0
    kinds = {}
    while time.time() - t0 < args.seconds:
        r = np.random.default_rng([args.seed, 4242, batch])
        cases, preds = [], []
        for i in range(args.batch):
            dt = int(r.choice([O.INT, O.LONG, O.FLOAT, O.DOUBLE, O.STRING], p=[.25, .2, .15, .15, .25]))
            n = min(pick_n(r), args.max_rows)
            rows = strings(r, n) if dt == O.STRING else numbers(r, dt, n)
            if dt in (O.FLOAT, O.DOUBLE):
                rows = np.where(np.isnan(rows), 1.25, rows).astype(NUM[dt])  # binary search over NaN entries: not exercised here
            if dt == O.STRING and any(b"\\" in v or b"%" in v or b"_" in v for v in rows[:50]):
                rows = [v.replace(b"\\", b"/").replace(b"%", b"p").replace(b"_", b"u") for v in rows]
            n = len(rows)
            data = O.rows_to_bytes(rows, dt)
            pack, et = O.compress_pack(data, dt, n, True)
            dec, consumed = O.decompress_pack(pack, dt, et, n, len(data))  # what the pack holds (-0.0 may have become 0.0)
            if consumed != len(pack):
                raise SystemExit("oracle could not decode its own pack")
            cases.append(Case(pack, dt, et, n, len(data), dec, "case %d type %d enc %d n %d" % (i, dt, et, n)))
            preds.append(random_predicate(r, dt, rows))
        descs, arena, out_bytes = build_arena(cases)
        plan = L.Plan(0, descs)
        bm_bytes = L.BITMAP_BYTES * len(cases)
        d_in, d_out, d_bm = L.DeviceBuffer(0, len(arena)), L.DeviceBuffer(0, out_bytes), L.DeviceBuffer(0, bm_bytes)
        sel, dst = [], 0
        for i, d in enumerate(descs):
            sd = L.SelectDesc(d.out_off, dst, i * L.BITMAP_BYTES, d.data_type, d.obj_count, d.data_size, 0)
            sel.append(sd)
            dst += (int(L.lib.ixr_select_bound(sd)) + 255) // 256 * 256
        d_sel = L.DeviceBuffer(0, dst)
        try:
            d_in.upload(arena)
            values, voffs = L.filter_keys([k for _, k in preds], [c.data_type == O.STRING for c in cases])
            fd = [L.FilterDesc(op, len(k), vo) for (op, k), vo in zip(preds, voffs)]
            res, _ = L.plan_run_filter(plan, d_in.ptr, len(arena), fd, values, d_bm.ptr, bm_bytes)
            bm_dict = d_bm.download().view("<u8").reshape(len(cases), 1024).copy()
            plan.run(d_in.ptr, len(arena), d_out.ptr, out_bytes)
            sdescs = [L.ScanFilterDesc(d.out_off, vo, d.data_type, d.obj_count, d.data_size, op, len(k), 0) for d, (op, k), vo in zip(descs, preds, voffs)]
            res2, _ = L.scan_filter_run(sdescs, d_out.ptr, out_bytes, values, d_bm.ptr, bm_bytes)
            bm_scan = d_bm.download().view("<u8").reshape(len(cases), 1024)
            cnt, nbad = L.select_run(sel, d_out.ptr, out_bytes, d_bm.ptr, bm_bytes, d_sel.ptr, dst)
            hsel = d_sel.download()
        finally:
            plan.destroy()
            d_in.free(); d_out.free(); d_bm.free(); d_sel.free()
        for i, (c, (op, keys)) in enumerate(zip(cases, preds)):
            where = "seed %d batch %d %s op %d" % (args.seed, batch, c.name, op)
            want, bm = O.filter_pack(c.pack, c.data_type, c.encode_type, c.n, op, keys)
            if int(res[i]) != want or (want == L.BITMAP_SOME and not (bm_dict[i] == bm).all()):
                raise SystemExit("FILTER DIFFERENCE %s: %d vs %d" % (where, int(res[i]), want))
            kinds[want] = kinds.get(want, 0) + 1
            want2, bm2 = O.scan_filter_pack(c.expect, c.data_type, c.n, op, keys)
            if int(res2[i]) != want2 or not (bm_scan[i] == bm2).all():
                raise SystemExit("SCAN FILTER DIFFERENCE %s: %d vs %d" % (where, int(res2[i]), want2))
            if want >= 0 and want2 >= 0:  # the two indexes agree on the rows (ALL on one side is every row)
                full = np.zeros(65536, dtype=bool); full[:c.n] = True
                rows_a = np.unpackbits(bm.view(np.uint8), bitorder="little").astype(bool) if want == L.BITMAP_SOME else (full if want == L.BITMAP_ALL else ~full & False)
                rows_b = np.unpackbits(bm2.view(np.uint8), bitorder="little").astype(bool)
                if not (rows_a[:c.n] == rows_b[:c.n]).all():  # not a decoder matter: the two reference classes read a key differently
                    print("note: ExtIndex_DictBits and ExtIndex_SimpleBits pick different rows: %s keys %s" % (where, keys[:3]), flush=True)
            wc, ids, vals = O.select_pack(c.expect, c.data_type, c.n, bm2)
            if int(cnt[i]) != wc:
                raise SystemExit("SELECT COUNT DIFFERENCE %s: %d vs %d" % (where, int(cnt[i]), wc))
            n, base = c.n, sel[i].dst_off
            if not (hsel[base:base + 4 * n].view("<i4")[:wc] == ids).all():
                raise SystemExit("SELECT ROW DIFFERENCE %s" % where)
            if c.data_type == O.STRING:
                ao = hsel[base + 4 * n:base + 8 * n].view("<i4")[:wc]
                al = hsel[base + 8 * n:base + 12 * n].view("<i4")[:wc]
                body = hsel[base + 12 * n:base + int(L.lib.ixr_select_bound(sel[i]))].tobytes()
                ok = [body[ao[k]:ao[k] + al[k]] for k in range(wc)] == vals
            else:
                w = vals.dtype.itemsize
                vo = base + ((4 * n + 7) & ~7)
                ok = hsel[vo:vo + w * wc].tobytes() == vals.tobytes()
            if not ok:
                raise SystemExit("SELECT VALUE DIFFERENCE %s" % where)
            rows_total += c.n
        total += len(cases)
        batch += 1
        print("filter batch %d: %d packs so far, %d rows, results %s, %.0f s" % (batch, total, rows_total, dict(sorted(kinds.items())), time.time() - t0), flush=True)
    print("filter soak ok: seed %d, %d batches, %d packs, %d rows, results %s, no difference" % (args.seed, batch, total, rows_total, dict(sorted(kinds.items()))))


def aux_soak(args):
    """the DictStruct batch form and the engine hand-back against the oracle"""
    from indexr_amd import lib as L
    from gpu_util import build_arena
    t0 = time.time()
    total = rows_total = 0
    batch = 0
    while time.time() - t0 < args.seconds:
        r = np.random.default_rng([args.seed, 999, batch])
        cases = []
        for i in range(args.batch):
            dt = int(r.choice([O.INT, O.LONG, O.FLOAT, O.DOUBLE, O.STRING], p=[.2, .3, .1, .15, .25]))
            n = min(pick_n(r), 30000)
            rows = strings(r, n) if dt == O.STRING else numbers(r, dt, n)
            n = len(rows)
            data = O.rows_to_bytes(rows, dt)
            pack, et = O.compress_pack(data, dt, n, bool(r.integers(0, 2)))
            dec, consumed = O.decompress_pack(pack, dt, et, n, len(data))
            cases.append(Case(pack, dt, et, n, len(data), dec, "case %d type %d enc %d n %d" % (i, dt, et, n)))
        descs, arena, out_bytes = build_arena(cases)
        ds_off, off = [], 0
        for c in cases:
            ds_off.append(off)
            off += (8 + c.data_size + 4 * c.n + 15) & ~15
        hb, hoff = [], 0
        for c, d in zip(cases, descs):
            sql = {O.INT: [L.SQL_INT], O.LONG: [L.SQL_BIGINT, L.SQL_DATE, L.SQL_DATETIME, L.SQL_TIME], O.FLOAT: [L.SQL_FLOAT],
                   O.DOUBLE: [L.SQL_DOUBLE], O.STRING: [L.SQL_VARCHAR]}[c.data_type]
            ro = int(r.integers(0, c.n)) if r.random() < 0.5 else 0
            rc = int(r.integers(0, c.n - ro + 1)) if r.random() < 0.5 else c.n - ro
            h = L.HandbackDesc(d.out_off, hoff, c.n, c.data_size, int(r.choice(sql)), ro, rc, 0)
            hb.append(h)
            hoff += (int(L.lib.ixr_handback_bound(h)) + 15) & ~15
        plan = L.Plan(0, descs)
        d_in, d_out, d_ds, d_hb = L.DeviceBuffer(0, len(arena)), L.DeviceBuffer(0, out_bytes), L.DeviceBuffer(0, off + 16), L.DeviceBuffer(0, hoff + 16)
        try:
            d_in.upload(arena)
            plan.run_dict_structs(d_in.ptr, len(arena), d_ds.ptr, off + 16, ds_off)
            st_ds, _ = plan.status()
            hds = d_ds.download()
            plan.run(d_in.ptr, len(arena), d_out.ptr, out_bytes)
            st, nbad = plan.status()
            status, _ = L.handback_run(hb, d_out.ptr, out_bytes, d_hb.ptr, hoff + 16)
            hhb = d_hb.download()
        finally:
            plan.destroy()
            d_in.free(); d_out.free(); d_ds.free(); d_hb.free()
        for i, (c, h) in enumerate(zip(cases, hb)):
            where = "seed %d batch %d %s" % (args.seed, batch, c.name)
            if st[i] != len(c.pack) or st_ds[i] != len(c.pack):
                raise SystemExit("STATUS %s: %d / %d vs %d" % (where, st[i], st_ds[i], len(c.pack)))
            if c.encode_type == O.DICT_COMPRESS:
                want, consumed = O.get_dict_struct(c.pack, c.data_type, c.n)
                if hds[ds_off[i]:ds_off[i] + len(want)].tobytes() != want.tobytes():
                    raise SystemExit("DICT STRUCT DIFFERENCE %s" % where)
            want, rr = O.handback(h.sql_type, c.n, c.expect, h.row_offset, h.row_count)
            if int(status[i]) != rr or (rr > 0 and hhb[h.dst_off:h.dst_off + rr].tobytes() != want.tobytes()):
                raise SystemExit("HAND-BACK DIFFERENCE %s sql %d rows [%d, +%d): %d vs %d" % (where, h.sql_type, h.row_offset, h.row_count, int(status[i]), rr))
            rows_total += c.n
        total += len(cases)
        batch += 1
        print("aux batch %d: %d packs so far, %d rows, %.0f s" % (batch, total, rows_total, time.time() - t0), flush=True)
    print("DictStruct / hand-back soak ok: seed %d, %d batches, %d packs, %d rows, no difference" % (args.seed, batch, total, rows_total))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--batch", type=int, default=600)
    ap.add_argument("--scratch-mib", type=int, default=0, help="decoder soak: scratch budget of the plan (0 = default); a small one splits a batch into many groups")
    ap.add_argument("--max-rows", type=int, default=20000, help="filter soak: largest pack")
    ap.add_argument("--encode", action="store_true", help="soak the device encoder instead of the decoder")
    ap.add_argument("--aux", action="store_true", help="soak the DictStruct batch form and the engine hand-back instead of the decoder")
    ap.add_argument("--filter", action="store_true", help="soak the two predicate entry points and the row selection instead of the decoder")
    args = ap.parse_args()
    if args.encode:
        return encode_soak(args)
    if args.filter:
        return filter_soak(args)
    if args.aux:
        return aux_soak(args)
    t0 = time.time()
    total = mutated = failed_as_expected = rows = 0
    batch = 0
    while time.time() - t0 < args.seconds:
        r = np.random.default_rng([args.seed, batch])
        cases = []
        for i in range(args.batch):
            cases += make_case(r, i)
        outs, status, plan = run_batch(cases, scratch_budget=args.scratch_mib << 20)
        plan.destroy()
        for c, out, st in zip(cases, outs, status):
            clean = not c.name.endswith("mutated")
            if clean:
                if st != len(c.pack) or out.tobytes() != c.expect.tobytes():
                    raise SystemExit("DIFFERENCE seed %d batch %d %s: status %d packSize %d" % (args.seed, batch, c.name, st, len(c.pack)))
            else:
                mutated += 1
                if c.expect is None:
                    if st == len(c.pack) and st >= 0:
                        raise SystemExit("ACCEPTED A PACK THE ORACLE REJECTS seed %d batch %d %s" % (args.seed, batch, c.name))
                    failed_as_expected += 1
                elif st == len(c.pack):
                    if out.tobytes() != c.expect.tobytes():
                        raise SystemExit("DIFFERENCE (mutated) seed %d batch %d %s" % (args.seed, batch, c.name))
                elif st >= 0:
                    raise SystemExit("status neither an error nor packSize: seed %d batch %d %s: %d" % (args.seed, batch, c.name, st))
            rows += c.n
        total += len(cases)
        batch += 1
        print("batch %d: %d packs so far (%d mutated, %d of them rejected), %d rows, %.0f s" % (batch, total, mutated, failed_as_expected, rows, time.time() - t0), flush=True)
    print("soak ok: seed %d, %d batches, %d packs (%d mutated), %d rows, no difference" % (args.seed, batch, total, mutated, rows))


if __name__ == "__main__":
    main()
