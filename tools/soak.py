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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--batch", type=int, default=600)
    args = ap.parse_args()
    t0 = time.time()
    total = mutated = failed_as_expected = rows = 0
    batch = 0
    while time.time() - t0 < args.seconds:
        r = np.random.default_rng([args.seed, batch])
        cases = []
        for i in range(args.batch):
            cases += make_case(r, i)
        outs, status, plan = run_batch(cases)
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
