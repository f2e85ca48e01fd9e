"""GPU parity tests aimed at the framing walkers and the kernels around them.

The id stream of a dictionary pack (RLEPackingHybridCodec.java:97-128) is walked 64 bytes at a time, every lane parsing a
header at its byte; a DELTA stream (DeltaCodec.java:100-141, 204-245) in rounds of guessed block starts between rounds of the
serial chain, out of a 16 KB window in small launches and a 4 KB window in large ones.
Every case here runs three ways -- a small launch, a launch of more than 1024 packs and the form of launches of thousands of
packs, a one-wave workgroup per stream -- and must give the oracle's bytes and status each time.  Bar: bit-exact.
"""
import os
import struct

import numpy as np
import pytest

import oracle_lib as O
import pyref as R
from gpu_util import Case, run_batch
from indexr_amd import lib as L
from test_gpu_decode import oracle_case
from test_oracle import rng

pytestmark = pytest.mark.gpu


def three_ways(cases):
    """cases: Case objects whose `expect` is the oracle's output, or None when the oracle rejects the pack."""
    filler = oracle_case(np.arange(300, dtype="<i4") // 7, L.INT, O.DICT_COMPRESS, name="filler")
    runs = [("small launch", cases, None), ("large launch", cases + [filler] * (1100 - min(len(cases), 1000)), None),
            ("a wave per stream", cases, "wave_walk_from")]
    for label, batch, switch in runs:
        if switch:
            L.debug_switch(switch, 1)
        try:
            outs, status, plan = run_batch(batch)
            plan.destroy()
        finally:
            if switch:
                L.debug_switch(switch, 0)
        for c, out, st in zip(batch, outs, status):
            if c.expect is None:
                assert st < 0, "%s, %s: the oracle rejects this pack, the GPU returned %d" % (label, c.name, st)
            else:
                assert st == len(c.pack), "%s, %s: status %d != packSize %d" % (label, c.name, st, len(c.pack))
                assert out.tobytes() == np.asarray(c.expect, np.uint8).tobytes(), "%s, %s: decoded bytes differ" % (label, c.name)


def judged_by_oracle(pack, data_type, encode_type, n, data_size, name):
    """A pack of unknown soundness: expected output = whatever the oracle makes of it."""
    pack = np.ascontiguousarray(pack, dtype=np.uint8)
    dec, r = O.decompress_pack(pack, data_type, encode_type, n, data_size)
    return Case(pack, data_type, encode_type, n, data_size, dec if r == len(pack) else None, name)


# ------------------------------------------------------------------ hand-built id streams
def uvarint(v, nbytes=0):
    """Unsigned varint of v, padded with continuation bytes to nbytes when given (readUnsignedVarInt accepts that)."""
    out = bytearray()
    R.put_uvarint(out, v)
    while len(out) < nbytes:
        out[-1] |= 0x80
        out.append(0)
    return bytes(out)


def int_dict_pack(entry_count, segments):
    """INT DICT_COMPRESS pack with a plain dictionary 0, 10, 20, ... and the given id stream.
    segments: ("run", count, id, header_bytes) or ("packed", [ids], header_bytes).  -> (pack bytes, n, rows)"""
    bw = max(entry_count - 1, 0).bit_length()
    bytew = (bw + 7) // 8
    body, rows = bytearray(), []
    for seg in segments:
        if seg[0] == "run":
            _, count, value, hb = seg
            body += uvarint(count << 1, hb) + int(value).to_bytes(4, "little")[:bytew]
            rows += [value] * count
        else:
            _, ids, hb = seg
            body += uvarint((len(ids) << 1) | 1, hb) + R.pack_bits(list(ids), bw)
            rows += list(ids)
    n = len(rows)
    entries = struct.pack("<%di" % entry_count, *[10 * i for i in range(entry_count)])
    ds_size = 4 + 4 * entry_count + 4 + 4 * n
    pack = struct.pack("<iiB", ds_size, entry_count, 0) + entries + struct.pack("<i", len(body)) + bytes(body)
    return np.frombuffer(pack, np.uint8), n, rows


def random_segments(r, entry_count, n, max_run, zero_every=0, pad_every=0):
    segs, rows, k = [], 0, 0
    while rows < n:
        k += 1
        hb = int(r.integers(2, 6)) if pad_every and k % pad_every == 0 else 0
        if zero_every and k % zero_every == 0:
            segs.append(("run", 0, int(r.integers(0, entry_count)), hb) if k % 2 else ("packed", [], hb))
            continue
        if r.integers(0, 2):
            c = min(int(r.integers(1, max_run + 1)), n - rows)
            segs.append(("run", c, int(r.integers(0, entry_count)), hb))
        else:
            c = min(int(r.integers(1, max_run + 1)), n - rows)
            segs.append(("packed", [int(x) for x in r.integers(0, entry_count, c)], hb))
        rows += c
    return segs


def test_hand_built_id_streams():
    r = rng(2024)
    cases = []

    def add(entry_count, segs, name):
        pack, n, rows = int_dict_pack(entry_count, segs)
        assert 0 < n <= 65536, name
        c = judged_by_oracle(pack, L.INT, L.ENC_DICT_COMPRESS, n, 4 * n, name)
        assert c.expect is not None, name  # these streams are sound
        assert (np.frombuffer(c.expect.tobytes(), "<i4") == 10 * np.array(rows)).all(), name
        cases.append(c)

    for ec in (3, 300, 40000):
        add(ec, random_segments(r, ec, 65536, 12), "short segments ec%d" % ec)
        add(ec, random_segments(r, ec, 65536, 40, zero_every=5), "zero-count segments ec%d" % ec)
        # (a dictionary with more entries than the pack has rows is refused by the device decoder: keep n >= ec)
        add(ec, random_segments(r, ec, max(30000, ec + 1), 70, pad_every=3), "padded headers ec%d" % ec)
        add(ec, random_segments(r, ec, 65536, 3000, zero_every=7, pad_every=4), "long and short ec%d" % ec)
    # a flood of empty segments before, between and after the rows
    add(7, [("packed", [], 0)] * 3000 + [("run", 2000, 3, 0)] + [("run", 0, 1, 0)] * 500 + [("packed", [1, 2, 3, 4, 5], 0)]
        + [("packed", [], 2)] * 200, "empty segments")
    # segment lengths around the lane-expanded / wave-expanded limit and the tile size, headers at every chunk phase
    for phase in range(0, 70, 7):
        segs = [("packed", [1] * phase, 0)] if phase else []
        for c in (31, 32, 33, 63, 64, 65, 1023, 1024, 1025, 8, 7, 1):
            segs += [("run", c, 2, 0), ("packed", [int(x) for x in r.integers(0, 5, c)], 0)]
        add(5, segs, "limits phase %d" % phase)
    # one segment for the whole pack, two- and three-byte headers
    add(1000, [("packed", [int(x) for x in r.integers(0, 1000, 65536)], 0)], "one packed segment")
    add(1000, [("run", 65536, 999, 0)], "one run")
    add(2, [("run", 8191, 1, 0), ("run", 8192, 0, 0), ("packed", [1, 0] * 4000, 0), ("run", 41153, 1, 5)], "multi-byte headers")
    three_ways(cases)


def test_broken_id_streams():
    r = rng(7)
    cases = []
    good, n, _ = int_dict_pack(300, random_segments(r, 300, 20000, 30))
    ids_at = 9 + 4 * 300  # the id stream's length word
    (body_len,) = struct.unpack_from("<i", good.tobytes(), ids_at)

    def variant(name, edit):
        p = bytearray(good.tobytes())
        edit(p)
        cases.append(judged_by_oracle(np.frombuffer(bytes(p), np.uint8), L.INT, L.ENC_DICT_COMPRESS, n, 4 * n, name))

    variant("sound", lambda p: None)
    variant("length word one short", lambda p: struct.pack_into("<i", p, ids_at, body_len - 1))
    variant("length word one long", lambda p: struct.pack_into("<i", p, ids_at, body_len + 1))
    variant("first header asks for too many rows", lambda p: p.__setitem__(ids_at + 4, 0xFE))
    variant("last byte is a continuation byte", lambda p: p.__setitem__(len(p) - 1, 0x80))
    for k in range(40):
        pos = ids_at + 4 + int(r.integers(0, body_len))
        bit = 1 << int(r.integers(0, 8))
        variant("flip %d@%d" % (k, pos), lambda p, pos=pos, bit=bit: p.__setitem__(pos, p[pos] ^ bit))
    for cut in (1, 2, 3, 17, 64, 65):
        cases.append(judged_by_oracle(good[:len(good) - cut], L.INT, L.ENC_DICT_COMPRESS, n, 4 * n, "cut %d" % cut))
    # Bit-packed segments of one length, a few hundred bytes apart (what the writer makes of ids without runs): the walker takes
    # them by guessed header positions, 64 a round.  Flips in and around the headers, an odd segment in the middle, cuts.
    segs = [("packed", [int(x) for x in r.integers(0, 300, 504)], 0) for _ in range(100)]
    segs[41] = ("packed", [int(x) for x in r.integers(0, 300, 496)], 0)
    segs[77] = ("run", 700, 5, 0)
    steady, n2, _ = int_dict_pack(300, segs)
    (body2,) = struct.unpack_from("<i", steady.tobytes(), ids_at)
    headers, at = [], ids_at + 4   # where each segment's header starts (9-bit ids, run values in 2 bytes)
    for seg in segs:
        headers.append(at)
        at += len(uvarint(seg[1] << 1)) + 2 if seg[0] == "run" else len(uvarint(len(seg[1]) << 1 | 1)) + (len(seg[1]) + 7) // 8 * 9
    assert at == ids_at + 4 + body2 and headers[1] - headers[0] == 569

    def variant2(name, edit):
        p = bytearray(steady.tobytes())
        edit(p)
        cases.append(judged_by_oracle(np.frombuffer(bytes(p), np.uint8), L.INT, L.ENC_DICT_COMPRESS, n2, 4 * n2, name))

    variant2("steady segments", lambda p: None)
    variant2("steady, length word one short", lambda p: struct.pack_into("<i", p, ids_at, body2 - 1))
    variant2("steady, length word one long", lambda p: struct.pack_into("<i", p, ids_at, body2 + 1))
    for k in (0, 1, 2, 7, 8, 9, 30, 40, 63, 64, 65, 98):
        hdr = headers[k]
        for bit in (0, 1, 3, 7):
            variant2("steady, header %d bit %d" % (k, bit), lambda p, hdr=hdr, bit=bit: p.__setitem__(hdr, p[hdr] ^ (1 << bit)))
        variant2("steady, byte behind header %d" % k, lambda p, hdr=hdr: p.__setitem__(hdr + 1, p[hdr + 1] ^ 0x80))
        variant2("steady, an id of segment %d" % k, lambda p, hdr=hdr: p.__setitem__(hdr + 20, p[hdr + 20] ^ 0x01))
    for cut in (1, 2, 9, 500, 568, 569, 570, 5000):
        cases.append(judged_by_oracle(steady[:len(steady) - cut], L.INT, L.ENC_DICT_COMPRESS, n2, 4 * n2, "steady cut %d" % cut))
    assert any(c.expect is None for c in cases) and any(c.expect is not None for c in cases)
    three_ways(cases)


# ------------------------------------------------------------------ DELTA streams with and without a steady block size
def test_delta_streams_of_steady_and_unsteady_block_size():
    r = rng(31)
    cases = []

    def longs(vals, name):
        cases.append(oracle_case(np.asarray(vals).astype("<i8"), L.LONG, O.DELTA, name=name))

    def ints(vals, name):
        cases.append(oracle_case((np.asarray(vals, dtype=np.int64) & 0xFFFFFFFF).astype(np.uint32).view("<i4"), L.INT, O.DELTA, name=name))

    for n in (129, 257, 2048, 2049, 8191, 65535, 65536):
        longs(np.cumsum(r.integers(0, 2001, n)) + 1500000000000, "steady 11 bit n%d" % n)
        ints(np.cumsum(r.integers(0, 4, n)), "steady 2 bit n%d" % n)
    steps = r.integers(0, 2001, 65536)
    steps[np.arange(37 * 128 + 5, 65536, 37 * 128)] = 1 << 40      # every 37th block is wider
    longs(np.cumsum(steps), "an odd block every 37")
    steps = r.integers(100, 2001, 65536)
    steps[(np.arange(65536) // 128) % 5 == 0] += 70000              # every 5th block: another width
    longs(np.cumsum(steps), "an odd block every 5")
    steps = r.integers(0, 1000, 65536) + np.where((np.arange(65536) // 128) % 3 == 0, 0, 64)  # block minima of one and two varint bytes
    longs(np.cumsum(steps), "varint lengths alternate")
    ints(r.integers(-2**31, 2**31, 65536), "random 32 bit")
    longs(r.integers(-2**63, 2**63 - 1, 65536, dtype=np.int64), "random 64 bit")
    longs(np.cumsum(r.integers(0, 1 << 50, 65536).astype(np.int64)), "50 bit steps")
    half = np.concatenate([np.cumsum(r.integers(0, 3, 30000)), np.cumsum(r.integers(0, 1 << 20, 35536)) + 100000])
    ints(half, "block size changes for good")
    dbl = (np.arange(65536) / 10.0).astype("<f8")
    cases.append(oracle_case(dbl, L.DOUBLE, O.DICT_COMPRESS, name="sorted doubles as a dictionary"))
    cases.append(oracle_case((np.cumsum(r.integers(0, 2001, 65536)) + 1500000000000).astype("<i8"), L.LONG, O.DICT_COMPRESS,
                             name="timestamps as a dictionary"))
    three_ways(cases)


def test_broken_delta_streams():
    r = rng(32)
    cases = []
    steady = oracle_case((np.cumsum(r.integers(0, 2001, 65536)) + 1500000000000).astype("<i8"), L.LONG, O.DELTA, name="steady")
    narrow = oracle_case(np.cumsum(r.integers(0, 4, 40000)).astype("<i4"), L.INT, O.DELTA, name="narrow")
    for base in (steady, narrow):
        cases.append(base)
        p = base.pack
        for k in range(30):
            q = p.copy()
            pos = int(r.integers(4, len(q)))
            q[pos] ^= 1 << int(r.integers(0, 8))
            cases.append(judged_by_oracle(q, base.data_type, base.encode_type, base.n, base.data_size, "%s flip%d@%d" % (base.name, k, pos)))
        for cut in (1, 5, 181, 2000):
            cases.append(judged_by_oracle(p[:len(p) - cut], base.data_type, base.encode_type, base.n, base.data_size, "%s cut %d" % (base.name, cut)))
        q = p.copy()
        struct.pack_into("<i", q, 0, struct.unpack_from("<i", q.tobytes(), 0)[0] - 3)  # the stream's length word
        cases.append(judged_by_oracle(q, base.data_type, base.encode_type, base.n, base.data_size, base.name + " short length word"))
    # Wide values: blocks of a kilobyte, whose framing bytes the guessed rounds fetch from global memory lane by lane.  Random
    # flips mostly land in the values, so the width bytes and block minima are also hit on purpose.
    for base in (oracle_case(r.integers(-2**63, 2**63 - 1, 65536, dtype=np.int64).astype("<i8"), L.LONG, O.DELTA, name="wide longs"),
                 oracle_case(r.integers(-2**31, 2**31, 30000).astype("<i4"), L.INT, O.DELTA, name="wide ints"),
                 oracle_case(np.cumsum(r.integers(0, 1 << 40, 50000).astype(np.int64)).astype("<i8"), L.LONG, O.DELTA, name="40 bit steps")):
        cases.append(base)
        p = base.pack
        W = 64 if base.data_type == L.LONG else 32
        framing = delta_framing_positions(p.tobytes(), base.n, W)
        assert len(framing) >= 5 * (base.n // 128)
        for k in range(40):
            q = p.copy()
            pos = framing[int(r.integers(0, len(framing)))] if k % 4 else int(r.integers(4, len(q)))
            q[pos] ^= 1 << int(r.integers(0, 8))
            cases.append(judged_by_oracle(q, base.data_type, base.encode_type, base.n, base.data_size, "%s flip%d@%d" % (base.name, k, pos)))
        for cut in (1, 7, 300, 5000):
            cases.append(judged_by_oracle(p[:len(p) - cut], base.data_type, base.encode_type, base.n, base.data_size, "%s cut %d" % (base.name, cut)))
    assert any(c.expect is None for c in cases)
    three_ways(cases)


def delta_framing_positions(pack, n, W):
    """Offsets of the framing bytes (block minima and width bytes) of a sound DELTA stream at the head of `pack`
    (DeltaCodec.java:100-141, 204-245: length word, first value, then per 128 values a varint and four mini-blocks)."""
    pos = 4
    while pack[pos] & 0x80:
        pos += 1
    pos += 1
    out, left = [], n
    while left > 0:
        while pack[pos] & 0x80:
            out.append(pos)
            pos += 1
        out.append(pos)
        pos += 1
        cnt = min(128, left)
        for m in range(4):
            if m * 32 >= cnt:
                break
            c = min(32, cnt - m * 32)
            out.append(pos)
            pos += 1 + ((c + 7) // 8) * min(pack[pos], W)
        left -= cnt
    assert pos == 4 + struct.unpack_from("<i", pack, 0)[0], "the walk must end where the length word says"
    return out


# ------------------------------------------------------------------ string dictionaries: length table and one-pass scan
def test_string_dictionaries_of_many_sizes_and_long_entries():
    r = rng(33)
    cases = []

    def words(m, longest=0):
        w = [b"w%06d" % i + bytes(r.integers(97, 123, int(r.integers(0, 6))).astype(np.uint8)) for i in range(m)]
        if longest:
            w[m // 2] = bytes(r.integers(97, 123, longest).astype(np.uint8))
        return w

    def pack_of(w, n, name, et=O.DICT_COMPRESS):
        rows = [w[i] for i in r.integers(0, len(w), n)]
        rows[:len(w)] = w
        cases.append(oracle_case(rows, L.STRING, et, name=name))

    for m in (1, 2, 1023, 1024, 1025, 2047, 2048, 2049, 4097):       # dictionary sizes around the scan's 1024-entry chunks
        pack_of(words(m), max(m, 3000), "%d entries" % m)
    pack_of(words(3000, longest=255), 20000, "longest entry 255 bytes")   # one byte per length still
    pack_of(words(3000, longest=256), 20000, "longest entry 256 bytes")   # two bytes per length
    pack_of(words(32768, longest=300), 65536, "32768 entries, two bytes per length: the table is full")
    pack_of(words(32769, longest=300), 65536, "32769 entries, two bytes per length: lengths gathered from memory")
    pack_of(words(65536), 65536, "65536 entries, one byte per length")
    pack_of(words(500, longest=70000), 2000, "an entry longer than 65535 bytes")
    pack_of(words(2000), 65536, "plain dictionary", et=O.SIMPLE_DICT)
    three_ways(cases)


def test_numeric_dictionaries_around_the_lds_limits():
    """Numeric dictionaries of 2 to 64 KB are looked up in LDS by a kernel of their own: entry counts either side of both
    limits, for 32- and 64-bit values, bit-packed and run-heavy id streams."""
    r = rng(34)
    cases = []
    for data_type, dt, counts in ((L.INT, "<i4", (511, 512, 513, 5000, 16383, 16384, 16385)), (L.FLOAT, "<f4", (512, 16384)),
                                  (L.LONG, "<i8", (255, 256, 257, 3000, 8191, 8192, 8193)), (L.DOUBLE, "<f8", (256, 8192))):
        for m in counts:
            vals = (np.arange(m) * 37 - 5000).astype(dt)
            rows = vals[r.integers(0, m, 65536)]
            rows[:m] = vals  # every entry is used
            cases.append(oracle_case(rows, data_type, O.DICT_COMPRESS, name="type%d %d entries" % (data_type, m)))
            runs = np.repeat(vals[r.integers(0, m, 3000)], r.integers(1, 40, 3000))[:60000]
            runs = np.concatenate([vals, runs])[:65536]
            cases.append(oracle_case(runs, data_type, O.DICT_COMPRESS, name="type%d %d entries, runs" % (data_type, m)))
            cases.append(oracle_case(rows[:max(m, 1500)], data_type, O.SIMPLE_DICT, name="type%d %d entries, plain" % (data_type, m)))
    three_ways(cases)
