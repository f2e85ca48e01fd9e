"""The parameter grids of the reference's own codec tests, run through the GPU batch path (C ABI) and compared bit for bit
with the CPU oracle on the same inputs.  The reference's tests draw from an unseeded java.util.Random; here every draw is
seeded, the ranges are the reference's:

  DictCodecTest.java:26-274   unique = 10, 1010, ..., 65010; 65536 values; ints, longs, floats (k/10), doubles (k/10),
                              strings; SimpleDictCodec and DictCompressCodec (the two dictionary pack encodings
                              PackCompressor_VLT.decompress dispatches on, PackCompressor_VLT.java:130-170)
  DeltaCodecTest.java:23-169  100 draws of valCount in [0, 65536); random walks with steps in [-50, 50) (ints) and
                              [-500, 500) (longs); 100 distinct prefixed strings
  PackingTest.java:31-80      counts 0..99 x bit widths 1..31 (ints) and 1..63 (longs)
  RLEPackingHybridCodecTest.java:18-69  100 draws of valCount, values below a random max, runs of 0..15

Expected bytes come from the oracle (a restatement of the Java codecs, itself not pinned to JVM output: the reference
ships no binary fixtures), so "equal" here means bit-exact against the restated oracle."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_util import Case, check_cases
from indexr_amd import lib as L
from test_oracle import rng

pytestmark = pytest.mark.gpu


def oracle_case(rows, data_type, encode_type, name):
    n = len(rows)
    data = O.rows_to_bytes(rows, data_type)
    pack = O.compress_pack_as(data, data_type, n, encode_type)
    dec, consumed = O.decompress_pack(pack, data_type, encode_type, n, len(data))
    assert consumed == len(pack) and (dec == data).all(), name
    return Case(pack, data_type, encode_type, n, len(data), dec, name)


def _dict_rows(r, data_type, unique, n=65536):
    """DictCodecTest's generators: abs(random) % unique, divided by 10 for floating point"""
    k = r.integers(0, unique, size=n)
    if data_type == O.INT:
        return k.astype("<i4")
    if data_type == O.LONG:
        return k.astype("<i8")
    if data_type == O.FLOAT:
        return (k.astype(np.float32) / np.float32(10)).astype("<f4")
    if data_type == O.DOUBLE:
        return (k / 10.0).astype("<f8")
    return [b"%d_unique_string" % int(v) for v in k]  # strings drawn from `unique` distinct values


@pytest.mark.parametrize("data_type", [O.INT, O.LONG, O.FLOAT, O.DOUBLE, O.STRING])
def test_dict_codec_grid(data_type):
    uniques = list(range(10, 65536, 1000))
    if data_type == O.STRING:
        uniques = uniques[::6]  # 11 of the 66 cardinalities: building 65536 Python strings per case is the slow part of this test
    cases = []
    for unique in uniques:
        rows = _dict_rows(rng(unique * 8 + data_type), data_type, unique)
        for et in (O.DICT_COMPRESS, O.SIMPLE_DICT):
            cases.append(oracle_case(rows, data_type, et, "type %d encoding %d unique %d" % (data_type, et, unique)))
    check_cases(cases)


def test_delta_codec_grid_random_counts():
    cases = []
    r = rng(20260101)
    for it in range(100):
        n = int(r.integers(0, 65536))
        if n == 0:
            continue  # valCount 0 is the single-pack entry point's case, see below
        start32 = int(r.integers(-2**31, 2**31))
        walk32 = (start32 + np.cumsum(r.integers(-50, 50, size=n))).astype(np.int64)
        cases.append(oracle_case((walk32 & 0xFFFFFFFF).astype(np.uint32).view("<i4"), O.INT, O.DELTA, "ints draw %d n %d" % (it, n)))
        start64 = int(r.integers(-2**63, 2**63 - 1, dtype=np.int64))
        with np.errstate(over="ignore"):
            walk64 = (np.int64(start64) + np.cumsum(r.integers(-500, 500, size=n)).astype(np.int64)).astype("<i8")
        cases.append(oracle_case(walk64, O.LONG, O.DELTA, "longs draw %d n %d" % (it, n)))
    prefixes = [b"www.", b"http://", b"indexr.io/", b"", b"a"]
    for it in range(100):
        n = int(r.integers(1, 65536)) if it % 4 == 0 else int(r.integers(1, 6000))
        distinct = [prefixes[int(r.integers(0, len(prefixes)))] + bytes(r.integers(65, 91, size=10, dtype=np.uint8)) for _ in range(100)]
        rows = [distinct[int(i)] for i in r.integers(0, 100, size=n)]
        cases.append(oracle_case(rows, O.STRING, O.DELTA, "strings draw %d n %d" % (it, n)))
    check_cases(cases)
    # valCount == 0: every codec returns 0 and touches nothing (DeltaCodec.java:100, DataPack.java:166-170)
    out, st = L.decode_pack(np.zeros(16, dtype=np.uint8), O.INT, O.DELTA, 0, 0)
    assert st == 0 and len(out) == 0


def test_packing_grid_counts_and_widths():
    """PackingTest's grid reaches the packers through the DELTA codec: a pack of `count` values whose deltas need exactly
    `width` bits exercises IntPacker / LongPacker at that width with that tail (count % 32 values in the last mini-block,
    count % 8 in its last group: DeltaCodec.java:60-98, IntPacker tail rule PackingUtil)."""
    cases = []
    r = rng(77)
    for count in range(1, 100):
        for width in range(1, 32):
            d = r.integers(0, 1 << width, size=count, dtype=np.int64)
            d[int(r.integers(0, count))] = (1 << width) - 1  # one delta uses the top bit ...
            d[int(r.integers(0, count))] = 0                 # ... and the block minimum is 0 (unless it is the same slot)
            vals = (np.cumsum(d) & 0xFFFFFFFF).astype(np.uint32).view("<i4")
            cases.append(oracle_case(vals, O.INT, O.DELTA, "ints count %d width %d" % (count, width)))
        for width in range(1, 64):
            d = r.integers(0, 1 << width, size=count, dtype=np.uint64)
            d[int(r.integers(0, count))] = (1 << width) - 1
            d[int(r.integers(0, count))] = 0
            with np.errstate(over="ignore"):
                vals = np.cumsum(d, dtype=np.uint64).view("<i8")
            cases.append(oracle_case(vals, O.LONG, O.DELTA, "longs count %d width %d" % (count, width)))
    check_cases(cases)


def test_rle_codec_grid_random_counts():
    cases = []
    r = rng(4242)
    for it in range(100):
        n = int(r.integers(1, 65536))
        mx = int(r.integers(1, 10000000))
        vals = np.empty(n, dtype=np.int64)
        i = 0
        while i < n:
            run = int(r.integers(0, 16))
            vals[i:i + run] = int(r.integers(0, mx))
            i += run
        # through the pack codec the RLE/bit-packing stream carries dictionary ids: the same run structure, ids for values
        cases.append(oracle_case(vals.astype("<i4"), O.INT, O.DICT_COMPRESS, "draw %d n %d max %d" % (it, n, mx)))
    check_cases(cases)
