"""Numeric dictionary packs whose rows ask for the sorted dictionary's entries in order (columns that grow with the row
number).  The device does not materialise such a dictionary: k_delta leaves one record per 128-entry block and
k_dict_num decodes, per 1024-row tile, the stretch of entries the tile's ids span.  The choice looks at the first id of
every tile only, so the cases include id streams that pass that look and then misbehave inside the tiles, which must
still decode to the oracle's rows (DictStruct.decodeInts/Longs/Floats/Doubles, DictStruct.java:57-95)."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_util import check_cases
from indexr_amd import lib as L
from test_gpu_decode import oracle_case
from test_oracle import rng

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[1, 0], ids=["per tile", "through scratch"])
def _slicing(request):
    """every case runs with plans that decode such packs per tile (the library's default) and with plans that send their
    dictionaries through scratch like all others (the library's switch)"""
    L.debug_switch("slice", request.param)
    yield
    L.debug_switch("slice", 1)


def _growing(r, n, step, dtype):
    return (np.cumsum(r.integers(0, step, n)) + 1500000000000).astype(dtype)


def test_growing_columns_of_every_numeric_type():
    r = rng(2024)
    cases = []
    for n in (65536, 40000, 20000, 9000):
        cases.append(oracle_case(_growing(r, n, 2001, "<i8"), O.LONG, O.DICT_COMPRESS, name="long ms timestamps n%d" % n))
        cases.append(oracle_case(_growing(r, n, 3, "<i8"), O.LONG, O.DICT_COMPRESS, name="long steps 0..2 n%d" % n))
        cases.append(oracle_case((_growing(r, n, 50, "<i8") / 8.0).astype("<f8"), O.DOUBLE, O.DICT_COMPRESS, name="double growing n%d" % n))
    for n in (65536, 30000, 17000):
        cases.append(oracle_case((np.cumsum(r.integers(0, 5, n)) - 70000).astype("<i4"), O.INT, O.DICT_COMPRESS, name="int growing n%d" % n))
        cases.append(oracle_case((np.cumsum(r.integers(1, 4, n)) / 4.0).astype("<f4"), O.FLOAT, O.DICT_COMPRESS, name="float growing n%d" % n))
    # every value distinct: the ids are 0, 1, 2, ...; and a descending column (ids fall: not sliced)
    cases.append(oracle_case(np.arange(65536, dtype="<i8") * 977 - 5, O.LONG, O.DICT_COMPRESS, name="long distinct"))
    cases.append(oracle_case((np.arange(65536, dtype="<i8") * -977)[:50000], O.LONG, O.DICT_COMPRESS, name="long descending"))
    # the widest mini-block that is still decoded per tile (28 bits: two entries in one 8-byte load) and the first that is not
    for top in (1 << 28, 1 << 29, 1 << 40):
        cases.append(oracle_case((np.cumsum(r.integers(1, top, 65536, dtype=np.int64)) - (1 << 44)).astype("<i8"), O.LONG, O.DICT_COMPRESS,
                                 name="long steps below 2^%d" % (top.bit_length() - 1)))
    cases.append(oracle_case((np.cumsum(r.integers(1, 1 << 14, 65536)) - 5).astype("<i4"), O.INT, O.DICT_COMPRESS, name="int steps below 2^14"))
    # a last block of the dictionary with few entries, and a last tile with few rows
    cases.append(oracle_case(_growing(r, 65536 - 1000 + 3, 2001, "<i8"), O.LONG, O.DICT_COMPRESS, name="long ragged end"))
    cases.append(oracle_case(np.repeat(_growing(r, 8200, 2001, "<i8"), 7)[:57001], O.LONG, O.DICT_COMPRESS, name="long every value seven times"))
    # wrap-around: deltas that overflow int64 inside the dictionary's DELTA stream
    big = np.sort(r.integers(-(1 << 63), (1 << 63) - 1, 30000, endpoint=True).astype("<i8"))
    cases.append(oracle_case(big, O.LONG, O.DICT_COMPRESS, name="long sorted full range"))
    check_cases(cases)


def test_ids_that_rise_from_tile_to_tile_and_jump_inside():
    """The first row of every 1024-row tile continues the rising sequence, other rows hold values from anywhere in the
    column: one far row per tile, a whole stretch of far rows, rows that alternate between both ends of the dictionary."""
    r = rng(7)
    n = 65536
    base = _growing(r, n, 2001, "<i8")
    cases = []
    one_far = base.copy()
    for t in range(0, n, 1024):
        one_far[t + 5] = base[(t + 33000) % n]
    cases.append(oracle_case(one_far, O.LONG, O.DICT_COMPRESS, name="one far row per tile"))
    stretch = base.copy()
    stretch[3000:3900] = base[60000:60900]
    stretch[40000:40100] = base[100:200]
    cases.append(oracle_case(stretch, O.LONG, O.DICT_COMPRESS, name="stretches of far rows"))
    ends = base.copy()
    ends[1:1024:2] = base[-1]  # inside the first tile every other row is the column's last value
    ends[20481:21504:3] = base[0]
    cases.append(oracle_case(ends, O.LONG, O.DICT_COMPRESS, name="both ends of the dictionary inside a tile"))
    perm = base.copy()
    inner = np.arange(n).reshape(-1, 1024)
    for row in inner:
        r.shuffle(row[1:])  # every tile keeps its first row, the others are shuffled within the tile
    cases.append(oracle_case(base[inner.reshape(-1)], O.LONG, O.DICT_COMPRESS, name="shuffled inside every tile"))
    ints = (np.cumsum(r.integers(0, 4, n)) + 11).astype("<i4")
    ints[700::1024] = ints[-1]
    cases.append(oracle_case(ints, O.INT, O.DICT_COMPRESS, name="int: the last value once per tile"))
    check_cases(cases)


def test_dictionaries_that_are_not_taken_in_order_stay_on_the_gather_path():
    r = rng(9)
    cases = [oracle_case(r.permutation(np.arange(65536, dtype="<i8") * 1000003), O.LONG, O.DICT_COMPRESS, name="random permutation"),
             oracle_case(r.integers(0, 30000, 65536).astype("<i8") * 7, O.LONG, O.DICT_COMPRESS, name="30000 values at random"),
             oracle_case((r.integers(0, 50000, 65536) / 10.0).astype("<f8"), O.DOUBLE, O.DICT_COMPRESS, name="k / 10")]
    check_cases(cases)
