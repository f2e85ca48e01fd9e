"""Golden vectors (tests/golden/cases.json, self-generated: see make_golden.py for provenance).

CPU tests: the oracle, the second spec codec and the product's host encoder all agree with the committed bytes.
GPU test: the HIP decoder reproduces the committed decoded bytes through the C ABI.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import pyref
from indexr_amd import lib as L

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "cases.json")) as fh:
    DOC = json.load(fh)
CASES = DOC["cases"]
IDS = [c["name"] for c in CASES]


def _b(c, key):
    return np.frombuffer(bytes.fromhex(c[key]), dtype=np.uint8)


def test_fixture_file_states_its_provenance():
    assert "self-generated" in DOC["provenance"]
    assert len(CASES) >= 25


@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_oracle_decodes_golden(c):
    data, pack = _b(c, "data"), _b(c, "pack")
    out, consumed = O.decompress_pack(pack, c["data_type"], c["encode_type"], c["n"], len(data))
    assert consumed == len(pack)  # PackCompressor_VLT.java:168
    assert np.array_equal(out, data)


@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_encoders_reproduce_golden(c):
    data, pack = _b(c, "input" if "input" in c else "data"), _b(c, "pack")
    if c["forced"]:
        again = O.compress_pack_as(data, c["data_type"], c["n"], c["encode_type"])
        assert np.array_equal(again, pack)
        return
    again, et = O.compress_pack(data, c["data_type"], c["n"], c["is_indexed"])
    assert et == c["encode_type"] and np.array_equal(again, pack)
    mine, et2 = L.compress_pack(data, c["data_type"], c["n"], c["is_indexed"])  # product's independent host encoder
    assert et2 == c["encode_type"] and np.array_equal(np.asarray(mine, dtype=np.uint8), pack)


@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_spec_codec_decodes_golden(c):
    data, pack = _b(c, "data"), _b(c, "pack")
    out, consumed = pyref.decode_pack(bytes(pack), c["n"], c["data_type"], c["encode_type"])
    assert consumed == len(pack)
    assert bytes(out) == data.tobytes()


@pytest.mark.gpu
def test_gpu_decodes_golden():
    import gpu_util as G
    cases = [G.Case(_b(c, "pack"), c["data_type"], c["encode_type"], c["n"], len(_b(c, "data")), expect=_b(c, "data"), name=c["name"])
             for c in CASES]
    G.check_cases(cases)
    # and one at a time through the Decoder-style single-pack entry point
    for c in CASES:
        data, pack = _b(c, "data"), _b(c, "pack")
        out, consumed = L.decode_pack(pack, c["data_type"], c["encode_type"], c["n"], len(data))
        assert consumed == len(pack), c["name"]
        assert np.array_equal(np.asarray(out, dtype=np.uint8), data), c["name"]


@pytest.mark.gpu
def test_gpu_decodes_golden_in_a_very_large_batch():
    """Batches of 16384 packs and more take the lane-per-stream framing walk instead of the LDS-window one; the golden
    packs repeated 530 times (16430 packs) must decode identically through it."""
    import gpu_util as G
    cases = []
    for rep in range(530):
        for c in CASES:
            cases.append(G.Case(_b(c, "pack"), c["data_type"], c["encode_type"], c["n"], len(_b(c, "data")), expect=_b(c, "data"),
                                name="%s#%d" % (c["name"], rep)))
    assert len(cases) >= 16384
    G.check_cases(cases)
