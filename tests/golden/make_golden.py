#!/usr/bin/env python3
"""Generates tests/golden/cases.json: small packs with their decoded bytes.

Provenance: SELF-GENERATED.  The packs are produced by the CPU oracle's encoder (oracle/, a restatement of
PackCompressor_VLT.compress); the reference itself cannot run in this environment (no JVM), so these vectors freeze
today's bytes against regressions and give the GPU tests fixed inputs, they do not pin agreement with a JVM.
The inputs are the literal value lists of the reference's own tests (DataPackTest.java:108-169,196;
TestRows.java:39-44) plus small seeded cases for every codec branch.

usage: python tests/golden/make_golden.py        (rewrites cases.json next to this file)
"""
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

I32_MAX, I32_MIN = 2**31 - 1, -2**31
I64_MAX, I64_MIN = 2**63 - 1, -2**63
F32_MIN = struct.unpack("<f", struct.pack("<I", 1))[0]  # Float.MIN_VALUE


def cases():
    out = []

    def add(name, data_type, rows, is_indexed=False, force=None):
        data = O.rows_to_bytes(rows, data_type)
        n = len(rows)
        if force is None:
            pack, et = O.compress_pack(data, data_type, n, is_indexed)
        else:
            pack, et = O.compress_pack_as(data, data_type, n, force), force
        # "data" is what decoding the pack yields.  It equals the encoder's input except where the reference's dictionary
        # lookup is lossy by construction (entries ordered by Float/Double.compare but searched with plain < and >, so
        # -0.0 and NaN rows do not find their own entry: DictCodec.java:79-131); those cases keep the input separately.
        decoded, consumed = O.decompress_pack(pack, data_type, et, n, len(data))
        assert consumed == len(pack), name
        rec = {"name": name, "data_type": data_type, "n": n, "is_indexed": bool(is_indexed), "forced": force is not None,
               "encode_type": int(et), "data": decoded.tobytes().hex(), "pack": pack.tobytes().hex()}
        if not np.array_equal(decoded, data):
            assert "specials" in name, name
            rec["input"] = data.tobytes().hex()
        out.append(rec)

    # --- literal inputs of the reference's tests
    add("DataPackTest_ints", O.INT, [1, 2, 3, 5, 6, 7, 8, 9, -1111, 333, 0])
    add("DataPackTest_int_zero", O.INT, [0])
    add("DataPackTest_int_same", O.INT, [10, 10, 10, 10])
    add("DataPackTest_longs", O.LONG, [1, I64_MAX, 3, I64_MIN, 6, 7, 8, 9, 1111, 333, 0])
    add("DataPackTest_floats", O.FLOAT, np.array([1, 2, 3, 5, 6, 7, 8, 9, 1111.4666, 333, 0], dtype="<f4"))
    add("DataPackTest_float_pair", O.FLOAT, np.array([1.1, 1.1], dtype="<f4"))
    add("DataPackTest_doubles", O.DOUBLE, [1, 2, 3, 5, 4.9e-324, 7, 1.7976931348623157e308, 9, 1111, -333.4444444, 0])
    str_vals = [b"qwfwrfwe", b"121343%$#22342342", b"2211", b"sdfdsgkwkek00-\t\t\x05\x03\x01\bsdaasds", b""]
    add("DataPackTest_strings", O.STRING, str_vals)
    add("DataPackTest_strings_indexed", O.STRING, str_vals, is_indexed=True)
    add("DataPackTest_strings_all_empty", O.STRING, [b""] * 4)
    strs = [b"121343%$#22342342", b"#$%#%242rwef23r23dfsdf", b"", b"2211", b"sdfdsgkwkek00-\t\t\x05\x03\x01\bsdaasds", b"", b"323weewew",
            b"[][][][[]]"] + [b"windows", b"mac", b"linux", b"android"] * 3 + [b"qwfwrfwe"]
    add("DataPackTest_test_str", O.STRING, strs)
    # the four sample rows of TestRows.java:39-44 cycled, column by column
    add("TestRows_ints", O.INT, [89, 3, 14121, I32_MAX] * 8)
    add("TestRows_longs", O.LONG, [222222, I64_MAX, 99, 11] * 8)
    add("TestRows_floats", O.FLOAT, np.array([4.5, 4.5, 2.5, F32_MIN] * 8, dtype="<f4"))
    add("TestRows_doubles", O.DOUBLE, [9.1, 9.199, 11.1, 1.51] * 8)
    add("TestRows_strings", O.STRING, [b"windows", b"mac", b"linux", b"android"] * 8)
    # --- every codec branch, small and seeded
    rng = np.random.default_rng(20240611)
    add("int_delta_monotone", O.INT, np.cumsum(rng.integers(0, 4, 700)).astype("<i4"))
    add("int_delta_wrapping", O.INT, rng.integers(I32_MIN, I32_MAX, 300, dtype=np.int64).astype("<i4"), force=O.DELTA)
    add("int_dict_small", O.INT, rng.integers(0, 50, 500).astype("<i4") * 1000003 % 65521)
    add("int_dict_runs", O.INT, np.repeat(rng.integers(0, 9, 40), rng.integers(1, 40, 40)).astype("<i4"), is_indexed=True)
    add("int_simple_dict", O.INT, rng.integers(0, 30, 400).astype("<i4"), force=O.SIMPLE_DICT)
    add("long_dict_delta_entries", O.LONG, (1_700_000_000_000 + np.cumsum(rng.integers(0, 2000, 900))).astype("<i8"))
    add("long_delta", O.LONG, rng.integers(I64_MIN, I64_MAX, 200, dtype=np.int64), force=O.DELTA)
    add("float_dict", O.FLOAT, (rng.integers(0, 300, 600) / 10.0).astype("<f4"))
    add("float_specials", O.FLOAT, np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.5, -0.0, np.nan, 0.0], dtype="<f4"))
    add("double_dict", O.DOUBLE, (rng.integers(0, 2000, 800) / 10.0).astype("<f8"))
    add("double_specials", O.DOUBLE, np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 2.5, -0.0, np.nan], dtype="<f8"))
    urls = [b"www.site%d.com/path/%d/item" % (i % 13, i * 7919 % 101) for i in range(300)]
    add("string_dict_front_coded", O.STRING, [urls[i] for i in rng.integers(0, 300, 900)])
    add("string_dict_plain", O.STRING, [[b"red", b"green", b"blue", b""][i] for i in rng.integers(0, 4, 500)])
    add("string_delta_whole_pack", O.STRING, sorted(urls)[:120], force=O.DELTA)
    add("string_simple_dict", O.STRING, [[b"x", b"yy", b"zzz"][i] for i in rng.integers(0, 3, 200)], force=O.SIMPLE_DICT)
    return out


if __name__ == "__main__":
    doc = {"provenance": "self-generated by oracle/ (restated PackCompressor_VLT.compress); not JVM output", "cases": cases()}
    with open(os.path.join(HERE, "cases.json"), "w") as fh:
        json.dump(doc, fh, indent=0)
    print("wrote %d cases, %d bytes" % (len(doc["cases"]), os.path.getsize(os.path.join(HERE, "cases.json"))))
