"""Per-pack latency of the single-pack entry points on full 65536-row packs, beside the oracle on one host core.
For dictionary packs the calls are the pair DataPack.decompress makes (DataPack.java:172-176): getDictStruct on the pack,
then getDataFromDictStruct on what it returned; ixr_decode_pack is timed for every kind as well (what a caller gets that
asks for the rows in one call)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np

import oracle_lib as O
from indexr_amd import lib as L

N = 30


def timed(fn):
    for _ in range(3):
        r = fn()
    t = time.perf_counter()
    for _ in range(N):
        r = fn()
    return (time.perf_counter() - t) / N * 1e3, r


for kind in (2, 0, 4, 7, 8, 10, 9):
    descs, arena, ob, sums = L.synth_build([(kind, 0, 65536, 0)], 0xC0FFEE, threads=1)
    d = descs[0]
    pack = arena[d.in_off:d.in_off + d.pack_size].copy()
    g_ms, (out, st) = timed(lambda: L.decode_pack(pack, d.data_type, d.encode_type, d.obj_count, d.data_size))
    c_ms, (o2, c2) = timed(lambda: O.decompress_pack(pack, d.data_type, d.encode_type, d.obj_count, d.data_size))
    line = '%-7s decode_pack %.3f ms (oracle %.3f) ok %s' % (L.COL_NAMES[kind], g_ms, c_ms, st == d.pack_size and np.array_equal(out, o2))
    if d.encode_type == L.ENC_DICT_COMPRESS:
        def gpu_pair():
            ds, consumed = L.get_dict_struct(pack, d.data_type, d.obj_count)
            rows, written = L.get_data_from_dict_struct(ds, d.data_type, d.obj_count, d.data_size)
            return ds, consumed, rows, written

        def cpu_pair():
            ds, consumed = O.get_dict_struct(pack, d.data_type, d.obj_count)
            rows, written = O.data_from_dict_struct(ds, d.data_type, d.obj_count, d.data_size)
            return ds, consumed, rows, written
        gp_ms, (ds, consumed, rows, written) = timed(gpu_pair)
        cp_ms, (ods, oc, orows, ow) = timed(cpu_pair)
        g1_ms, _ = timed(lambda: L.get_dict_struct(pack, d.data_type, d.obj_count))
        ok = consumed == d.pack_size and np.array_equal(ds, ods) and np.array_equal(rows, orows) and written == ow
        line += ' | getDictStruct + getDataFromDictStruct %.3f ms (getDictStruct alone %.3f; oracle pair %.3f) ok %s' % (gp_ms, g1_ms, cp_ms, ok)
    print(line, flush=True)
