"""Phase times of k_enc_str (needs a library built with IXR_DEFS=-DIXR_ENC_TIMING; set IXR_LIB to it)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from indexr_amd import lib as L
kind = int(sys.argv[1]) if len(sys.argv) > 1 else 10
npk = int(sys.argv[2]) if len(sys.argv) > 2 else 512
descs, arena, out_bytes, _ = L.synth_build([(kind, p, 65536, 0) for p in range(npk)], 0xC0FFEE, threads=16)
plan = L.Plan(0, descs)
d_in, d_dec = L.DeviceBuffer(0, len(arena)), L.DeviceBuffer(0, out_bytes + 16)
d_in.upload(arena)
plan.run(d_in.ptr, len(arena), d_dec.ptr, out_bytes + 16)
ed, dst = [], 0
for d in descs:
    ed.append(L.EncodeDesc(d.out_off, dst, d.data_type, 0, d.obj_count, d.data_size, 0, 0, 0, 0, 0))
    dst += (L.lib.ixr_compress_bound(d.obj_count, d.data_size) + 15) & ~15
d_cmp = L.DeviceBuffer(0, dst)
import time
for it in range(2):
    t = time.perf_counter()
    arr, bad = L.encode_run(ed, d_dec.ptr, out_bytes + 16, d_cmp.ptr, dst)
    print("run", it, "ms", (time.perf_counter() - t) * 1e3, "bad", bad)
ph = np.zeros(5)
for a in arr:
    ph += [a.min_value >> 32, a.min_value & 0xffffffff, a.max_value >> 32, a.max_value & 0xffffffff, a.sum]
ph = ph / len(arr) / 100.0  # 100 MHz ticks -> microseconds
print("avg per pack (us): hash %.0f  compact+sort %.0f  ids+lens %.0f  entries %.0f  rle %.0f" % tuple(ph))
