import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from indexr_amd import lib as L
import oracle_lib as O
for kind in (2, 4, 8, 10, 0):
    descs, arena, ob, sums = L.synth_build([(kind, 0, 65536, 0)], 0xC0FFEE, threads=1)
    d = descs[0]
    pack = arena[d.in_off:d.in_off + d.pack_size].copy()
    for _ in range(3):
        out, st = L.decode_pack(pack, d.data_type, d.encode_type, d.obj_count, d.data_size)
    t = time.perf_counter(); N = 20
    for _ in range(N):
        out, st = L.decode_pack(pack, d.data_type, d.encode_type, d.obj_count, d.data_size)
    dt = (time.perf_counter() - t) / N
    t = time.perf_counter()
    for _ in range(N):
        o2, c2 = O.decompress_pack(pack, d.data_type, d.encode_type, d.obj_count, d.data_size)
    dc = (time.perf_counter() - t) / N
    print('kind', kind, 'gpu single-pack %.3f ms' % (dt * 1e3), 'cpu oracle %.3f ms' % (dc * 1e3), 'ok', st == d.pack_size and np.array_equal(out, o2))
