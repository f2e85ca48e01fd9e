import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
import oracle_lib as O
from indexr_amd import lib as L
for first in (1 << 20, 1 << 26, (1 << 27) - 1, 1 << 27, 1 << 28, 1 << 29, 1 << 30, (1 << 31) - 1, -(1 << 31)):
    v = np.array([first], dtype='<i4')
    data = O.rows_to_bytes(v, O.INT)
    pack = O.compress_pack_as(data, O.INT, 1, O.DELTA)
    out, st = L.decode_pack(pack, O.INT, O.DELTA, 1, len(data))
    print(first, 'pack', pack.tobytes().hex(), 'status', st, 'out', out.view('<i4'))
for first in (1 << 40, 1 << 62, -(1 << 63)):
    v = np.array([first], dtype='<i8')
    data = O.rows_to_bytes(v, O.LONG)
    pack = O.compress_pack_as(data, O.LONG, 1, O.DELTA)
    out, st = L.decode_pack(pack, O.LONG, O.DELTA, 1, len(data))
    print(first, 'pack', pack.tobytes().hex(), 'status', st, 'out', out.view('<i8'))
