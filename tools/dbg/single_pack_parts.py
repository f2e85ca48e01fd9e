import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
from indexr_amd import lib as L
descs, arena, ob, sums = L.synth_build([(8, 0, 65536, 0)], 0xC0FFEE, threads=1)
d = descs[0]
pack = arena[d.in_off:d.in_off + d.pack_size].copy()
def t(f, n=30):
    for _ in range(3): f()
    a = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - a) / n * 1e3
print("decode_pack      %.3f ms" % t(lambda: L.decode_pack(pack, d.data_type, d.encode_type, d.obj_count, d.data_size)))
def plan_only():
    p = L.Plan(0, [L.PackDesc(0, 0, d.pack_size, d.data_size, d.obj_count, d.encode_type, d.data_type, 0)]); p.destroy()
print("plan create+destroy %.3f ms" % t(plan_only))
def bufs():
    a = L.DeviceBuffer(0, d.pack_size + 32); b = L.DeviceBuffer(0, d.data_size + 32); a.free(); b.free()
print("2 x alloc+free   %.3f ms" % t(bufs))
p = L.Plan(0, [L.PackDesc(0, 0, d.pack_size, d.data_size, d.obj_count, d.encode_type, d.data_type, 0)])
a = L.DeviceBuffer(0, p.in_bytes); b = L.DeviceBuffer(0, p.out_bytes + 16)
buf = np.zeros(p.in_bytes, dtype=np.uint8); buf[:d.pack_size] = pack
def up(): a.upload(buf)
print("H2D              %.3f ms" % t(up))
def run():
    p.run(a.ptr, p.in_bytes, b.ptr, p.out_bytes + 16); p.status()
print("run + status     %.3f ms" % t(run))
print("D2H              %.3f ms" % t(lambda: b.download()))
