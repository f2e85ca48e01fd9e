"""Times successive ixr_plan_run_host calls on one slice of the headline table (1024 pack ids of every column), for plans
made in different ways: which call of a plan is the slow one, and whether making and destroying another plan first changes it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

from indexr_amd import lib as L
from indexr_amd import table as T

sys.path.insert(0, ROOT)
import bench

dev = 0
assert L.lib.ixr_init(dev) == 0
kinds = bench.WORKLOADS["mixed"][1]
rows = 1024 * 65536
table = T.build_table(kinds, rows, 0, 0xC0FFEE, threads=16, log=lambda m: None)
descs, arena = table.sample
in_bytes = len(arena)
probe = L.Plan(dev, descs, 16 << 30)
out_bytes, total_out, need = probe.out_bytes + 16, probe.total_out, probe.scratch_bytes
probe.destroy()
h_in, h_out = L.PinnedBuffer(in_bytes), L.PinnedBuffer(out_bytes)
h_in.array[:] = arena
print("slice: %d packs, %.2f GB -> %.2f GB, scratch need %.2f GB" % (len(descs), in_bytes / 1e9, total_out / 1e9, need / 1e9), flush=True)


def calls(plan, n=5):
    out = []
    for _ in range(n):
        L.lib.ixr_dev_synchronize(dev)
        t = time.perf_counter()
        plan.run_host(h_in.ptr, in_bytes, h_out.ptr, out_bytes)
        out.append(total_out / (time.perf_counter() - t) / 1e9)
    return " ".join("%5.1f" % x for x in out)


for label, budget, pre in (("budget need/8", need // 8, False), ("budget need/8 again", need // 8, False), ("budget 2 GiB", 2 << 30, False),
                           ("a 16 GiB-budget plan made and destroyed first, then need/8", need // 8, True),
                           ("the same, 2 GiB", 2 << 30, True), ("budget need/8, no other plan before", need // 8, False)):
    if pre:
        w = L.Plan(dev, descs, 16 << 30)
        w.destroy()
    p = L.Plan(dev, descs, max(budget, 64 << 20))
    print("%-60s GB/s per call: %s" % (label, calls(p)), flush=True)
    p.destroy()
# a resident table beside it: 40 GB of device memory held while the same plan runs
big = L.DeviceBuffer(dev, 40 << 30)
p = L.Plan(dev, descs, max(need // 8, 64 << 20))
print("%-60s GB/s per call: %s" % ("40 GB of device memory held elsewhere", calls(p)), flush=True)
p.destroy()
big.free()
h_in.free(); h_out.free()
