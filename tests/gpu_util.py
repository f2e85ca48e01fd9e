"""Helpers shared by the GPU parity tests: run packs through the C ABI batch path and hand back bytes + statuses."""
import numpy as np

from indexr_amd import lib as L


def align16(v):
    return (v + 15) & ~15


class Case:
    def __init__(self, pack, data_type, encode_type, n, data_size, expect=None, name=""):
        self.pack = np.ascontiguousarray(pack, dtype=np.uint8)
        self.data_type, self.encode_type, self.n, self.data_size = data_type, encode_type, n, data_size
        self.expect = expect  # np.uint8 decoded bytes, or None when the case must fail
        self.name = name


def build_arena(cases):
    descs, in_off, out_off = [], 0, 0
    for c in cases:
        descs.append(L.PackDesc(in_off, out_off, len(c.pack), c.data_size, c.n, c.encode_type, c.data_type, 0))
        in_off += align16(len(c.pack))
        out_off += align16(c.data_size)
    arena = np.zeros(in_off + L.IN_SLACK, dtype=np.uint8)
    for c, d in zip(cases, descs):
        arena[d.in_off:d.in_off + len(c.pack)] = c.pack
    return descs, arena, out_off + 16


def run_batch(cases, device=0, scratch_budget=0, fill=0xAB):
    """-> (list of decoded byte arrays, statuses)"""
    descs, arena, out_bytes = build_arena(cases)
    plan = L.Plan(device, descs, scratch_budget)
    d_in = L.DeviceBuffer(device, len(arena))
    d_out = L.DeviceBuffer(device, out_bytes)
    try:
        d_in.upload(arena)
        d_out.upload(np.full(out_bytes, fill, dtype=np.uint8))
        plan.run(d_in.ptr, len(arena), d_out.ptr, out_bytes)
        status, _ = plan.status()
        host = d_out.download()
        outs = [host[d.out_off:d.out_off + c.data_size] for c, d in zip(cases, descs)]
        # nothing may be written outside a pack's own output range
        guard = np.ones(out_bytes, dtype=bool)
        for c, d in zip(cases, descs):
            guard[d.out_off:d.out_off + c.data_size] = False
        assert (host[guard] == fill).all(), "decoder wrote outside the output ranges"
        return outs, status, plan
    finally:
        d_in.free()
        d_out.free()


def check_cases(cases, **kw):
    outs, status, plan = run_batch(cases, **kw)
    plan.destroy()
    for c, out, st in zip(cases, outs, status):
        if c.expect is None:
            assert st < 0, "%s: expected an error, got status %d" % (c.name, st)
        else:
            assert st == len(c.pack), "%s: status %d != packSize %d" % (c.name, st, len(c.pack))
            assert out.tobytes() == np.asarray(c.expect, dtype=np.uint8).tobytes(), "%s: decoded bytes differ" % c.name
