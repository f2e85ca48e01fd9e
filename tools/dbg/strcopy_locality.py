#!/usr/bin/env python3
"""How much of k_str_copy's time is the dictionary's cache behaviour?  The same 65536-row URL-shaped pack shapes with
different id distributions (hot set that fits L1 / uniform over 11K entries / Zipf), each replicated over `--packs`
packs, timed per kernel with the plan's HIP events.  Run on the GPU box."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from indexr_amd import lib as L  # noqa: E402


def urls(rng, n):
    doms = [bytes(rng.integers(97, 123, rng.integers(4, 13)).astype(np.uint8)) for _ in range(300)]
    out = set()
    while len(out) < n:
        out.add(b"www." + doms[rng.integers(0, 300)] + b".com/" + bytes(rng.integers(97, 123, rng.integers(6, 60)).astype(np.uint8)))
    return sorted(out)


def pack_of(entries, ids):
    rows = [entries[i] for i in ids]
    off = np.zeros(len(rows) + 1, dtype="<i4")
    off[1:] = np.cumsum([len(r) for r in rows])
    data = np.frombuffer(off.tobytes() + b"".join(rows), dtype=np.uint8)
    pack, et = L.compress_pack(data, L.STRING, len(rows), False)
    return pack, et, len(data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--packs", type=int, default=1024)
    ap.add_argument("--only", default="", help="run only the shapes whose name contains this")
    args = ap.parse_args()
    rng = np.random.default_rng(3)
    ent = urls(rng, 11000)
    n = 65536
    zipf = np.minimum(rng.zipf(1.1, n) - 1, len(ent) - 1)
    perm = rng.permutation(len(ent))
    shapes = {
        "hot16 (16 distinct entries of 11000 used)": perm[rng.integers(0, 16, n)],
        "hot256": perm[rng.integers(0, 256, n)],
        "hot2048": perm[rng.integers(0, 2048, n)],
        "uniform 11000": rng.integers(0, len(ent), n),
        "zipf(1.1) scattered": perm[zipf],
        "zipf(1.1) hot entries adjacent": zipf,
        "sorted ids": np.sort(rng.integers(0, len(ent), n)),
    }
    assert L.lib.ixr_init(0) == 0
    for name, ids in shapes.items():
        if args.only and args.only not in name:
            continue
        # keep the dictionary the same 11000 entries in every shape: append one row of every entry
        ids = np.concatenate([ids[: n - len(ent)], np.arange(len(ent))])
        ids = ids if name == "sorted ids" else rng.permutation(ids)
        pack, et, data_size = pack_of(ent, ids)
        step_in, step_out = (len(pack) + 15) & ~15, (data_size + 15) & ~15
        descs = [L.PackDesc(i * step_in, i * step_out, len(pack), data_size, n, et, L.STRING, 0) for i in range(args.packs)]
        arena = np.zeros(args.packs * step_in + L.IN_SLACK, dtype=np.uint8)
        for i in range(args.packs):
            arena[i * step_in:i * step_in + len(pack)] = pack
        plan = L.Plan(0, descs)
        d_in = L.DeviceBuffer(0, len(arena)); d_out = L.DeviceBuffer(0, args.packs * step_out + 16)
        d_in.upload(arena)
        plan.set_overlap(False)
        plan.run(d_in.ptr, len(arena), d_out.ptr, d_out.nbytes)
        plan.status()
        plan.set_profiling(True)
        for _ in range(3):
            plan.run(d_in.ptr, len(arena), d_out.ptr, d_out.nbytes)
        _, bad = plan.status()
        assert bad == 0
        t = {k: (ms / l, alg) for k, ms, l, alg in plan.kernel_times()}
        ms, alg = t.get("k_str_fused") or t["k_str_copy"]
        print("%-46s k_str_copy %.3f ms  %.0f GB/s body  | str_index %.3f front_decode %.3f | dict %d B" % (
            name, ms, alg / ms / 1e6, t["k_str_index"][0], t.get("k_front_decode", (0, 0))[0], sum(len(e) for e in ent)), flush=True)
        if hasattr(L.lib, "ixr_debug_phases"):  # build with IXR_DEFS=-DIXR_PHASES: cycles per phase, summed over waves
            import ctypes
            ph = (ctypes.c_ulonglong * 16)()
            L.lib.ixr_debug_phases(ph)
            waves = 4 * 64 * args.packs  # 4 runs
            print("   phases, cycles per wave:", " ".join("%d" % (v / waves) for v in ph[:12]), flush=True)
        plan.destroy(); d_in.free(); d_out.free()


if __name__ == "__main__":
    main()
