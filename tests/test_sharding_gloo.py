"""The N > 1 path of the benchmark on CPU: two gloo ranks, each building and (with the CPU oracle standing in for the
GPU) decoding its own pack-id range; the union must equal the single-process table and the reduced counters must add up.
Plus unit tests of the partition rules (SegmentAssigner.java:176-188)."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

from indexr_amd import shard as S  # noqa: E402

KINDS = [2, 4, 8, 10]
ROWS_PER_RANK = 2 * 65536 + 1000  # 3 packs per column, the last one short
SEED = 0xC0FFEE


def test_contiguous_range_covers_everything_once():
    for total in (0, 1, 7, 16, 15259, 65537):
        for world in (1, 2, 3, 8):
            got, sizes = [], []
            for r in range(world):
                first, count = S.contiguous_range(total, world, r)
                got.extend(range(first, first + count))
                sizes.append(count)
            assert got == list(range(total))
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        S.contiguous_range(10, 2, 2)


def test_balanced_ranges_are_contiguous_and_even():
    rng = np.random.default_rng(5)
    w = [int(v) for v in rng.integers(50_000, 700_000, 500)]
    for world in (1, 2, 4, 8):
        parts = S.balanced_ranges(w, world)
        assert parts[0][0] == 0 and sum(c for _, c in parts) == len(w)
        for (f0, c0), (f1, _) in zip(parts, parts[1:]):
            assert f0 + c0 == f1
        loads = [sum(w[f:f + c]) for f, c in parts]
        assert max(loads) - min(loads) <= 2 * max(w)
    assert S.balanced_ranges([], 2) == [(0, 0), (0, 0)]


def _decode_with_oracle(table):
    """checksums of the oracle's decode of every pack of the table"""
    import ctypes as C
    sys.path.insert(0, HERE)
    import oracle_lib as O
    from indexr_amd import lib as L
    arena = table.input_arena()
    n = len(table.descs)
    arr = (O.PackDesc * n)()
    for i, d in enumerate(table.descs):
        arr[i] = O.PackDesc(d.in_off, d.out_off, d.pack_size, d.data_size, d.obj_count, d.encode_type, d.data_type, 0)
    out = np.zeros(table.out_bytes, dtype=np.uint8)
    bad = O.lib.ixo_decode_batch(arr, n, arena.ctypes.data_as(C.POINTER(C.c_uint8)), out.ctypes.data_as(C.POINTER(C.c_uint8)), 2)
    assert bad == 0
    sums = np.array([L.checksum(out[d.out_off:d.out_off + d.data_size]) for d in table.descs], dtype=np.uint64)
    return sums


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import time
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from indexr_amd import table as T
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        n_packs = T.pack_count(ROWS_PER_RANK)
        first = S.weak_scaling_first_pack(rank, n_packs)
        table = T.build_table(KINDS, ROWS_PER_RANK, first, SEED, threads=2, chunk_packs=2)
        dist.barrier()
        t0 = time.perf_counter()
        sums = _decode_with_oracle(table)
        dist.barrier()
        elapsed = time.perf_counter() - t0
        ok = bool((sums == table.sums).all())
        rows = sum(d.obj_count for d in table.descs)
        out_b = sum(d.data_size for d in table.descs)
        in_b = sum(d.pack_size for d in table.descs)
        t_max, totals = T.reduce_over_ranks(dist, torch, "cpu", elapsed, [rows, out_b, in_b, 0 if ok else 1])
        q.put((rank, ok, elapsed, t_max, totals, [int(v) for v in table.sums], [(d.pack_size, d.data_size, d.obj_count) for d in table.descs]))
    finally:
        dist.destroy_process_group()


def test_two_ranks_decode_disjoint_pack_ranges_and_reduce():
    import torch.multiprocessing as mp
    from indexr_amd import table as T
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, ok0, e0, tmax0, tot0, sums0, geo0), (r1, ok1, e1, tmax1, tot1, sums1, geo1) = res
    assert (r0, r1) == (0, 1) and ok0 and ok1
    # both ranks hold the same reduced figures: max time, summed counters
    assert tot0 == tot1 and tmax0 == tmax1 and tmax0 >= max(e0, e1) - 1e-9
    assert tot0[0] == 2 * ROWS_PER_RANK * len(KINDS) and tot0[3] == 0
    assert tot0[1] == sum(g[1] for g in geo0 + geo1) and tot0[2] == sum(g[0] for g in geo0 + geo1)
    # a pack depends on (seed, kind, pack id) only: rank 1's packs are the packs [P, 2P) of a single-process table
    n_packs = T.pack_count(ROWS_PER_RANK)
    whole = T.build_table(KINDS, ROWS_PER_RANK, n_packs, SEED, threads=2)
    assert [int(v) for v in whole.sums] == sums1
    assert [(d.pack_size, d.data_size, d.obj_count) for d in whole.descs] == geo1
    assert sums0 != sums1
