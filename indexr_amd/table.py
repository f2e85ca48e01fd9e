"""Synthetic column tables for the benchmark and the sharding tests: packs of the requested column kinds for a range of
pack ids, compressed by the library's host encoder, laid out as one input arena (in chunks) plus pack descriptors.

Every pack's content depends on (seed, column kind, pack id) only, so any rank regenerates exactly the packs of its own
pack-id range without shipping data (pack geometry: DataPack.java:36-38, rowCountToPackCount DataPack.java:773-803)."""
import numpy as np

from . import lib as L

PACK_ROWS = 65536


def pack_count(rows):
    return (rows + PACK_ROWS - 1) >> 16


class Table:
    def __init__(self):
        self.descs = []        # L.PackDesc with absolute arena offsets
        self.chunks = []       # (arena offset, np.uint8 bytes) pieces of the input arena
        self.in_bytes = 0      # input arena size including the trailing slack
        self.out_bytes = 0     # output arena size
        self.sums = None       # expected checksum of every decoded pack (ixr_checksum of the source rows)
        self.sample = None     # (descs, arena) of the first chunk with chunk-relative offsets: the CPU baseline's input

    def input_arena(self):
        """the whole input arena in one host array (small tables only)"""
        a = np.zeros(self.in_bytes, dtype=np.uint8)
        for off, c in self.chunks:
            a[off:off + len(c)] = c
        return a


def build_table(kinds, rows, first_pack, seed, threads=0, chunk_packs=1024, log=None):
    """Packs [first_pack, first_pack + pack_count(rows)) of every column kind in `kinds`, column-interleaved per pack id."""
    n_packs = pack_count(rows)
    last_rows = rows - (n_packs - 1) * PACK_ROWS
    t = Table()
    in_off = out_off = 0
    sums = []
    for c0 in range(0, n_packs, chunk_packs):
        c1 = min(n_packs, c0 + chunk_packs)
        reqs = []
        for p in range(c0, c1):
            r = last_rows if p == n_packs - 1 else PACK_ROWS
            for k in kinds:
                reqs.append((k, first_pack + p, r, 0))
        descs, arena, chunk_out, chunk_sums = L.synth_build(reqs, seed, threads=threads)
        body = len(arena) - L.IN_SLACK
        for d in descs:
            t.descs.append(L.PackDesc(d.in_off + in_off, d.out_off + out_off, d.pack_size, d.data_size, d.obj_count,
                                      d.encode_type, d.data_type, 0))
        sums.append(chunk_sums)
        if t.sample is None:
            t.sample = (list(descs), arena)
        t.chunks.append((in_off, arena[:body]))
        in_off += body
        out_off += chunk_out
        if log is not None and (c0 // chunk_packs) % 4 == 0:
            log("built %d / %d packs per column" % (c1, n_packs))
    t.in_bytes = in_off + L.IN_SLACK
    t.out_bytes = out_off + 16
    t.sums = np.concatenate(sums) if sums else np.zeros(0, dtype=np.uint64)
    return t


def reduce_over_ranks(dist, torch, device, elapsed, counters):
    """Whole-job figures of a multi-rank run: max of the ranks' elapsed time, sum of their counters (rows, bytes, bad
    packs).  `device` is where the small tensors live ("cuda" under RCCL, "cpu" under gloo).  Nothing of the data path."""
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([int(v) for v in counters], dtype=torch.int64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), [int(v) for v in c.tolist()]
