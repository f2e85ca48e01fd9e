"""Synthetic column tables for the benchmark and the sharding tests: packs of the requested column kinds for a range of
pack ids, compressed by the library's host encoder, laid out as one input arena (in chunks) plus pack descriptors.

Every pack's content depends on (seed, column kind, pack id) only, so any rank regenerates exactly the packs of its own
pack-id range without shipping data (pack geometry: DataPack.java:36-38, rowCountToPackCount DataPack.java:773-803)."""
import numpy as np

from . import lib as L

PACK_ROWS = 65536


def pack_count(rows):
    return (rows + PACK_ROWS - 1) >> 16


def rows_of_pack(total_rows, pack_id):
    """rows of pack `pack_id` of a column of `total_rows` rows: every pack is full but the last (DataPack.java:773-803)"""
    n = pack_count(total_rows)
    if not 0 <= pack_id < n:
        raise ValueError("pack id %d outside a table of %d packs" % (pack_id, n))
    return total_rows - (n - 1) * PACK_ROWS if pack_id == n - 1 else PACK_ROWS


class Table:
    def __init__(self):
        self.descs = []        # L.PackDesc with absolute arena offsets
        self.pack_ids = []     # pack id (row range of the table) of every descriptor
        self.chunks = []       # (arena offset, np.uint8 bytes) pieces of the input arena
        self.in_bytes = L.IN_SLACK  # input arena size including the trailing slack
        self.out_bytes = 16    # output arena size
        self.sums = np.zeros(0, dtype=np.uint64)  # expected checksum of every decoded pack (ixr_checksum of the source rows)
        self.sample = None     # (descs, arena) of the first chunk with chunk-relative offsets: the CPU baseline's input

    def input_arena(self):
        """the whole input arena in one host array (small tables only)"""
        a = np.zeros(self.in_bytes, dtype=np.uint8)
        for off, c in self.chunks:
            a[off:off + len(c)] = c
        return a

    def weights(self):
        """{pack id: sum over its columns of packSize + dataSize}: what a pack id costs to decode, the balancing weight"""
        w = {}
        for d, p in zip(self.descs, self.pack_ids):
            w[p] = w.get(p, 0) + d.pack_size + d.data_size
        return w


def append_packs(t, kinds, total_rows, pack_ids, seed, id_base=0, threads=0, chunk_packs=1024, log=None):
    """Appends the packs `pack_ids` (ids inside a table of `total_rows` rows) of every column kind to table `t`,
    column-interleaved per pack id.  The synthetic content is that of pack id_base + pack id."""
    pack_ids = list(pack_ids)
    in_off, out_off = t.in_bytes - L.IN_SLACK, t.out_bytes - 16
    sums = [t.sums]
    for c0 in range(0, len(pack_ids), chunk_packs):
        ids = pack_ids[c0:c0 + chunk_packs]
        reqs = [(k, id_base + p, rows_of_pack(total_rows, p), 0) for p in ids for k in kinds]
        descs, arena, chunk_out, chunk_sums = L.synth_build(reqs, seed, threads=threads)
        body = len(arena) - L.IN_SLACK
        for d, (_, sp, _, _) in zip(descs, reqs):
            t.descs.append(L.PackDesc(d.in_off + in_off, d.out_off + out_off, d.pack_size, d.data_size, d.obj_count,
                                      d.encode_type, d.data_type, 0))
            t.pack_ids.append(sp - id_base)
        sums.append(chunk_sums)
        if t.sample is None:
            t.sample = (list(descs), arena)
        t.chunks.append((in_off, arena[:body]))
        in_off += body
        out_off += chunk_out
        if log is not None and (c0 // chunk_packs) % 4 == 0:
            log("built %d / %d packs per column" % (c0 + len(ids), len(pack_ids)))
    t.in_bytes = in_off + L.IN_SLACK
    t.out_bytes = out_off + 16
    t.sums = np.concatenate(sums)
    return t


def build_table(kinds, rows, first_pack, seed, threads=0, chunk_packs=1024, log=None):
    """A whole table of `rows` rows whose synthetic content is that of pack ids [first_pack, first_pack + pack_count(rows))."""
    return append_packs(Table(), kinds, rows, range(pack_count(rows)), seed, id_base=first_pack, threads=threads,
                        chunk_packs=chunk_packs, log=log)


def build_range(kinds, total_rows, first, count, seed, threads=0, chunk_packs=1024, log=None):
    """Packs [first, first + count) of a table of `total_rows` rows: one rank's share of a partitioned table."""
    return append_packs(Table(), kinds, total_rows, range(first, first + count), seed, threads=threads,
                        chunk_packs=chunk_packs, log=log)


def retarget(t, kinds, total_rows, first, count, seed, threads=0, chunk_packs=1024):
    """Makes `t` hold exactly the pack ids [first, first + count): builds the ones it lacks, drops the descriptors of the
    others (their bytes stay in the arena, unused).  Used after the ranks have agreed on byte-balanced ranges."""
    have = set(t.pack_ids)
    missing = [p for p in range(first, first + count) if p not in have]
    if missing:
        append_packs(t, kinds, total_rows, missing, seed, threads=threads, chunk_packs=chunk_packs)
    keep = [i for i, p in enumerate(t.pack_ids) if first <= p < first + count]
    if len(keep) != len(t.descs):
        t.descs = [t.descs[i] for i in keep]
        t.pack_ids = [t.pack_ids[i] for i in keep]
        t.sums = t.sums[np.asarray(keep, dtype=np.int64)] if keep else np.zeros(0, dtype=np.uint64)
    return t


def reduce_over_ranks(dist, torch, device, elapsed, counters):
    """Whole-job figures of a multi-rank run: max of the ranks' elapsed time, sum of their counters (rows, bytes, bad
    packs).  `device` is where the small tensors live ("cuda" under RCCL, "cpu" under gloo).  Nothing of the data path."""
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([int(v) for v in counters], dtype=torch.int64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), [int(v) for v in c.tolist()]


def gather_weights(dist, torch, device, n_packs, local_weights):
    """Every rank contributes the weights of the pack ids it built (a dict); all ranks get the whole table's weight list.
    This is plan metadata (the DataPackNode table a reader loads when it opens a segment, StorageColumn.java:47-67), a few
    bytes per pack id, exchanged once before anything is timed."""
    w = torch.zeros(n_packs, dtype=torch.int64, device=device)
    if local_weights:
        ids = torch.tensor(sorted(local_weights), dtype=torch.int64, device=device)
        w[ids] = torch.tensor([local_weights[p] for p in sorted(local_weights)], dtype=torch.int64, device=device)
    dist.all_reduce(w, op=dist.ReduceOp.SUM)
    return [int(v) for v in w.tolist()]
