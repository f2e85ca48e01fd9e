"""Partitioning of pack work over GPUs.

Every pack decodes on its own (PackCompressor_VLT.java:131-170 touches only its dpn and cmpData), so the decode shards
with no data-path exchange.  Two assignments:

  contiguous_range   the reference's SegmentAssigner rule: item i of `total` goes to the g-th contiguous chunk, chunk
                     sizes differ by at most one (indexr-query-opt/.../SegmentAssigner.java:176-188)
  balanced_ranges    contiguous chunks balanced by bytes (packSize + dataSize) instead of count; string packs vary in size
"""


def contiguous_range(total, world, rank):
    """-> (first, count) of the rank's chunk; chunks are contiguous, sizes differ by <= 1, larger chunks first."""
    if world <= 0 or not 0 <= rank < world or total < 0:
        raise ValueError("bad partition request: total=%r world=%r rank=%r" % (total, world, rank))
    base, extra = divmod(total, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def balanced_ranges(weights, world):
    """-> list of (first, count), one per rank: contiguous chunks cut where the running weight crosses k/world of the
    total (each cut placed on the nearer side), so chunk weights differ by at most one item's weight."""
    if world <= 0:
        raise ValueError("world must be positive")
    n, total = len(weights), sum(weights)
    cuts, i, acc = [0], 0, 0
    for r in range(1, world):
        target = total * r / world
        while i < n and acc + weights[i] / 2.0 <= target:
            acc += weights[i]
            i += 1
        cuts.append(i)
    cuts.append(n)
    return [(cuts[r], cuts[r + 1] - cuts[r]) for r in range(world)]


def weak_scaling_first_pack(rank, packs_per_rank):
    """bench.py's weak scaling: rank r owns pack ids [r * packs_per_rank, (r + 1) * packs_per_rank)."""
    return rank * packs_per_rank
