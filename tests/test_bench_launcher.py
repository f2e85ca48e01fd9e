"""bench.py's own launcher and its strong-scaling partition, on CPU: `python bench.py --gpus 2` starts two ranks itself
(gloo instead of RCCL, --plan-only instead of the decode), which build their shares of ONE table, agree on byte-balanced
contiguous pack-id ranges and reduce their counters.  The union must be the single-process table."""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

from indexr_amd import shard as S  # noqa: E402
from indexr_amd import table as T  # noqa: E402

ROWS = 11 * 65536 + 4321  # 12 packs per column, the last one short
KINDS = "2,4,8,10"


def _bench(*extra, expect_rc=0):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--plan-only", "--backend", "gloo", "--rows", str(ROWS), "--kinds", KINDS,
           "--threads", "2"] + list(extra)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert (p.returncode == 0) == (expect_rc == 0), p.stderr.decode()[-2000:]
    if p.returncode:
        return None
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, "exactly one JSON line on stdout, got %r" % lines
    return json.loads(lines[0])


def test_two_ranks_started_by_bench_itself_cover_the_table_once():
    one = _bench("--gpus", "1", "--scaling", "strong")
    two = _bench("--gpus", "2")
    three = _bench("--gpus", "3")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and three["n_gpus"] == 3
    assert two["scaling"] == "strong" and three["scaling"] == "strong"
    n_packs = T.pack_count(ROWS)
    for doc in (two, three):
        ranges = doc["pack_id_ranges"]
        assert len(ranges) == doc["n_gpus"] and ranges[0][0] == 0
        for (f0, c0), (f1, _) in zip(ranges, ranges[1:]):
            assert f0 + c0 == f1 and c0 > 0
        assert ranges[-1][0] + ranges[-1][1] == n_packs
        # same rows, same bytes, same packs (checksum of checksums) as the table in one process
        for key in ("rows", "decoded_bytes", "compressed_bytes", "checksum_halves", "packs"):
            assert doc[key] == one[key], key
    assert one["rows"] == ROWS * 4


def test_weak_scaling_gives_every_rank_a_table_of_its_own():
    one = _bench("--gpus", "1")
    two = _bench("--gpus", "2", "--scaling", "weak")
    assert one["scaling"] == "weak" and two["scaling"] == "weak" and two["n_gpus"] == 2
    assert two["rows"] == 2 * one["rows"] and two["packs"] == 2 * one["packs"]
    assert two["checksum_halves"] != [2 * v for v in one["checksum_halves"]]  # rank 1 holds other pack ids than rank 0


def test_a_failing_rank_fails_the_run():
    # one pack id cannot be split over two ranks: rank 1 is left without work and exits non-zero, the launcher reports it
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--plan-only", "--backend", "gloo", "--rows", "1000", "--kinds", "2",
           "--gpus", "2", "--threads", "1"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0


def test_gpus_must_match_the_world_size():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--plan-only", "--rows", "1000", "--kinds", "2", "--gpus", "2"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=env)
    assert p.returncode != 0 and b"WORLD_SIZE" in p.stderr


def test_retarget_builds_the_missing_pack_ids_and_drops_the_others():
    kinds = [2, 10]
    t = T.build_range(kinds, ROWS, 0, 4, 7, threads=2, chunk_packs=3)
    T.retarget(t, kinds, ROWS, 2, 9, 7, threads=2, chunk_packs=3)
    want = T.build_range(kinds, ROWS, 2, 9, 7, threads=2)
    key = lambda tab: sorted((p, d.data_type, d.pack_size, d.data_size, d.obj_count, int(s)) for p, d, s in zip(tab.pack_ids, tab.descs, tab.sums))
    assert key(t) == key(want)
    # every kept descriptor still points at its own bytes
    a, b = t.input_arena(), want.input_arena()
    bytes_of = lambda tab, arena: sorted(arena[d.in_off:d.in_off + d.pack_size].tobytes() for d in tab.descs)
    assert bytes_of(t, a) == bytes_of(want, b)
    outs = sorted((d.out_off, d.data_size) for d in t.descs)
    assert all(o0 + s0 <= o1 for (o0, s0), (o1, _) in zip(outs, outs[1:])) and outs[-1][0] + outs[-1][1] <= t.out_bytes
    # the last pack of the table is the short one
    assert T.rows_of_pack(ROWS, 11) == 4321 and T.rows_of_pack(ROWS, 10) == 65536
    w = want.weights()
    assert sorted(w) == list(range(2, 11)) and all(v > 0 for v in w.values())
    parts = S.balanced_ranges([w[p] for p in sorted(w)], 2)
    assert parts[0][0] == 0 and parts[0][1] + parts[1][1] == 9
