"""One 100-pack segment of the full column set of the 100 M-row numeric table (7 columns) and of the 100 M-row
varchar-heavy table (4 columns), every pack compared with the oracle's decode of the same bytes; and the rule
ixr_plan_run_host has for the layout of its output arena."""
import numpy as np
import pytest

import oracle_lib as O
from indexr_amd import lib as L

pytestmark = pytest.mark.gpu

NUMERIC = [L.COL_I_SEQ, L.COL_I_RAND, L.COL_I_CAT, L.COL_I_RUNS, L.COL_L_TS, L.COL_L_CARD, L.COL_L_RAND]
VARCHAR = [L.COL_S_CAT, L.COL_S_URL, L.COL_S_UNIQ, L.COL_S_EMPTY]


def _segment(kinds, seed, packs=100, last_rows=16960):
    """a segment of 6 553 600 rows is 100 packs per column; the table's last segment ends with a short pack"""
    reqs = []
    for pack_id in range(packs):
        for kind in kinds:
            reqs.append((kind, pack_id, last_rows if pack_id == packs - 1 else 65536, 0))
    return reqs


@pytest.mark.parametrize("name,kinds", [("numeric", NUMERIC), ("varchar", VARCHAR)])
def test_one_segment_of_each_100m_row_table_against_the_oracle(name, kinds):
    reqs = _segment(kinds, 0)
    descs, arena, out_bytes, sums = L.synth_build(reqs, 0xC0FFEE, threads=8)
    plan = L.Plan(0, descs)
    d_in, d_out = L.DeviceBuffer(0, len(arena)), L.DeviceBuffer(0, out_bytes + 16)
    try:
        d_in.upload(arena)
        plan.run(d_in.ptr, len(arena), d_out.ptr, out_bytes + 16)
        status, bad = plan.status()
        assert bad == 0
        got_sums = plan.checksums(d_out.ptr, out_bytes + 16)
        assert (got_sums == sums).all()  # every pack against the checksum of its source rows
        host = d_out.download()
        kinds_seen = set()
        for (kind, pack_id, rows, _), d in zip(reqs, descs):
            pack = arena[d.in_off:d.in_off + d.pack_size]
            want, consumed = O.decompress_pack(pack, d.data_type, d.encode_type, rows, d.data_size)
            assert consumed == d.pack_size, (L.COL_NAMES[kind], pack_id)
            assert host[d.out_off:d.out_off + d.data_size].tobytes() == want.tobytes(), (L.COL_NAMES[kind], pack_id)
            kinds_seen.add((kind, d.encode_type))
        assert len({k for k, _ in kinds_seen}) == len(kinds)
    finally:
        plan.destroy()
        d_in.free()
        d_out.free()


def test_host_arenas_with_a_gap_between_two_packs_are_refused():
    """ixr_plan_run_host copies a group's rows back as one stretch of the output arena: packs of one group that do not lie
    back to back (more than the alignment bytes between them) would have whatever the caller keeps in the gap overwritten,
    so such a plan is refused and the arena stays as it was; the same plan runs through ixr_plan_run."""
    reqs = [(L.COL_I_CAT, 0, 65536, 0), (L.COL_L_TS, 0, 65536, 0), (L.COL_S_URL, 0, 20000, 0)]
    descs, arena, out_bytes, sums = L.synth_build(reqs, 5, threads=2)
    gap = 4096
    descs[1].out_off += gap
    descs[2].out_off += gap
    out_bytes += gap + 16
    plan = L.Plan(0, descs)
    h_in, h_out = L.PinnedBuffer(len(arena)), L.PinnedBuffer(out_bytes)
    d_in, d_out = L.DeviceBuffer(0, len(arena)), L.DeviceBuffer(0, out_bytes)
    try:
        h_in.array[:] = arena
        h_out.array[:] = 0x77
        with pytest.raises(L.IxrError) as e:
            plan.run_host(h_in.ptr, len(arena), h_out.ptr, out_bytes)
        assert e.value.code == L.ERR_INVALID_ARGUMENT and "back to back" in str(e.value)
        assert (h_out.array == 0x77).all()
        # the device-resident run takes the same layout and leaves the gap alone
        d_in.upload(arena)
        d_out.upload(np.full(out_bytes, 0x77, dtype=np.uint8))
        plan.run(d_in.ptr, len(arena), d_out.ptr, out_bytes)
        status, bad = plan.status()
        assert bad == 0
        host = d_out.download()
        lo = descs[0].out_off + descs[0].data_size
        assert (host[lo:descs[1].out_off] == 0x77).all()
        for (kind, pack_id, rows, _), d in zip(reqs, descs):
            assert host[d.out_off:d.out_off + d.data_size].tobytes() == L.synth_column(kind, 5, pack_id, rows).tobytes()
    finally:
        plan.destroy()
        h_in.free(); h_out.free(); d_in.free(); d_out.free()
