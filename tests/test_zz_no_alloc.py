"""ixr_decode_batch_static, the entry point that allocates nothing, against ixr_plan_run on the same packs: same decoded
bytes, same per-pack status, on two calls in a row of which the second is smaller (the static scratch keeps the first
call's contents)."""
import ctypes as C

import numpy as np
import pytest

from indexr_amd import lib as L

pytestmark = pytest.mark.gpu


def _with_plan(descs, arena, out_bytes, fill):
    plan = L.Plan(0, descs)
    d_in = L.DeviceBuffer(0, len(arena)); d_out = L.DeviceBuffer(0, out_bytes)
    d_in.upload(arena); d_out.upload(np.full(out_bytes, fill, dtype=np.uint8))
    plan.run(d_in.ptr, len(arena), d_out.ptr, out_bytes)
    status, _ = plan.status()
    host = d_out.download()
    plan.destroy(); d_in.free(); d_out.free()
    return host, status


def _static(descs, arena, out_bytes, fill, stream=None):
    n = len(descs)
    arr = descs if isinstance(descs, C.Array) else (L.PackDesc * n)(*descs)
    d_in = L.DeviceBuffer(0, len(arena)); d_out = L.DeviceBuffer(0, out_bytes); d_st = L.DeviceBuffer(0, 4 * n)
    d_in.upload(arena); d_out.upload(np.full(out_bytes, fill, dtype=np.uint8)); d_st.upload(np.full(4 * n, 0xEE, dtype=np.uint8))
    rc = L.lib.ixr_decode_batch_static(0, arr, n, d_in.ptr, len(arena), d_out.ptr, out_bytes, d_st.ptr, stream)
    assert rc == 0, L.lib.ixr_last_error()
    assert L.lib.ixr_stream_synchronize(0, stream) == 0
    host = d_out.download()
    status = d_st.download().view("<i4").copy()
    d_in.free(); d_out.free(); d_st.free()
    return host, status


def _batch(reqs, seed, corrupt=()):
    descs, arena, out_bytes, _ = L.synth_build(reqs, seed, threads=4)
    for i in corrupt:  # break the pack's framing: its first length word
        arena[descs[i].in_off + 1] ^= 0x5A
    return descs, arena, out_bytes + 16


def test_static_batch_equals_the_plan_on_two_calls_in_a_row():
    assert L.lib.ixr_init(0) == 0
    # first call: every column kind, full and partial packs (about 26 MB of working arrays)
    first = [(kind, 3, 65536 if kind % 2 == 0 else 5000, 0) for kind in range(13)]
    # second call: fewer and smaller packs, one of them corrupted, other column kinds in other places
    second = [(10, 9, 777, 0), (4, 9, 65536, 0), (12, 1, 3000, 0), (0, 2, 1, 0), (8, 5, 40000, 0)]
    for reqs, corrupt, fill in ((first, (), 0xAB), (second, (1,), 0x3C), (first[::-1], (2, 7), 0x00)):
        descs, arena, out_bytes = _batch(reqs, 0xBEEF, corrupt)
        want, want_status = _with_plan(descs, arena, out_bytes, fill)
        got, got_status = _static(descs, arena, out_bytes, fill)
        assert (got_status == want_status).all(), (got_status, want_status)
        for i in corrupt:
            assert got_status[i] < 0
        good = [d for d, st in zip(descs, want_status) if st == d.pack_size]
        assert len(good) == len(reqs) - len(corrupt)
        for d in good:
            assert got[d.out_off:d.out_off + d.data_size].tobytes() == want[d.out_off:d.out_off + d.data_size].tobytes()
        # nothing outside the packs' output ranges is touched
        guard = np.ones(out_bytes, dtype=bool)
        for d in descs:
            guard[d.out_off:d.out_off + d.data_size] = False
        assert (got[guard] == fill).all()


def test_static_batch_on_an_explicit_stream():
    st = L.lib.ixr_stream_create(0)
    assert st
    descs, arena, out_bytes = _batch([(2, 0, 65536, 0), (10, 0, 20000, 0)], 5)
    want, want_status = _with_plan(descs, arena, out_bytes, 0x11)
    got, got_status = _static(descs, arena, out_bytes, 0x11, stream=st)
    assert (got_status == want_status).all() and got.tobytes() == want.tobytes()
    L.lib.ixr_stream_destroy(0, st)


def test_static_batch_rejects_what_does_not_fit():
    descs, arena, out_bytes = _batch([(2, 0, 100, 0)] * 65, 1)
    d_in = L.DeviceBuffer(0, len(arena)); d_out = L.DeviceBuffer(0, out_bytes); d_st = L.DeviceBuffer(0, 4 * 65)
    rc = L.lib.ixr_decode_batch_static(0, descs, 65, d_in.ptr, len(arena), d_out.ptr, out_bytes, d_st.ptr, None)
    assert rc == L.ERR_INVALID_ARGUMENT and b"1..64" in L.lib.ixr_last_error()
    rc = L.lib.ixr_decode_batch_static(0, descs, 0, d_in.ptr, len(arena), d_out.ptr, out_bytes, d_st.ptr, None)
    assert rc == L.ERR_INVALID_ARGUMENT
    rc = L.lib.ixr_decode_batch_static(0, descs, 4, d_in.ptr, 64, d_out.ptr, out_bytes, d_st.ptr, None)
    assert rc == L.ERR_INVALID_ARGUMENT  # input arena too small
    rc = L.lib.ixr_decode_batch_static(0, descs, 4, d_in.ptr, len(arena), d_out.ptr, out_bytes, None, None)
    assert rc == L.ERR_INVALID_ARGUMENT  # no status array
    d_in.free(); d_out.free(); d_st.free()
    # ten full URL packs need about 50 MB of working arrays: more than the static scratch holds
    big, arena, out_bytes = _batch([(10, p, 65536, 0) for p in range(10)], 2)
    d_in = L.DeviceBuffer(0, len(arena)); d_out = L.DeviceBuffer(0, out_bytes); d_st = L.DeviceBuffer(0, 64)
    rc = L.lib.ixr_decode_batch_static(0, big, 10, d_in.ptr, len(arena), d_out.ptr, out_bytes, d_st.ptr, None)
    assert rc == L.ERR_INVALID_ARGUMENT and b"static scratch" in L.lib.ixr_last_error()
    d_in.free(); d_out.free(); d_st.free()
