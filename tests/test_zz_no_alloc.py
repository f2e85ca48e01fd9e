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


def test_static_batch_under_graph_capture():
    """What the entry point exists for: it must be usable where nothing may be allocated and nothing waited for.  The call is
    made between hipStreamBeginCapture (global mode, in which an allocation, a free or a synchronisation anywhere in the
    process invalidates the capture) and hipStreamEndCapture; the captured graph is then launched twice and its output
    compared with the plan's."""
    hip = C.CDLL("libamdhip64.so.7")  # the runtime libixrgpu is linked against (already mapped into the process)
    hip.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
    hip.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    hip.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    hip.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipGraphExecDestroy.argtypes = [C.c_void_p]
    hip.hipGraphDestroy.argtypes = [C.c_void_p]
    assert L.lib.ixr_init(0) == 0  # resolves the static array's address outside the capture
    st = L.lib.ixr_stream_create(0)
    assert st
    reqs = [(2, 1, 65536, 0), (4, 1, 65536, 0), (10, 1, 30000, 0), (0, 1, 65536, 0), (12, 1, 9000, 0), (8, 1, 65536, 0)]
    descs, arena, out_bytes = _batch(reqs, 77, corrupt=(3,))
    n = len(descs)
    want, want_status = _with_plan(descs, arena, out_bytes, 0x42)
    d_in = L.DeviceBuffer(0, len(arena)); d_out = L.DeviceBuffer(0, out_bytes); d_st = L.DeviceBuffer(0, 4 * n)
    d_in.upload(arena)
    graph, graph_exec = C.c_void_p(), C.c_void_p()
    try:
        assert hip.hipStreamBeginCapture(st, 0) == 0  # hipStreamCaptureModeGlobal
        rc = L.lib.ixr_decode_batch_static(0, descs, n, d_in.ptr, len(arena), d_out.ptr, out_bytes, d_st.ptr, st)
        end = hip.hipStreamEndCapture(st, C.byref(graph))
        assert rc == 0, L.lib.ixr_last_error()
        assert end == 0 and graph.value, "the capture was invalidated: the call allocated, freed or waited (hip error %d)" % end
        assert hip.hipGraphInstantiate(C.byref(graph_exec), graph, None, None, 0) == 0
        for rep in range(2):
            d_out.upload(np.full(out_bytes, 0x42, dtype=np.uint8))
            d_st.upload(np.full(4 * n, 0xEE, dtype=np.uint8))
            assert hip.hipGraphLaunch(graph_exec, st) == 0
            assert L.lib.ixr_stream_synchronize(0, st) == 0
            got = d_out.download()
            got_status = d_st.download().view("<i4")
            assert (got_status == want_status).all(), (rep, got_status, want_status)
            assert got_status[3] < 0
            for i, d in enumerate(descs):
                if want_status[i] == d.pack_size:
                    assert got[d.out_off:d.out_off + d.data_size].tobytes() == want[d.out_off:d.out_off + d.data_size].tobytes(), (rep, i)
    finally:
        if graph_exec.value:
            hip.hipGraphExecDestroy(graph_exec)
        if graph.value:
            hip.hipGraphDestroy(graph)
        d_in.free(); d_out.free(); d_st.free()
        L.lib.ixr_stream_destroy(0, st)


def test_static_batch_that_fills_the_scratch_to_the_last_bytes():
    """Batches whose working arrays land just below and just above the limit: the one that fits decodes like a plan does, the
    one that does not is refused (the layout keeps the same spare bytes behind its last array as a plan's scratch has)."""
    # numeric dictionary packs need about 0.6 MB each: grow the batch until it no longer fits, then check both sides
    fits = None
    for count in range(58, 65):
        reqs = [(4, p, 65536, 0) for p in range(count)]
        descs, arena, out_bytes = _batch(reqs, 9)
        d_in = L.DeviceBuffer(0, len(arena)); d_out = L.DeviceBuffer(0, out_bytes); d_st = L.DeviceBuffer(0, 4 * count)
        d_in.upload(arena)
        rc = L.lib.ixr_decode_batch_static(0, descs, count, d_in.ptr, len(arena), d_out.ptr, out_bytes, d_st.ptr, None)
        assert L.lib.ixr_stream_synchronize(0, None) == 0
        if rc == 0:
            fits = count
            status = d_st.download().view("<i4")
            got = d_out.download()
            assert all(int(s) == d.pack_size for s, d in zip(status, descs))
            last = descs[count - 1]
            rows = L.synth_column(4, 9, count - 1, 65536)
            assert got[last.out_off:last.out_off + last.data_size].tobytes() == rows.tobytes()
        else:
            assert rc == L.ERR_INVALID_ARGUMENT and b"static scratch" in L.lib.ixr_last_error()
        d_in.free(); d_out.free(); d_st.free()
        if rc != 0:
            break
    assert fits is not None and fits >= 58
