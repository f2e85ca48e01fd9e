"""The C ABI from several threads at once, the way the reference's readers use the decode path: one reader thread per
task, each calling DataPack.decompress on its own packs (DPSegment.java:52: a DataPack belongs to one thread, the
codecs below it are stateless, PackCompressor_VLT.java:135-147).  Single-pack calls of four threads overlap on the
device through the library's pooled contexts; two plans run side by side on two streams.  Everything is compared with
the oracle."""
import ctypes as C
import threading

import numpy as np
import pytest

import oracle_lib as O
from gpu_util import build_arena
from indexr_amd import lib as L
from test_gpu_decode import oracle_case
from test_oracle import ALL_TYPES, _column, rng

pytestmark = pytest.mark.gpu


def _mixed_cases(seed, count):
    """packs of every type and codec, a few hundred to a few thousand rows each"""
    r = rng(seed)
    cases = []
    codecs = (O.DICT_COMPRESS, O.DELTA, O.SIMPLE_DICT)
    for k in range(count):
        data_type = ALL_TYPES[k % len(ALL_TYPES)]
        n = int(r.integers(1, 6000))
        cases.append(oracle_case(_column(r, data_type, n, int(r.integers(1, 400))), data_type, codecs[(k // 5) % 3]))
    return cases


def test_single_pack_calls_from_four_threads():
    """4 threads x 200 calls of the three single-pack entry points on mixed kinds: ixr_decode_pack for every pack, and the
    pair DataPack.decompress makes for dictionary packs (getDictStruct, then getDataFromDictStruct on what it returned)."""
    assert L.lib.ixr_init(0) == 0
    cases = _mixed_cases(4242, 40)
    structs = {}
    for i, c in enumerate(cases):
        if c.encode_type == O.DICT_COMPRESS:
            ds, consumed = O.get_dict_struct(c.pack, c.data_type, c.n)
            assert consumed == len(c.pack)
            structs[i] = ds
    errors = []

    def worker(tid):
        try:
            for k in range(200):
                i = (tid * 7 + k * 3) % len(cases)
                c = cases[i]
                out, consumed = L.decode_pack(c.pack, c.data_type, c.encode_type, c.n, c.data_size)
                if consumed != len(c.pack) or out.tobytes() != c.expect.tobytes():
                    errors.append("thread %d call %d: decode_pack differs on %s (%d)" % (tid, k, c.name, consumed))
                    return
                if i in structs and k % 2 == 0:
                    ds, consumed = L.get_dict_struct(c.pack, c.data_type, c.n)
                    if consumed != len(c.pack) or ds.tobytes() != structs[i].tobytes():
                        errors.append("thread %d call %d: get_dict_struct differs on %s (%d)" % (tid, k, c.name, consumed))
                        return
                    rows, written = L.get_data_from_dict_struct(ds, c.data_type, c.n, c.data_size)
                    if written < 0 or rows.tobytes() != c.expect.tobytes():
                        errors.append("thread %d call %d: get_data_from_dict_struct differs on %s (%d)" % (tid, k, c.name, written))
                        return
        except Exception as e:  # a thread must not die silently
            errors.append("thread %d: %r" % (tid, e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_two_plans_on_two_streams_at_once():
    """Two plans, each with its own arenas, scratch and stream, enqueued from two threads at the same time and run
    several times over: neither disturbs the other's rows or status."""
    assert L.lib.ixr_init(0) == 0
    batches = [_mixed_cases(11, 120), _mixed_cases(12, 90)]
    errors = []

    def worker(which):
        try:
            cases = batches[which]
            descs, arena, out_bytes = build_arena(cases)
            plan = L.Plan(0, descs, (8 << 20) if which else 0)  # the second plan in several groups
            stream = L.lib.ixr_stream_create(0)
            d_in, d_out = L.DeviceBuffer(0, len(arena)), L.DeviceBuffer(0, out_bytes)
            try:
                d_in.upload(arena)
                for rep in range(6):
                    d_out.upload(np.full(out_bytes, 0x5A + rep, dtype=np.uint8))
                    plan.run(d_in.ptr, len(arena), d_out.ptr, out_bytes, stream)
                    status, bad = plan.status(stream)
                    host = d_out.download()
                    for c, d, st in zip(cases, descs, status):
                        if st != len(c.pack) or host[d.out_off:d.out_off + c.data_size].tobytes() != c.expect.tobytes():
                            errors.append("plan %d run %d: %s differs (status %d)" % (which, rep, c.name, st))
                            return
            finally:
                plan.destroy()
                d_in.free()
                d_out.free()
                L.lib.ixr_stream_destroy(0, stream)
        except Exception as e:
            errors.append("plan %d: %r" % (which, e))

    threads = [threading.Thread(target=worker, args=(w,)) for w in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
