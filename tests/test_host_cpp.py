"""Runs the C++ harness of the host layer (tests/cpp/host_test.cpp): the reference's DataPackTest-style life cycle,
the DPN record codec, the IntegratedSegment v8 file reader and the 1 M-row mixed segment scan, through the C ABI."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "cpp", "host_test")


def _run(mode, tmp_path):
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "cpp"), "-s"])
    r = subprocess.run([BIN, mode, str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "host_test %s failed:\n%s\n%s" % (mode, r.stdout[-2000:], r.stderr[-4000:])
    assert "0 failed" in r.stdout


def test_host_layer_cpu_part(tmp_path):
    _run("--cpu", tmp_path)


@pytest.mark.gpu
def test_host_layer_gpu_part(tmp_path):
    _run("--gpu", tmp_path)
