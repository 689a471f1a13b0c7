"""The C++ Knowhere IndexNode above the C ABI (knowhere_amd/host): builds and loads on CPU; on the
GPU the reference's own GPU search test flow (tests/cpp/test_hip_index.cc, a re-run of reference
tests/ut/test_gpu_search.cc through IndexFactory / Index::Build / Search / Serialize) must pass."""
import ctypes
import os
import subprocess

import pytest

from conftest import ROOT

HOST = os.path.join(ROOT, "knowhere_amd", "host")


def test_node_library_builds_and_loads():
    subprocess.check_call(["make", "-s", "-C", HOST])
    lib = ctypes.CDLL(os.path.join(ROOT, "knowhere_amd", "libknowhere_hip_node.so"))
    assert lib is not None
    assert os.path.exists(os.path.join(HOST, "test_hip_index"))


@pytest.mark.gpu
def test_reference_gpu_search_flow_through_plugin_api():
    exe = os.path.join(HOST, "test_hip_index")
    assert os.path.exists(exe), "build with __graft_entry__.build()"
    p = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    print(p.stdout[-4000:])
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-2000:]
