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


def test_node_fails_loudly_without_a_gpu():
    """no CPU fallback: on a box without a device Build reports Status::cuda_runtime_error (22, reused for HIP,
    SURVEY.md 8b) and Search on the unbuilt index Status::empty_index (6) -- it never pretends to work"""
    import numpy as np
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    L = ctypes.CDLL(os.path.join(ROOT, "knowhere_amd", "libknowhere_hip_node.so"))
    L.knhip_node_create.restype = ctypes.c_void_p
    assert L.knhip_node_create(b"NO_SUCH_INDEX") is None
    h = L.knhip_node_create(b"GPU_HIP_IVF_FLAT")
    assert h
    x = np.random.default_rng(0).random((500, 16), dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    rc = L.knhip_node_build(ctypes.c_void_p(h), x.ctypes.data_as(fp), ctypes.c_int64(500), ctypes.c_int64(16),
                            b"metric_type=L2;nlist=4")
    assert rc == 22
    ids, dis = np.zeros((5, 1), np.int64), np.zeros((5, 1), np.float32)
    rc = L.knhip_node_search(ctypes.c_void_p(h), x.ctypes.data_as(fp), ctypes.c_int64(5), ctypes.c_int64(16), b"k=1",
                             None, ctypes.c_int64(0), ctypes.c_int64(1),
                             ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), dis.ctypes.data_as(fp))
    assert rc == 6
    L.knhip_node_destroy(ctypes.c_void_p(h))


@pytest.mark.gpu
def test_reference_gpu_search_flow_through_plugin_api():
    exe = os.path.join(HOST, "test_hip_index")
    assert os.path.exists(exe), "build with __graft_entry__.build()"
    p = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    print(p.stdout[-4000:])
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-2000:]
