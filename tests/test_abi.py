"""CPU tests: the C-ABI library loads and exports every symbol include/knhip.h declares; host-only
entry points (no GPU needed) behave; error convention."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, assert_parity, gen_data
from oracle import binding as ob


def _lib():
    from knowhere_amd import _lib
    return _lib


def test_library_built_and_loads():
    L = _lib().load()
    assert L.knhip_abi_version() == _lib().ABI_VERSION == 9


def test_exports_match_header():
    hdr = open(os.path.join(ROOT, "include", "knhip.h")).read()
    declared = set(re.findall(r"\b(knhip_[a-zA-Z0-9_]+)\s*\(", hdr))
    declared -= {"knhip_index", "knhip_desc"}
    L = _lib().load()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, f"libknhip.so does not export: {missing}"
    assert set(_lib().SYMBOLS) == declared


def test_shard_library_exports_match_header():
    """include/knhip_shards.h (the C++ multi-GPU host, knowhere_amd/host/shard_group.cc): every declared entry point is
    exported by libknhip_shards.so; without a device the group refuses loudly"""
    hdr = open(os.path.join(ROOT, "include", "knhip_shards.h")).read()
    declared = set(re.findall(r"\b(knhip_shard_group_[a-zA-Z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 5
    _lib().load()
    L = C.CDLL(os.path.join(ROOT, "knowhere_amd", "libknhip_shards.so"))
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, f"libknhip_shards.so does not export: {missing}"
    import torch
    if not torch.cuda.is_available():
        g = C.c_void_p()
        dev = (C.c_int32 * 1)(0)
        assert L.knhip_shard_group_create(C.c_int32(1), dev, C.c_int32(1), C.byref(g)) != 0


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from knowhere_amd import GpuIndex, KnhipError
    with pytest.raises(KnhipError):
        GpuIndex(0, 0, 16)  # no HIP device: must raise, never fall back to a CPU path


@pytest.mark.parametrize("metric", [ob.L2, ob.IP])
def test_merge_topk_host_vs_oracle(port, metric):
    from knowhere_amd.index import merge_topk_host
    r = np.random.default_rng(0)
    nshard, nq, k = 4, 33, 10
    D = r.random((nshard, nq, k), dtype=np.float32)
    D.sort(axis=2)
    if metric == ob.IP:
        D = D[:, :, ::-1].copy()
    I = r.permutation(nshard * nq * k).reshape(nshard, nq, k).astype(np.int64)
    I[1, :, 7:] = -1  # short shard
    D[1, :, 7:] = np.finfo(np.float32).max if metric == ob.L2 else -np.finfo(np.float32).max
    # duplicates across shards to exercise tie order
    D[2, :, 0] = D[0, :, 0]
    Do, Io = port.merge_topk(metric, D, I)
    Dg, Ig = merge_topk_host(metric, D, I)
    assert_parity(Do, Io, Dg, Ig, metric)
