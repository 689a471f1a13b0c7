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
    assert L.knhip_abi_version() == 3


def test_exports_match_header():
    hdr = open(os.path.join(ROOT, "include", "knhip.h")).read()
    declared = set(re.findall(r"\b(knhip_[a-zA-Z0-9_]+)\s*\(", hdr))
    declared -= {"knhip_index", "knhip_desc"}
    L = _lib().load()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, f"libknhip.so does not export: {missing}"
    assert set(_lib().SYMBOLS) == declared


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
