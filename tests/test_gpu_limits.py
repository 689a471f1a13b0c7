"""GPU parity tests (-m gpu) of the shapes opened by the larger row selection (topk.hip: up to 16384 keys sorted in LDS,
up to 65536 in a global scratch row; was 4096): nprobe above 4096 and range search on indexes with more than 4096
lists, against the oracle, bit for bit.

First run on hardware in round 3 (13 cases green)."""
import os

import numpy as np
import pytest

from conftest import assert_parity, gen_data
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def _gpu(ix):
    from knowhere_amd import GpuIndex
    return GpuIndex.from_data(ix, device=0)


@pytest.mark.parametrize("kind", [ob.IVF_FLAT, ob.IVF_SQ8], ids=["ivfflat", "ivfsq8"])
@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_nprobe_above_4096(port, kind, metric):
    nb, d, nlist = 30000, 16, 6000
    xb, xq = gen_data(nb, d, 42), gen_data(12, d, 44)
    ix = ob.make_index(port, kind, metric, xb, nlist=nlist)
    g = _gpu(ix)
    for k, nprobe in ((10, 5000), (100, nlist), (3, 4097)):
        Do, Io = port.search(ix, xq, k, nprobe)
        D, I = g.search(xq, k, nprobe)
        assert_parity(Do, Io, D, I, metric, f"kind={kind} k={k} nprobe={nprobe}")
    g.close()


@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_range_search_above_4096_lists(port, metric):
    nb, d, nlist = 30000, 16, 5000
    xb, xq = gen_data(nb, d, 42), gen_data(10, d, 44)
    ix = ob.make_index(port, ob.IVF_FLAT, metric, xb, nlist=nlist)
    g = _gpu(ix)
    D, _ = port.search(ix, xq, 40, nlist)
    for radius in (float(np.median(D[:, 5])), float(np.median(D[:, 39]))):
        for max_empty in (0, 2):
            exp = port.range_search(ix, xq, radius, max_empty)
            got = g.range_search(xq, np.float32(radius), max_empty)
            assert np.array_equal(exp[0], got[0]), "lims differ"
            assert np.array_equal(exp[1], got[1]), "ids differ (or are in a different order)"
            assert np.array_equal(exp[2].view(np.uint32), got[2].view(np.uint32)), "distances differ bitwise"
    g.close()


def test_nprobe_and_range_above_16384_lists(port):
    """more keys than the LDS sorts: the global-scratch sort of the row selection"""
    nb, d, nlist = 40000, 8, 20000
    xb, xq = gen_data(nb, d, 42), gen_data(6, d, 44)
    ix = ob.make_index(port, ob.IVF_FLAT, ob.L2, xb, nlist=nlist)
    g = _gpu(ix)
    for k, nprobe in ((10, 17000), (5, nlist)):
        Do, Io = port.search(ix, xq, k, nprobe)
        D, I = g.search(xq, k, nprobe)
        assert_parity(Do, Io, D, I, ob.L2, f"k={k} nprobe={nprobe}")
    D, _ = port.search(ix, xq, 40, nlist)
    radius = float(np.median(D[:, 20]))
    exp = port.range_search(ix, xq, radius, 2)
    got = g.range_search(xq, np.float32(radius), 2)
    assert np.array_equal(exp[0], got[0]) and np.array_equal(exp[1], got[1])
    assert np.array_equal(exp[2].view(np.uint32), got[2].view(np.uint32))
    g.close()


@pytest.mark.parametrize("M", [8, 16, 64])
@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_range_search_ivfpq_other_code_widths(port, monkeypatch, M, metric):
    """range.hip::pq_adc_dump_kernel: range search on IVF-PQ with m != 32"""
    from helpers import finish_ivfpq
    nb, d, nlist = 12000, 128, 48
    xb, xq = gen_data(nb, d, 42), gen_data(30, d, 44)
    ix = finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=nlist, M=M))
    g = _gpu(ix)
    D, _ = port.search(ix, xq, 40, nlist)
    for radius in (float(np.median(D[:, 5])), float(np.median(D[:, 39]))):
        for max_empty in (0, 2):
            exp = port.range_search(ix, xq, radius, max_empty)
            got = g.range_search(xq, np.float32(radius), max_empty)
            assert np.array_equal(exp[0], got[0]), "lims differ"
            assert np.array_equal(exp[1], got[1]), "ids differ (or are in a different order)"
            assert np.array_equal(exp[2].view(np.uint32), got[2].view(np.uint32)), "distances differ bitwise"
    g.close()
