"""GPU parity tests (-m gpu) of the matrix-core ADC prefilter + exact finish of the IVF-PQ scan
(knowhere_amd/csrc/pq_filter.hip, pq_decode.hip): the prefilter path (KNHIP_PQF=1: whenever the shape allows), the exact kernels
(KNHIP_PQF=0) and the oracle must agree bit for bit -- distances AND ids."""
import os

import numpy as np
import pytest

from conftest import assert_parity, gen_data
from helpers import finish_ivfpq, sort_lists_by_id
from oracle import binding as ob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch


def _gpu(ix, **kw):
    from knowhere_amd import GpuIndex
    return GpuIndex.from_data(ix, device=0, **kw)


def _bitset(n, frac, seed):
    filt = np.random.default_rng(seed).random(n) < frac
    return np.packbits(filt, bitorder="little")


FORM = {"v": "half"}  # the filter form the module's tests run with (fixture below)


@pytest.fixture(autouse=True, params=["half", "int8", "decode"])
def form(request):
    """KNHIP_PQF_FORM: half = half-precision tables, 8 queries per unit (v_mfma_f32_16x16x32_f16); int8 = integer tables,
    16 queries per unit (v_mfma_i32_16x16x64_i8); decode = rows decoded once per (list, <= 128 queries), dense half-precision
    contraction (pq_decode.hip, v_mfma_f32_32x32x16_f16).  All must return the exact kernels' and the oracle's bits."""
    FORM["v"] = request.param
    return request.param


def _pair(monkeypatch, ix, guard=True):
    monkeypatch.setenv("KNHIP_PQF", "0")  # read when the lists are attached
    g0 = _gpu(ix)
    monkeypatch.setenv("KNHIP_PQF", "1")
    monkeypatch.setenv("KNHIP_PQF_GUARD", "1" if guard else "0")
    monkeypatch.setenv("KNHIP_PQF_FORM", FORM["v"])
    g1 = _gpu(ix)
    monkeypatch.delenv("KNHIP_PQF", raising=False)
    monkeypatch.delenv("KNHIP_PQF_GUARD", raising=False)
    monkeypatch.delenv("KNHIP_PQF_FORM", raising=False)
    return g0, g1


def _check(port, ix, g0, g1, xq, k, nprobe, metric, what, bs=None, nbits=0):
    Do, Io = port.search(ix, xq, k, nprobe, bs, nbits)
    D0, I0 = g0.search(xq, k, nprobe, bs, nbits)
    g1.profile_enable(True)
    g1.profile_reset()
    D1, I1 = g1.search(xq, k, nprobe, bs, nbits)
    p = g1.profile_get()
    g1.profile_enable(False)
    assert_parity(Do, Io, D1, I1, metric, f"{what}: prefilter vs oracle")
    assert np.array_equal(I0, I1) and np.array_equal(D0.view(np.uint32), D1.view(np.uint32)), f"{what}: prefilter vs exact"
    return p


@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_pqf_matches_exact_and_oracle(torch_cuda, port, monkeypatch, metric):
    nb, d, nlist = 60000, 128, 48
    xb, xq = gen_data(nb, d, 42), gen_data(150, d, 44)  # 150 queries: ragged 8-query units
    ix = finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=nlist, M=32))
    g0, g1 = _pair(monkeypatch, ix)
    used = 0
    for k, nprobe in ((10, 8), (1, 2), (100, 16), (10, nlist), (128, 5)):
        p = _check(port, ix, g0, g1, xq, k, nprobe, metric, f"metric={metric} k={k} nprobe={nprobe}")
        # (the test data is isotropic: where many rows are probed the selectivity guard hands the batch to the exact
        # kernel and the prefilter counters stay 0)
        assert p["mscan_queries"] + p["mscan_overflow_queries"] in (0, len(xq))
        used += p["mscan_queries"]
    assert used > 0, "the prefilter path never finished a query"
    bs = _bitset(nb, 0.4, 1)
    _check(port, ix, g0, g1, xq, 10, 8, metric, "bitset 40%", bs, nb)
    bs = _bitset(nb, 0.98, 2)  # closest lists hold fewer than k unfiltered rows: queries overflow -> exact fallback
    _check(port, ix, g0, g1, xq, 10, nlist, metric, "bitset 98%", bs, nb)
    _check(port, ix, g0, g1, xq[:3], 10, 8, metric, "nq=3")  # units with one or two pairs
    g0.close()
    g1.close()


def test_pqf_empty_lists_ties_and_ids(torch_cuda, port, monkeypatch):
    nb, d = 9000, 128
    xb, xq = gen_data(nb, d, 42), gen_data(70, d, 44)
    xb[100:160] = xb[7]  # exact duplicates: distance ties, broken by id
    ids = np.random.default_rng(5).permutation(nb).astype(np.int64) * 3 + 1
    ix = sort_lists_by_id(finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, ob.L2, xb, nlist=23, M=32, ids=ids)))
    for l in (0, 4):  # lists emptied by hand
        ix.list_codes[l] = ix.list_codes[l][:0]
        ix.list_ids[l] = ix.list_ids[l][:0]
    g0, g1 = _pair(monkeypatch, ix)
    for k, nprobe in ((10, 23), (64, 6), (3, 2), (4, 2), (8, 1), (2, 1)):  # (2^n with all probed lists empty: see mscan_finish_kernel)
        _check(port, ix, g0, g1, xq, k, nprobe, ob.L2, f"k={k} nprobe={nprobe}")
    g0.close()
    g1.close()


def test_pqf_overflow_goes_through_the_exact_kernel(torch_cuda, port, monkeypatch):
    """99.7 % of the ids filtered: no query finds k unfiltered rows in its sample, all are flagged and redone by the
    exact 4-query kernel over one-pair items"""
    nb, d = 6000, 128
    xb, xq = gen_data(nb, d, 42), gen_data(40, d, 44)
    ix = finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, ob.L2, xb, nlist=100, M=32))
    g0, g1 = _pair(monkeypatch, ix, guard=False)
    p = _check(port, ix, g0, g1, xq, 100, 64, ob.L2, "short lists")
    assert p["mscan_queries"] > 0
    bs = _bitset(nb, 0.997, 7)
    p = _check(port, ix, g0, g1, xq, 10, 100, ob.L2, "99.7 % filtered", bs, nb)
    # (18 unfiltered rows in all: a query whose sample happens to hold k of them gets a bound and is finished by the
    # prefilter path; on hardware that is 1 query of 40)
    assert p["mscan_overflow_queries"] >= len(xq) - 4
    assert p["mscan_queries"] + p["mscan_overflow_queries"] == len(xq)
    g0.close()
    g1.close()


@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_pqf_large_k(torch_cuda, port, monkeypatch, metric):
    """128 < k <= 1024 (Knowhere's refine asks the first stage for k * refine_k candidates): sample, filter and finish
    take any k; the queries that overflow are redone by the systolic exact kernel over one-pair items.  Bits equal to the
    exact kernels and the oracle, with the prefilter really used, and with the overflow fallback forced by a bitset."""
    nb, d, nlist = 60000, 128, 48
    xb, xq = gen_data(nb, d, 42), gen_data(70, d, 44)
    ix = finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=nlist, M=32))
    g0, g1 = _pair(monkeypatch, ix, guard=False)
    used = 0
    for k, nprobe in ((200, 8), (1000, 4), (1024, nlist), (129, 2)):
        p = _check(port, ix, g0, g1, xq, k, nprobe, metric, f"large k metric={metric} k={k} nprobe={nprobe}")
        assert p["mscan_queries"] + p["mscan_overflow_queries"] == len(xq)
        used += p["mscan_queries"]
    assert used > 0
    bs = _bitset(nb, 0.995, 7)  # ~300 unfiltered rows: no query finds 200 of them in its sample
    p = _check(port, ix, g0, g1, xq, 200, nlist, metric, "large k, 99.5 % filtered", bs, nb)
    assert p["mscan_overflow_queries"] > 0
    g0.close()
    g1.close()


@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_pqf_retry_round(torch_cuda, port, monkeypatch, metric):
    """a tiny candidate capacity makes most queries overflow with candidates in hand: retried as one-query units with the
    exact k-th of those candidates as their bound"""
    nb, d, nlist = 40000, 128, 40
    xb, xq = gen_data(nb, d, 42), gen_data(90, d, 44)
    ix = finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=nlist, M=32))
    monkeypatch.setenv("KNHIP_MSCAN_CAP", "26")  # (>= 2 (k + 1): the search runs for k + 1 results)
    g0, g1 = _pair(monkeypatch, ix, guard=False)  # (the selectivity guard would hand these batches to the exact kernel)
    for k, nprobe in ((10, 16), (3, nlist), (12, 8)):
        p = _check(port, ix, g0, g1, xq, k, nprobe, metric, f"retry metric={metric} k={k} nprobe={nprobe}")
        assert p["mscan_queries"] + p["mscan_overflow_queries"] == len(xq)
    bs = _bitset(nb, 0.4, 1)
    _check(port, ix, g0, g1, xq, 10, 16, metric, "retry + bitset", bs, nb)
    g0.close()
    g1.close()


def test_pqd_parked_records_overflow_route(torch_cuda, port, monkeypatch):
    """decode form (pq_decode.hip): a wave parks passing lanes in 192 LDS records, then in its share of a global region; when
    both are full the lane's query takes the overflow route (retry with the bound of what it gathered, else the exact
    kernels).  KNHIP_PQD_SPILL=4 leaves one global record per wave, k = 100 on isotropic data with the guard off lets
    hundreds of rows of a list pass: the route is taken by most queries and the answer stays the oracle's."""
    if FORM["v"] != "decode":
        pytest.skip("the decode form's record regions")
    nb, d = 120000, 128
    xb, xq = gen_data(nb, d, 42), gen_data(300, d, 44)
    ix = finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, ob.L2, xb, nlist=40, M=32))
    monkeypatch.setenv("KNHIP_PQD_SPILL", "4")
    g0, g1 = _pair(monkeypatch, ix, guard=False)
    monkeypatch.delenv("KNHIP_PQD_SPILL")
    for k, nprobe in ((100, 8), (10, 4), (128, 40)):
        p = _check(port, ix, g0, g1, xq, k, nprobe, ob.L2, f"record overflow k={k} nprobe={nprobe}")
        assert p["mscan_queries"] + p["mscan_overflow_queries"] == len(xq)
    assert p["mscan_overflow_queries"] > 0, "the overflow route was not taken"
    bs = _bitset(nb, 0.5, 3)
    _check(port, ix, g0, g1, xq, 100, 8, ob.L2, "record overflow + bitset", bs, nb)
    g0.close()
    g1.close()


@pytest.mark.parametrize("nq", [127, 129, 130, 257])
def test_pqf_exact_fallback_item_counts_off_the_multiple_of_eight(torch_cuda, port, monkeypatch, nq):
    """k = 500, two short probed lists, 90 % of the ids filtered: no query finds k rows in its sample, every query is flagged and its (query, list)
    pairs become one-pair items of the systolic exact kernel (k > 128) -- 2 nq items, here NOT a multiple of 8.  That kernel
    spreads its items over the blocks [0, round_up(items, 8)) (xcd_item): launched with one block per pair, the last up to
    seven items had no block and their partial lists stayed unwritten (whatever the freshly allocated scratch held was merged:
    found by tests/test_gpu_pqd_fuzz.py in round 6, present since the prefilter's k > 128 fallback of round 3).  A FRESH index
    per case: a second search of the same shape finds the first one's correct partial lists in the scratch."""
    nb, d, nlist = 3000, 128, 40
    xb = gen_data(nb, d, 42)
    xq = gen_data(nq, d, 44)
    ix = finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, ob.IP, xb, nlist=nlist, M=32))
    bs = _bitset(nb, 0.9, 5)  # (a tenth of the rows left: no sample holds k of them -> every query is flagged)
    g0, g1 = _pair(monkeypatch, ix, guard=False)
    p = _check(port, ix, g0, g1, xq, 500, 2, ob.IP, f"one-pair items, nq={nq}", bs, nb)
    assert p["mscan_overflow_queries"] >= nq - 2
    g0.close()
    g1.close()


def test_pqf_headline_shape_long_lists(torch_cuda, port, monkeypatch):
    """d = 128, m = 32, lists of ~3000 codes (many windows per wave), batch large enough for full 8-query units"""
    nb, d = 200000, 128
    xb, xq = gen_data(nb, d, 42), gen_data(400, d, 44)
    ix = finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, ob.L2, xb, nlist=64, M=32))
    g0, g1 = _pair(monkeypatch, ix)
    p = _check(port, ix, g0, g1, xq, 10, 16, ob.L2, "headline shape")
    p = _check(port, ix, g0, g1, xq, 100, 32, ob.L2, "headline shape k=100")
    g0.close()
    g1.close()
    g0, g1 = _pair(monkeypatch, ix, guard=False)  # isotropic data: only without the guard does the prefilter take this shape
    p = _check(port, ix, g0, g1, xq, 10, 16, ob.L2, "headline shape, guard off")
    assert p["mscan_queries"] + p["mscan_overflow_queries"] == len(xq)
    g0.close()
    g1.close()
