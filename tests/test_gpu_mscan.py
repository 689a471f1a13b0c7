"""GPU parity tests (-m gpu) of the MFMA prefilter + exact finish of the IVF-Flat / IVF-SQ8 list scans
(knowhere_amd/csrc/mfma_scan.hip): KNHIP_MSCAN=1 (always), =0 (the exact VALU kernels) and the oracle must agree
bit for bit -- distances AND ids -- for every metric, ragged dimensions, k below and above a wave, bitsets, empty
lists, and when the candidate lists overflow (the flagged queries are redone by the exact kernels)."""
import numpy as np
import pytest

from conftest import assert_parity, gen_data
from helpers import sort_lists_by_id
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


def _pair(monkeypatch, ix):
    monkeypatch.setenv("KNHIP_MSCAN", "0")  # read when the lists are attached
    g0 = _gpu(ix)
    monkeypatch.setenv("KNHIP_MSCAN", "1")
    g1 = _gpu(ix)
    return g0, g1


def _check(port, ix, g0, g1, xq, k, nprobe, metric, what, bs=None, nbits=0):
    Do, Io = port.search(ix, xq, k, nprobe, bs, nbits)
    D0, I0 = g0.search(xq, k, nprobe, bs, nbits)
    g1.profile_enable(True)
    g1.profile_reset()
    D1, I1 = g1.search(xq, k, nprobe, bs, nbits)
    p = g1.profile_get()
    g1.profile_enable(False)
    assert_parity(Do, Io, D1, I1, metric, f"{what}: mscan vs oracle")
    assert np.array_equal(I0, I1) and np.array_equal(D0.view(np.uint32), D1.view(np.uint32)), f"{what}: mscan vs exact"
    return p


@pytest.mark.parametrize("kind", [ob.IVF_FLAT, ob.IVF_SQ8], ids=["flat", "sq8"])
@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_mscan_matches_exact_and_oracle(torch_cuda, port, monkeypatch, kind, metric):
    nb, d, nlist = 40000, 64, 50
    xb, xq = gen_data(nb, d, 42), gen_data(150, d, 44)  # 150 queries: ragged 64- and 32-query units
    ix = ob.make_index(port, kind, metric, xb, nlist=nlist)
    g0, g1 = _pair(monkeypatch, ix)
    used = 0
    for k, nprobe in ((10, 8), (1, 2), (100, 16), (10, nlist), (200, 5)):
        p = _check(port, ix, g0, g1, xq, k, nprobe, metric, f"kind={kind} k={k} nprobe={nprobe}")
        assert p["mscan_queries"] + p["mscan_overflow_queries"] == len(xq)
        used += p["mscan_queries"]
    assert used > 0, "the MFMA path never finished a query"
    bs = _bitset(nb, 0.4, 1)
    _check(port, ix, g0, g1, xq, 10, 8, metric, "bitset 40%", bs, nb)
    bs = _bitset(nb, 0.98, 2)  # closest lists hold fewer than k unfiltered rows: queries overflow -> exact fallback
    p = _check(port, ix, g0, g1, xq, 10, nlist, metric, "bitset 98%", bs, nb)
    # few queries: units with one or two pairs
    _check(port, ix, g0, g1, xq[:3], 10, 8, metric, "nq=3")
    g0.close()
    g1.close()


@pytest.mark.parametrize("kind", [ob.IVF_FLAT, ob.IVF_SQ8], ids=["flat", "sq8"])
def test_mscan_ragged_dims_empty_lists_and_ties(torch_cuda, port, monkeypatch, kind):
    for d in (30, 5, 100):  # d not a multiple of the 16- / 32-dim step, of 4, odd chunk counts
        nb = 6000
        xb, xq = gen_data(nb, d, 42), gen_data(70, d, 44)
        xb[100:160] = xb[7]  # exact duplicates: distance ties, broken by id
        ids = np.random.default_rng(5).permutation(nb).astype(np.int64) * 3 + 1
        ix = sort_lists_by_id(ob.make_index(port, kind, ob.L2, xb, nlist=23, ids=ids))  # (shuffled ids, stored in id order)
        for l in (0, 4):  # lists emptied by hand
            ix.list_codes[l] = ix.list_codes[l][:0]
            ix.list_ids[l] = ix.list_ids[l][:0]
        g0, g1 = _pair(monkeypatch, ix)
        for k, nprobe in ((10, 23), (64, 6), (3, 2), (4, 2), (8, 1), (2, 1)):  # (2^n with all probed lists empty: see mscan_finish_kernel)
            _check(port, ix, g0, g1, xq, k, nprobe, ob.L2, f"kind={kind} d={d} k={k} nprobe={nprobe}")
        g0.close()
        g1.close()


def test_mscan_overflow_goes_through_the_exact_kernels(torch_cuda, port, monkeypatch):
    """no bound for a query = fewer than k unfiltered rows in its sample (the probes in coarse order until enough rows
    are covered): with 99.7 % of the ids filtered every query is flagged, its (query, list) pairs are compacted into
    one-query items for the exact kernels, and the result is still the oracle's.  Tiny lists alone (12 rows each) are
    NOT such a case: the sample then spans several lists."""
    nb, d = 3000, 32
    xb, xq = gen_data(nb, d, 42), gen_data(40, d, 44)
    ix = ob.make_index(port, ob.IVF_FLAT, ob.L2, xb, nlist=250)
    g0, g1 = _pair(monkeypatch, ix)
    p = _check(port, ix, g0, g1, xq, 100, 64, ob.L2, "tiny lists")
    assert p["mscan_queries"] > 0
    _check(port, ix, g0, g1, xq, 2, 64, ob.L2, "tiny lists, small k")
    bs = _bitset(nb, 0.997, 7)
    p = _check(port, ix, g0, g1, xq, 10, 250, ob.L2, "99.7 % filtered", bs, nb)
    assert p["mscan_overflow_queries"] == len(xq)
    for kind, metric in ((ob.IVF_SQ8, ob.IP), (ob.IVF_SQ8, ob.L2), (ob.IVF_FLAT, ob.IP)):
        ix2 = ob.make_index(port, kind, metric, xb, nlist=250)
        h0, h1 = _pair(monkeypatch, ix2)
        p = _check(port, ix2, h0, h1, xq, 10, 250, metric, f"99.7 % filtered kind={kind} metric={metric}", bs, nb)
        assert p["mscan_overflow_queries"] == len(xq)
        h0.close()
        h1.close()
    g0.close()
    g1.close()


def test_mscan_config_shapes(torch_cuda, port, monkeypatch):
    """BASELINE.json configs[1] (IVF-Flat L2 d=128) and configs[4] (IVF-SQ8 IP d=768, int8-valued rows) at sizes the
    oracle finishes in seconds, through the MFMA path"""
    xb, xq = gen_data(60000, 128, 42), gen_data(200, 128, 44)
    ix = ob.make_index(port, ob.IVF_FLAT, ob.L2, xb, nlist=128)
    g0, g1 = _pair(monkeypatch, ix)
    p = _check(port, ix, g0, g1, xq, 10, 64, ob.L2, "IVF-Flat L2 d=128 nprobe=64")
    assert p["mscan_queries"] > 0
    g0.close()
    g1.close()
    r = np.random.default_rng(5)
    xb = r.integers(-128, 128, (20000, 768)).astype(np.float32)
    xq = r.integers(-128, 128, (100, 768)).astype(np.float32)
    for metric in (ob.IP, ob.L2):
        ix = ob.make_index(port, ob.IVF_SQ8, metric, xb, nlist=48)
        g0, g1 = _pair(monkeypatch, ix)
        for k, nprobe in ((10, 16), (100, 48)):
            p = _check(port, ix, g0, g1, xq, k, nprobe, metric, f"SQ8 d=768 metric={metric} k={k}")
        g0.close()
        g1.close()


@pytest.mark.parametrize("kind,metric", [(ob.IVF_FLAT, ob.L2), (ob.IVF_FLAT, ob.IP), (ob.IVF_SQ8, ob.IP), (ob.IVF_SQ8, ob.L2)],
                         ids=["flat-l2", "flat-ip", "sq8-ip", "sq8-l2"])
def test_mscan_retry_round(torch_cuda, port, monkeypatch, kind, metric):
    """a tiny candidate capacity (KNHIP_MSCAN_CAP) makes most queries overflow with candidates in hand: they are retried
    with the exact k-th of those candidates as their bound (one-query units of the filter kernel) and finished from the
    second, short candidate list; only what overflows again reaches the exact kernels.  Results stay the oracle's."""
    nb, d, nlist = 30000, 64, 40
    xb, xq = gen_data(nb, d, 42), gen_data(90, d, 44)
    ix = ob.make_index(port, kind, metric, xb, nlist=nlist)
    monkeypatch.setenv("KNHIP_MSCAN_CAP", "26")  # (>= 2 (k + 1): the search runs for k + 1 results, conftest.assert_parity)
    g0, g1 = _pair(monkeypatch, ix)
    for k, nprobe in ((10, 16), (3, nlist), (12, 8)):
        p = _check(port, ix, g0, g1, xq, k, nprobe, metric, f"retry kind={kind} metric={metric} k={k} nprobe={nprobe}")
        assert p["mscan_queries"] + p["mscan_overflow_queries"] == len(xq)
    bs = _bitset(nb, 0.4, 1)
    _check(port, ix, g0, g1, xq, 10, 16, metric, "retry + bitset", bs, nb)
    g0.close()
    g1.close()


@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
@pytest.mark.parametrize("d", [128, 320, 36], ids=["d128", "d320", "d36"])
def test_mscan_flat_filter_on_the_bf16_pipe(torch_cuda, port, monkeypatch, d, metric):
    """the IVF-Flat filter pass on split-bf16 operands (mfma_scan_bf16.hip; round 5) against the fp32 filter
    (KNHIP_MSCAN_FLAT=fp32) and the oracle: units of 128 queries in full, partly filled and one-tile units (300 queries
    probing every list), the 64-query form of wide rows (d = 320), a ragged dimension, a bitset, and rows of
    different magnitudes."""
    nb, nlist, nq = 24000, 24, 300
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    xb[::97] *= 3.0  # a few rows longer than the rest (the bound scales with the LARGEST row norm of the index)
    ix = ob.make_index(port, ob.IVF_FLAT, metric, xb, nlist=nlist)
    monkeypatch.setenv("KNHIP_MSCAN", "1")
    monkeypatch.setenv("KNHIP_MSCAN_FLAT", "fp32")
    g32 = _gpu(ix)
    monkeypatch.delenv("KNHIP_MSCAN_FLAT")
    gb = _gpu(ix)
    bs = _bitset(nb, 0.5, 3)
    for k, nprobe, b, nbits in ((10, nlist, None, 0), (100, 8, None, 0), (1, 3, None, 0), (10, nlist, bs, nb)):
        Do, Io = port.search(ix, xq, k, nprobe, b, nbits)
        D0, I0 = g32.search(xq, k, nprobe, b, nbits)
        gb.profile_enable(True)
        gb.profile_reset()
        D1, I1 = gb.search(xq, k, nprobe, b, nbits)
        p = gb.profile_get()
        assert p["mscan_queries"] + p["mscan_overflow_queries"] == nq and p["mscan_queries"] > 0
        assert_parity(Do, Io, D1, I1, metric, f"bf16 filter d={d} k={k} nprobe={nprobe} bitset={b is not None}")
        assert np.array_equal(I0, I1) and np.array_equal(D0.view(np.uint32), D1.view(np.uint32))
    g32.close()
    gb.close()


@pytest.mark.parametrize("kind", [ob.IVF_FLAT, ob.IVF_SQ8], ids=["flat", "sq8"])
def test_mscan_finish_prunes_a_long_candidate_list(torch_cuda, port, monkeypatch, kind):
    """an unlucky sample: seven of eight rows among the first 1024 of every list are filtered, so tau comes from 128 rows per
    list and ~9 % of the 100k rows pass the filter -- more candidates per query than the finish kernel's pruning holds in
    registers (4096): the bound comes from the head of the list, the rest is streamed through the same test (SQ8: with the
    largest emission eps the units published for the query)."""
    nb, d, nlist, nq = 100_000, 32, 4, 40
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    ix = sort_lists_by_id(ob.make_index(port, kind, ob.L2, xb, nlist=nlist))
    filt = np.zeros(nb, bool)
    for l in range(nlist):
        head = np.asarray(ix.list_ids[l][:1024])
        filt[head[np.arange(head.size) % 8 != 0]] = True
    bs = np.packbits(filt, bitorder="little")
    monkeypatch.setenv("KNHIP_MSCAN", "1")
    monkeypatch.setenv("KNHIP_MSCAN_CAP", "16384")
    g = _gpu(ix)
    for k in (10, 100):
        Do, Io = port.search(ix, xq, k, nlist, bs, nb)
        g.profile_enable(True)
        g.profile_reset()
        D, I = g.search(xq, k, nlist, bs, nb)
        p = g.profile_get()
        assert_parity(Do, Io, D, I, ob.L2, f"long candidate lists k={k}")
        if k == 10:
            assert p["mscan_queries"] == nq and p["mscan_candidates"] > 4096 * nq, p
    g.close()

