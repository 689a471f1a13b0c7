"""GPU parity at scale (-m gpu): the matrix-core IVF-PQ prefilter (knowhere_amd/csrc/pq_filter.hip, both forms) against the
reference's own FAISS (oracle/_ref, scalar build, driven Knowhere-style) on a 10M x 128 index of the bench's data generator --
2048 queries, with and without a 40 % bitset, k = 10 and k = 100, bit for bit, no licence (until now the
largest oracle-checked -m gpu IVF-PQ case was 200k rows and parity at the headline scale lived only in bench.py).

The prefilter's exactness rests on error bounds (tests/test_pq_filter_bound.py); a hole in one shows up only where many rows
sit within the bound of a query's threshold.  The second test builds that on purpose: a codebook with one huge entry per
sub-quantizer inflates every query's table range, hence the integer form's step and eps, until most of a list passes the
filter -- capacity overflows, retry round and exact fallback included -- and duplicated rows put exact ties on the k-th
boundary."""
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


def _bitset(n, frac, seed):
    filt = np.random.default_rng(seed).random(n) < frac
    return np.packbits(filt, bitorder="little")


def _forms(monkeypatch, make):
    """the same lists behind the two forms of the filter (the switches are read when the lists are attached)"""
    out = {}
    for form in ("decode", "int8", "half"):
        monkeypatch.setenv("KNHIP_PQF", "1")
        monkeypatch.setenv("KNHIP_PQF_FORM", form)
        monkeypatch.setenv("KNHIP_PQF_GUARD", "0")
        out[form] = make()
    for v in ("KNHIP_PQF", "KNHIP_PQF_FORM", "KNHIP_PQF_GUARD"):
        monkeypatch.delenv(v, raising=False)
    return out


def test_ivfpq_10m_prefilter_equals_the_reference_build(torch_cuda, monkeypatch):
    torch = torch_cuda
    from knowhere_amd import build as kb
    from knowhere_amd import index as kidx
    nb, d, nlist, nprobe, nq = 10_000_000, 128, 4096, 64, 2048
    spec = kb.DataSpec(nb, d, kind="mixture", seed=42, ncenter=65536, sigma=0.35)
    built = kb.build_ivf(spec, kidx.IVF_PQ, kidx.L2, nlist, 32, device="cuda:0", train_per_centroid=64, niter=10)
    xq_t = kb.queries(spec, nq, torch.device("cuda:0"), seed=44)
    xq = xq_t.cpu().numpy()
    gs = _forms(monkeypatch, lambda: built.to_gpu_index(device=0))
    ix = built.export(ob.IndexData)  # the same index BYTES for the reference (0.4 GB of codes + ids on the host)
    if ob.Ref.available():
        ref = ob.Ref("scalar")
        h = ref.from_data(ix)
        import os
        nth = max(1, min(16, len(os.sched_getaffinity(0))))

        def oracle(k, bs, nbits):
            return ref.search(h, xq, k, nprobe, bs, nbits, nthreads=nth)
    else:  # (the GPU box normally carries oracle/_ref; the plain-C port is single-threaded: fewer queries)
        port = ob.Port()
        ix.use_precomputed_table = 1
        ix.precomputed_table = port.pq_precompute_table(ix.d, ix.M, 8, ix.centroids, ix.pq_centroids)
        nq = 192
        xq = xq[:nq]

        def oracle(k, bs, nbits):
            return port.search(ix, xq, k, nprobe, bs, nbits)
    bs = _bitset(nb, 0.4, 3)
    for k in (10, 100):
        for b, nbits in ((None, 0), (bs, nb)):
            Do, Io = oracle(k, b, nbits)
            for form, g in gs.items():
                g.profile_enable(True)
                g.profile_reset()
                D, I = g.search(xq[:nq], k, nprobe, b, nbits)
                p = g.profile_get()
                assert p["pq_filter_form"] == {"half": 1, "int8": 2, "decode": 3}[form], (form, p["pq_filter_form"])
                assert p["mscan_queries"] + p["mscan_overflow_queries"] == nq
                assert p["tie_anomalies"] == 0
                assert_parity(Do, Io, D, I, ob.L2, f"10M IVF-PQ, {form} form, k={k}, bitset={b is not None}")
    for g in gs.values():
        g.close()


@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_int8_lattice_bound_under_a_huge_table_entry(torch_cuda, monkeypatch, metric):
    port = ob.Port()
    nb, d, nlist, nq = 120_000, 128, 32, 400
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    xb[5000:5400] = xb[17]  # duplicated rows: identical codes, exact ties (also on the k-th boundary)
    ix = ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=nlist, M=32, nbits=8)
    # one far-away entry per sub-quantizer: every table's range is ~1000 x the spread of the entries the codes really
    # use, so the int8 step (range / 254) exceeds the differences between neighbouring rows and eps covers most of a list
    pq = ix.pq_centroids.reshape(32, 256, 4).copy()
    unused = np.ones((32, 256), bool)
    for l in range(nlist):
        c = ix.list_codes[l]
        for m in range(32):
            unused[m, np.unique(c[:, m])] = False
    for m in range(32):
        free = np.flatnonzero(unused[m])
        c0 = int(free[0]) if free.size else 255
        # (no free code: a few rows then really carry the huge entry -- fine, both sides see the same bytes)
        pq[m, c0] = 3000.0 if m % 2 else -3000.0
    ix.pq_centroids = np.ascontiguousarray(pq.reshape(ix.pq_centroids.shape), np.float32)
    ix.precomputed_table = None  # (derived from the codebook: recomputed by finish_ivfpq)
    ix = sort_lists_by_id(finish_ivfpq(port, ix))
    from knowhere_amd import GpuIndex
    gs = _forms(monkeypatch, lambda: GpuIndex.from_data(ix, device=0))
    bs = _bitset(nb, 0.3, 9)
    for k, nprobe in ((10, 8), (100, 16), (1, 32)):
        for b, nbits in ((None, 0), (bs, nb)):
            Do, Io = port.search(ix, xq, k, nprobe, b, nbits)
            for form, g in gs.items():
                g.profile_enable(True)
                g.profile_reset()
                D, I = g.search(xq, k, nprobe, b, nbits)
                p = g.profile_get()
                assert p["mscan_queries"] + p["mscan_overflow_queries"] == nq
                assert p["tie_anomalies"] == 0
                assert_parity(Do, Io, D, I, metric, f"huge table entry, {form} form, k={k} nprobe={nprobe} "
                                                    f"bitset={b is not None}")
    for g in gs.values():
        g.close()
