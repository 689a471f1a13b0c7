"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the oracle on the
same seeded inputs and against the committed golden fixtures.  Bar: distances bit-equal, ids
equal in canonical order (licensed: exact ties at the k-th boundary) -- see conftest.assert_parity.
"""
import os

import numpy as np
import pytest

from conftest import assert_parity, gen_data
from helpers import finish_ivfpq, golden_files, load_golden, sort_lists_by_id
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
    bs = np.zeros((n + 7) // 8, np.uint8)
    for i in np.nonzero(filt)[0]:
        bs[i >> 3] |= 1 << (i & 7)
    return bs


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", golden_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_golden(torch_cuda, port, path):
    ix, xq, cases = load_golden(path)
    g = _gpu(ix)
    for c in cases:
        D, I = g.search(xq, c["k"], c["nprobe"], c["bitset"], c["nbits"])
        assert_parity(c["D"], c["I"], D, I, ix.metric, f"{os.path.basename(path)} k={c['k']} nprobe={c['nprobe']}")
    g.close()


@pytest.mark.parametrize("metric", [ob.L2, ob.IP])
@pytest.mark.parametrize("nb,d", [(5000, 32), (20000, 128), (777, 30), (64, 4), (65, 5)])
def test_brute_force(torch_cuda, port, metric, nb, d):
    xb, xq = gen_data(nb, d, 42), gen_data(37, d, 44)
    ix = ob.make_index(port, ob.FLAT, metric, xb)
    g = _gpu(ix)
    for k in (1, 10, 64, 100, 300):
        k = min(k, 1024)
        Do, Io = port.search(ix, xq, k)
        D, I = g.search(xq, k)
        assert_parity(Do, Io, D, I, metric, f"BF nb={nb} d={d} k={k}")
    bs = _bitset(nb, 0.4, 1)
    Do, Io = port.search(ix, xq, 10, bitset=bs)
    D, I = g.search(xq, 10, bitset=bs, nbits=nb)
    assert_parity(Do, Io, D, I, metric, "BF bitset")
    g.close()


@pytest.mark.parametrize("metric", [ob.L2, ob.IP])
def test_brute_force_on_the_matrix_cores(torch_cuda, port, metric, monkeypatch):
    """BRUTE_FORCE with a batch of queries and no bitset runs the coarse quantizer's machinery over the base rows (bf16
    prefilter on the matrix pipe, exact re-rank, certificate; knhip_api.hip::bf_mfma_batch): same bits and order as the exact
    row scan (KNHIP_BF=exact) and as the oracle; 140000 rows = two chunks with a ragged last one; duplicated rows put ties
    inside the lists; k up to 300 (k >= 100: the reference's reservoir -- canonical answer, licensed as for the row scan)."""
    from knowhere_amd import GpuIndex
    nb, d, nq = 140_000, 96, 200
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    xb[5000:5040] = xb[17]
    xb[131070:131080] = xb[131000]  # (across the chunk boundary)
    g1 = GpuIndex(0, metric, d)
    g1.add_vectors(xb, id_offset=1000)
    monkeypatch.setenv("KNHIP_BF", "exact")  # read when the rows are added
    g0 = GpuIndex(0, metric, d)
    g0.add_vectors(xb, id_offset=1000)
    monkeypatch.delenv("KNHIP_BF")
    ix = ob.make_index(port, ob.FLAT, metric, xb)
    g1.profile_enable(True)
    for k in (1, 10, 99, 300):
        g1.profile_reset()
        D1, I1 = g1.search(xq, k)
        assert g1.profile_get()["pq_filter_form"] == 10, "the matrix-core path did not run"
        D0, I0 = g0.search(xq, k)
        assert np.array_equal(I0, I1) and np.array_equal(D0.view(np.uint32), D1.view(np.uint32)), f"k={k}: MFMA path vs row scan"
        Do, Io = port.search(ix, xq[:40], k)
        assert_parity(Do, Io + 1000, D1[:40], I1[:40], metric, f"BF on the matrix cores k={k}", licensed_ties=k >= 100)
    # a bitset, or a handful of queries: the row scan
    bs = _bitset(nb + 1000, 0.4, 1)  # (bits are indexed by id = row + 1000)
    g1.profile_reset()
    D1, I1 = g1.search(xq, 10, bitset=bs, nbits=nb + 1000)
    assert g1.profile_get()["pq_filter_form"] == 0
    D0, I0 = g0.search(xq, 10, bitset=bs, nbits=nb + 1000)
    assert np.array_equal(I0, I1) and np.array_equal(D0.view(np.uint32), D1.view(np.uint32))
    g0.close()
    g1.close()
    # four chunks: every chunk but the first takes the running k-th best (widened by the prefilter's eps) as its selection
    # bound and makes ONE pass; rows EQUAL to that k-th -- copies planted in the later chunks -- must still be found
    nb = 400_000
    xb = gen_data(nb, d, 43)
    xb[150_000:150_004] = xb[9]
    xb[390_000:390_003] = xb[9]
    xq2 = np.concatenate([xb[9:10] + 0.001, gen_data(nq, d, 45)]).astype(np.float32)
    g1 = GpuIndex(0, metric, d)
    g1.add_vectors(xb)
    monkeypatch.setenv("KNHIP_BF", "exact")
    g0 = GpuIndex(0, metric, d)
    g0.add_vectors(xb)
    monkeypatch.delenv("KNHIP_BF")
    g1.profile_enable(True)
    for k in (5, 10, 300):
        g1.profile_reset()
        D1, I1 = g1.search(xq2, k)
        assert g1.profile_get()["pq_filter_form"] == 10
        D0, I0 = g0.search(xq2, k)
        assert np.array_equal(I0, I1) and np.array_equal(D0.view(np.uint32), D1.view(np.uint32)), f"four chunks, k={k}"
    g0.close()
    g1.close()


def test_brute_force_self_hit(torch_cuda):
    # reference tests/ut/test_bruteforce.cc:70-76 and test_gpu_search.cc:83-86
    xb = gen_data(10000, 128, 42)
    from knowhere_amd import GpuIndex
    g = GpuIndex(0, ob.L2, 128)
    g.add_vectors(xb)
    D, I = g.search(xb[:1000], 5)
    assert (I[:, 0] == np.arange(1000)).all()
    assert (D[:, 0] == 0).all()
    g.close()


@pytest.mark.parametrize("metric", [ob.L2, ob.IP])
def test_coarse(torch_cuda, port, metric):
    torch = torch_cuda
    xb, xq = gen_data(6000, 64, 42), gen_data(50, 64, 44)
    ix = ob.make_index(port, ob.IVF_FLAT, metric, xb, nlist=300)
    g = _gpu(ix)
    for nprobe in (1, 7, 64, 128, 300):
        Do, Io = port.coarse_search(ix, xq, nprobe)
        D, I = g.coarse_search_device(torch.from_numpy(xq).cuda(), nprobe)
        torch.cuda.synchronize()
        assert_parity(Do, Io, D.cpu().numpy(), I.cpu().numpy(), metric, f"coarse nprobe={nprobe}")
    g.close()


@pytest.mark.parametrize("metric", [ob.L2, ob.IP])
@pytest.mark.parametrize("d", [32, 100])
def test_coarse_bf16_prefilter(torch_cuda, port, metric, d):
    """the coarse prefilter on the bf16 matrix pipe with the selection fused (coarse_gemm.hip, round 5: nlist >= 2048 with
    enough groups of 32 -- or, nprobe 40 and 64 here, 16 -- centroids): keys and coarse distances bit-equal to the reference's, whatever the prefilter's
    arithmetic -- and (second half) centroids tied in masses make its certificate fail and the exact fallback answer"""
    torch = torch_cuda
    from knowhere_amd import GpuIndex
    from knowhere_amd.index import IVF_FLAT
    nlist, nq = 4096 + 40, 333  # (neither a multiple of the 256 x 128 tile)
    rng = np.random.default_rng(5)
    cen = (rng.random((nlist, d), dtype=np.float32) * 100).astype(np.float32)
    xq = (rng.random((nq, d), dtype=np.float32) * 100).astype(np.float32)
    ix = ob.IndexData(ob.IVF_FLAT, metric, d, nlist, 0, 8)
    ix.centroids = cen
    ix.list_codes = [np.empty((0, d * 4), np.uint8)] * nlist  # (only the coarse quantizer is asked)
    ix.list_ids = [np.empty(0, np.int64)] * nlist
    g = GpuIndex(IVF_FLAT, metric, d, nlist=nlist, device=0)
    g.set_coarse_device(torch.from_numpy(cen).cuda())
    xq_t = torch.from_numpy(xq).cuda()
    g.profile_reset()
    for nprobe in (1, 8, 40, 64):
        Do, Io = port.coarse_search(ix, xq, nprobe)
        D, I = g.coarse_search_device(xq_t, nprobe)
        torch.cuda.synchronize()
        assert_parity(Do, Io, D.cpu().numpy(), I.cpu().numpy(), metric, f"bf16 coarse prefilter d={d} nprobe={nprobe}")
    assert g.profile_get()["coarse_fallback_queries"] == 0, "well separated centroids must not need the exact fallback"
    g.close()
    cen2 = cen.copy()
    cen2[:3000] = cen2[rng.integers(3000, 3002, 3000)]  # most centroids are copies of two
    xq2 = np.concatenate([cen2[3000:3002] + 0.01, xq[:20]]).astype(np.float32)
    ix.centroids = cen2
    g = GpuIndex(IVF_FLAT, metric, d, nlist=nlist, device=0)
    g.set_coarse_device(torch.from_numpy(cen2).cuda())
    g.profile_reset()
    for nprobe in (8, 40):
        Do, Io = port.coarse_search(ix, xq2, nprobe)
        D, I = g.coarse_search_device(torch.from_numpy(xq2).cuda(), nprobe)
        torch.cuda.synchronize()
        assert_parity(Do, Io, D.cpu().numpy(), I.cpu().numpy(), metric, f"bf16 coarse ties nprobe={nprobe}", licensed_ties=True)
    assert g.profile_get()["coarse_fallback_queries"] > 0
    g.close()


@pytest.mark.parametrize("metric", [ob.L2, ob.IP])
def test_coarse_certificate_fallback(torch_cuda, port, metric):
    """many identical centroids: the nprobe-th and the (nprobe + margin)-th prefilter values tie, the certificate of the
    MFMA prefilter cannot hold, and the flagged queries go through the exact fallback (flat_full restricted by the flags
    + their "any flag" summary).  Same keys and distances as the oracle; the fallback really ran."""
    torch = torch_cuda
    rng = np.random.default_rng(3)
    xb = gen_data(6000, 64, 42)
    xb[:4500] = xb[rng.integers(4500, 4502, 4500)]  # three quarters of the rows are copies of two vectors: ~110 of
    # the 300 sampled centroids are copies of each -- more than nprobe + margin
    xq = np.concatenate([xb[4500:4502] + 0.01 * gen_data(2, 64, 45), gen_data(30, 64, 44)]).astype(np.float32)
    ix = ob.make_index(port, ob.IVF_FLAT, metric, xb, nlist=300)
    g = _gpu(ix)
    g.profile_reset()
    for nprobe in (8, 32, 64, 128):
        Do, Io = port.coarse_search(ix, xq, nprobe)
        D, I = g.coarse_search_device(torch.from_numpy(xq).cuda(), nprobe)
        torch.cuda.synchronize()
        # (centroids tied at the nprobe-th place: the reference's IndexFlat::search keeps them by its heap for nprobe < 100
        # and by a reservoir above; the coarse stage returns the canonical order -- the one place of the path where
        # that is licensed, include/knhip.h)
        assert_parity(Do, Io, D.cpu().numpy(), I.cpu().numpy(), metric, f"coarse ties nprobe={nprobe}", licensed_ties=True)
    assert g.profile_get()["coarse_fallback_queries"] > 0
    g.close()


def test_ivfflat_get_vectors_direct_map(torch_cuda, port):
    """GetVectorByIds of IVF_FLAT from the index's own rows (knhip_index_get_vectors' direct map, built on first use):
    custom non-monotone ids, ragged and empty lists, an id that is not stored"""
    from knowhere_amd import KnhipError
    nb, d = 5000, 36
    xb = gen_data(nb, d, 42)
    ids = np.random.default_rng(5).permutation(nb).astype(np.int64) * 7 + 3
    ix = ob.make_index(port, ob.IVF_FLAT, ob.L2, xb, nlist=300, ids=ids)
    g = _gpu(ix)
    pick = np.random.default_rng(6).integers(0, nb, 257)
    got = g.get_vectors(ids[pick])
    assert np.array_equal(got.view(np.uint32), xb[pick].view(np.uint32))
    with pytest.raises(KnhipError):
        g.get_vectors(np.array([ids[0], 1], np.int64))  # 1 is not of the form 7 i + 3
    g.close()


KINDS = [(ob.IVF_FLAT, 0), (ob.IVF_PQ, 8), (ob.IVF_PQ, 16), (ob.IVF_PQ, 32), (ob.IVF_PQ, 64), (ob.IVF_SQ8, 0)]


@pytest.mark.parametrize("kind,M", KINDS, ids=lambda v: str(v))
@pytest.mark.parametrize("metric", [ob.L2, ob.IP])
def test_ivf(torch_cuda, port, kind, M, metric):
    nb, d, nlist = 12000, 64, 40
    xb, xq = gen_data(nb, d, 42), gen_data(45, d, 44)
    ix = ob.make_index(port, kind, metric, xb, nlist=nlist, M=max(M, 1))
    g = _gpu(ix)
    for k in (1, 10, 100, 200):
        for nprobe in (1, 8, nlist):
            Do, Io = port.search(ix, xq, k, nprobe)
            D, I = g.search(xq, k, nprobe)
            assert_parity(Do, Io, D, I, metric, f"kind={kind} M={M} k={k} nprobe={nprobe}")
    bs = _bitset(nb, 0.4, 1)
    Do, Io = port.search(ix, xq, 10, 8, bs, nb)
    D, I = g.search(xq, 10, 8, bs, nb)
    assert_parity(Do, Io, D, I, metric, "bitset 40%")
    bs = _bitset(nb, 0.98, 2)
    Do, Io = port.search(ix, xq, 10, nlist, bs, nb)
    D, I = g.search(xq, 10, nlist, bs, nb)
    assert_parity(Do, Io, D, I, metric, "bitset 98%")
    g.close()


@pytest.mark.parametrize("kind", [ob.IVF_FLAT, ob.IVF_SQ8])
@pytest.mark.parametrize("metric", [ob.L2, ob.IP])
def test_large_k_row_scans(torch_cuda, port, kind, metric):
    """k up to 1024 on the row scans (refine asks IVF_SQ8 for k x refine_k candidates, ivf.cc:1073-1103): the
    per-item query group shrinks as k grows (8 / 4 / 2 / 1)"""
    nb, d, nlist = 9000, 48, 12
    xb, xq = gen_data(nb, d, 42), gen_data(21, d, 44)
    ix = ob.make_index(port, kind, metric, xb, nlist=nlist)
    g = _gpu(ix)
    for k, nprobe in ((129, 3), (256, 12), (400, 5), (512, 12), (1000, 7), (1024, 12)):
        Do, Io = port.search(ix, xq, k, nprobe)
        D, I = g.search(xq, k, nprobe)
        assert_parity(Do, Io, D, I, metric, f"kind={kind} k={k} nprobe={nprobe}")
    bs = _bitset(nb, 0.7, 3)
    Do, Io = port.search(ix, xq, 600, 12, bs, nb)
    D, I = g.search(xq, 600, 12, bs, nb)
    assert_parity(Do, Io, D, I, metric, "large k + bitset (fewer survivors than k in places)")
    g.close()


def test_ivfpq_residual_tables(torch_cuda, port):
    # precomputed table over the limit -> reference falls back to per-list residual tables
    # (thirdparty/faiss/faiss/IndexIVFPQ.cpp:441-456)
    nb, d = 8000, 64
    xb, xq = gen_data(nb, d, 42), gen_data(20, d, 44)
    ix = ob.make_index(port, ob.IVF_PQ, ob.L2, xb, nlist=32, M=16)
    ix.use_precomputed_table = 0
    ix.precomputed_table = None
    g = _gpu(ix, precomputed_table_max_bytes=1024)
    assert g.uses_precomputed_table == 0
    for k, nprobe in ((10, 8), (1, 32)):
        Do, Io = port.search(ix, xq, k, nprobe)
        D, I = g.search(xq, k, nprobe)
        assert_parity(Do, Io, D, I, ob.L2, "residual tables")
    g.close()


def test_ivfpq_long_lists_headline_shape(torch_cuda, port):
    # d=128, m=32: the headline configuration's shape at a size the oracle finishes in seconds;
    # lists of ~3000 codes exercise many pipeline blocks per pipe
    nb, d = 200000, 128
    xb, xq = gen_data(nb, d, 42), gen_data(200, d, 44)
    ix = ob.make_index(port, ob.IVF_PQ, ob.L2, xb, nlist=64, M=32)
    g = _gpu(ix)
    Do, Io = port.search(ix, xq, 10, 16)
    D, I = g.search(xq, 10, 16)
    assert_parity(Do, Io, D, I, ob.L2, "headline shape")
    g.close()


def test_edge_cases(torch_cuda, port):
    d = 32
    xb = gen_data(300, d, 42)
    xb[100:140] = xb[5]  # exact duplicates: distance ties
    xq = gen_data(9, d, 44)
    xq[0] = xb[5]
    for kind in (ob.IVF_FLAT, ob.IVF_PQ, ob.IVF_SQ8):
        # nlist close to n -> many empty and tiny lists; custom non-monotone ids
        ids = np.random.default_rng(5).permutation(300).astype(np.int64) * 3 + 1
        ix = sort_lists_by_id(ob.make_index(port, kind, ob.L2, xb, nlist=64, M=8, ids=ids))  # (shuffled ids, stored in id order)
        g = _gpu(ix)
        for k, nprobe in ((10, 64), (128 if kind != ob.IVF_SQ8 else 100, 64), (3, 1)):
            Do, Io = port.search(ix, xq, k, nprobe)
            D, I = g.search(xq, k, nprobe)
            assert_parity(Do, Io, D, I, ob.L2, f"edge kind={kind} k={k} nprobe={nprobe}")
        # k larger than everything reachable: sentinel tail id -1 / FLT_MAX
        kbig = 400 if kind != ob.IVF_SQ8 else 128
        D, I = g.search(xq, kbig, 2)
        Do, Io = port.search(ix, xq, kbig, 2)
        assert_parity(Do, Io, D, I, ob.L2, "k > candidates")
        assert (I[:, -1] == -1).all() and (D[:, -1] == np.finfo(np.float32).max).all()
        # single query, and nq == 0
        D, I = g.search(xq[:1], 5, 4)
        Do, Io = port.search(ix, xq[:1], 5, 4)
        assert_parity(Do, Io, D, I, ob.L2, "nq=1")
        D, I = g.search(xq[:0], 5, 4)
        assert D.shape == (0, 5)
        g.close()


def test_error_convention(torch_cuda):
    from knowhere_amd import GpuIndex, KnhipError
    g = GpuIndex(2, ob.L2, 64, nlist=8, pq_m=8)
    with pytest.raises(KnhipError) as e:
        g.search(np.zeros((1, 64), np.float32), 5, 2)
    assert e.value.code == -2  # index_not_trained
    with pytest.raises(KnhipError):
        GpuIndex(2, ob.L2, 64, nlist=8, pq_m=7)  # m must divide dim
    with pytest.raises(KnhipError):
        GpuIndex(2, 5, 64, nlist=8, pq_m=8)  # bad metric
    g.close()


def test_device_boundary_and_determinism(torch_cuda, port):
    torch = torch_cuda
    xb, xq = gen_data(30000, 64, 42), gen_data(128, 64, 44)
    ix = ob.make_index(port, ob.IVF_PQ, ob.L2, xb, nlist=64, M=32)
    g = _gpu(ix)
    q = torch.from_numpy(xq).cuda()
    D0, I0 = g.search_device(q, 10, 16)
    torch.cuda.synchronize()
    for _ in range(3):  # work-item grouping uses atomics; results must not depend on it
        D1, I1 = g.search_device(q, 10, 16)
        torch.cuda.synchronize()
        assert torch.equal(D0, D1) and torch.equal(I0, I1)
    Do, Io = port.search(ix, xq, 10, 16)
    assert_parity(Do, Io, D0.cpu().numpy(), I0.cpu().numpy(), ob.L2, "device boundary")
    # shard merge on device == oracle merge
    from knowhere_amd.index import merge_topk_device
    halves = []
    for part in range(2):
        sub = ob.IndexData(ix.kind, ix.metric, ix.d, ix.nlist, ix.M, ix.nbits)
        sub.centroids, sub.pq_centroids = ix.centroids, ix.pq_centroids
        sub.precomputed_table, sub.use_precomputed_table = ix.precomputed_table, 1
        sub.list_codes = [c if l % 2 == part else c[:0] for l, c in enumerate(ix.list_codes)]
        sub.list_ids = [i if l % 2 == part else i[:0] for l, i in enumerate(ix.list_ids)]
        gs = _gpu(sub)
        halves.append(gs.search_device(q, 10, 16))
        torch.cuda.synchronize()
        gs.close()
    Dp = torch.stack([h[0] for h in halves])
    Ip = torch.stack([h[1] for h in halves])
    Dm, Im = merge_topk_device(ob.L2, Dp, Ip)
    torch.cuda.synchronize()
    assert_parity(Do, Io, Dm.cpu().numpy(), Im.cpu().numpy(), ob.L2, "list-sharded search + merge == monolithic")
    g.close()


def test_primitives(torch_cuda, port):
    torch = torch_cuda
    import ctypes as C
    from knowhere_amd import _lib
    L = _lib.load()
    r = np.random.default_rng(9)
    for d in (1, 3, 4, 17, 64, 100, 128, 256, 768):
        ny = 333
        x = (r.random(d, dtype=np.float32) * 100).astype(np.float32)
        y = (r.random((ny, d), dtype=np.float32) * 100).astype(np.float32)
        xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
        out = torch.empty(ny, dtype=torch.float32, device="cuda")
        p = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(L.knhip_fvec_L2sqr_ny(p(out), p(xt), p(yt), d, ny, None))
        torch.cuda.synchronize()
        assert out.cpu().numpy().tobytes() == port.fvec_L2sqr_ny(x, y).tobytes(), f"L2sqr_ny d={d}"
        _lib.check(L.knhip_fvec_inner_products_ny(p(out), p(xt), p(yt), d, ny, None))
        torch.cuda.synchronize()
        assert out.cpu().numpy().tobytes() == port.fvec_inner_products_ny(x, y).tobytes(), f"ip_ny d={d}"
        _lib.check(L.knhip_fvec_norms_L2sqr(p(out), p(yt), d, ny, None))
        torch.cuda.synchronize()
        nr = np.array([port.fvec_norm_L2sqr(np.ascontiguousarray(v)) for v in y], np.float32)
        assert out.cpu().numpy().tobytes() == nr.tobytes(), f"norms d={d}"
        xi = r.integers(-128, 128, d, dtype=np.int8)
        yi = r.integers(-128, 128, (ny, d), dtype=np.int8)
        xit, yit = torch.from_numpy(xi).cuda(), torch.from_numpy(yi).cuda()
        _lib.check(L.knhip_int8_vec_L2sqr_ny(p(out), p(xit), p(yit), d, ny, None))
        torch.cuda.synchronize()
        assert out.cpu().numpy().tobytes() == port.int8_ny(xi, yi, ob.L2).tobytes()
        _lib.check(L.knhip_int8_vec_inner_products_ny(p(out), p(xit), p(yit), d, ny, None))
        torch.cuda.synchronize()
        assert out.cpu().numpy().tobytes() == port.int8_ny(xi, yi, ob.IP).tobytes()
    a = (r.random(5000, dtype=np.float32) * 10).astype(np.float32)
    b = (r.random(5000, dtype=np.float32) * 10).astype(np.float32)
    at, bt = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    ct = torch.empty_like(at)
    _lib.check(L.knhip_fvec_madd(5000, C.c_void_p(at.data_ptr()), -2.0, C.c_void_p(bt.data_ptr()),
                                 C.c_void_p(ct.data_ptr()), None))
    torch.cuda.synchronize()
    assert ct.cpu().numpy().tobytes() == port.fvec_madd(a, -2.0, b).tobytes()


@pytest.mark.parametrize("metric", [ob.L2, ob.IP])
def test_refine(torch_cuda, port, metric):
    # Knowhere `refine`: IndexRefine over IVF_PQ (reference src/index/ivf/ivf.cc:1073-1103)
    torch = torch_cuda
    from knowhere_amd.index import refine_device
    nb, d = 15000, 64
    xb, xq = gen_data(nb, d, 42), gen_data(60, d, 44)
    ix = ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=32, M=16)
    g = _gpu(ix)
    q = torch.from_numpy(xq).cuda()
    base = torch.from_numpy(xb).cuda()
    for k, kbase, nprobe in ((10, 40, 8), (1, 64, 4), (100, 200, 32), (10, 500, 1)):
        Dc, Ic = g.search_device(q, kbase, nprobe)
        torch.cuda.synchronize()
        Do_c, Io_c = port.search(ix, xq, kbase, nprobe)
        assert_parity(Do_c, Io_c, Dc.cpu().numpy(), Ic.cpu().numpy(), metric, "refine stage 1")
        D, I = refine_device(metric, base, q, Ic, k)
        torch.cuda.synchronize()
        Do, Io = port.refine(metric, xb, xq, Io_c, k)
        assert_parity(Do, Io, D.cpu().numpy(), I.cpu().numpy(), metric, f"refine k={k} kbase={kbase}")
    g.close()


def test_refine_empty_base_and_long_rows(torch_cuda, port):
    """(a) a shard that holds NO rows (nbase == 0, base pointer possibly null) must answer "not here" /
    empty results instead of reading row 0 in the cooperative gather; (b) fp32 rows too long for the cooperative staging
    area beside four queries (d >= ~5.4k) fall back to the lane-per-row gather instead of failing."""
    torch = torch_cuda
    import ctypes as C
    from knowhere_amd import _lib
    from knowhere_amd.index import refine_device, refine_distances_device
    from knowhere_amd._lib import check
    L = _lib.load()
    d, nq, kbase = 64, 9, 40
    q = torch.from_numpy(gen_data(nq, d, 44)).cuda()
    cand = torch.randint(0, 1000, (nq, kbase), dtype=torch.int64, device="cuda")
    empty = torch.empty((0, d), dtype=torch.float32, device="cuda")
    D = refine_distances_device(ob.L2, empty, q, cand)
    torch.cuda.synchronize()
    assert (D.view(torch.int32) == -1).all()  # REFINE_NOT_HERE everywhere
    # ... with a null base pointer, as knhip_refine_distances_device allows for nbase == 0
    D2 = torch.zeros((nq, kbase), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    check(L.knhip_refine_distances_device(ob.L2, d, None, 0, 0, C.c_void_p(q.data_ptr()), nq, C.c_void_p(cand.data_ptr()),
                                          kbase, C.c_void_p(D2.data_ptr()), C.c_void_p(s)))
    torch.cuda.synchronize()
    assert (D2.view(torch.int32) == -1).all()
    Ds, Is = refine_device(ob.L2, empty, q, cand, 10)
    torch.cuda.synchronize()
    assert (Is == -1).all()
    # (b) d = 6000, k_base = 100: 4 d + 4 k_base floats = 97.6 KB, + 69.6 KB of staging > 160 KB
    nb, d2, k = 300, 6000, 5
    xb, xq = gen_data(nb, d2, 42), gen_data(6, d2, 44)
    rng = np.random.default_rng(3)
    ids = np.stack([rng.permutation(nb)[:100] for _ in range(6)]).astype(np.int64)
    Dg, Ig = refine_device(ob.L2, torch.from_numpy(xb).cuda(), torch.from_numpy(xq).cuda(), torch.from_numpy(ids).cuda(), k)
    torch.cuda.synchronize()
    Do, Io = port.refine(ob.L2, xb, xq, ids, k)
    assert_parity(Do, Io, Dg.cpu().numpy(), Ig.cpu().numpy(), ob.L2, "refine d=6000")


def test_gpu_builder_index_parity_and_recall(torch_cuda, port):
    # index trained/encoded by the GPU builder (bench.py's path): same bytes to oracle and GPU
    torch = torch_cuda
    from knowhere_amd import build as kb
    from knowhere_amd import index as kidx
    spec = kb.DataSpec(300000, 64, kind="mixture", ncenter=4096, sigma=0.35)
    built = kb.build_ivf(spec, kidx.IVF_PQ, kidx.L2, nlist=256, M=16, keep_vectors=True)
    g = built.to_gpu_index()
    xq = kb.queries(spec, 100, "cuda:0")
    D, I = g.search_device(xq, 50, 16)
    Dr, Ir = kidx.refine_device(kidx.L2, built.vectors, xq, I, 10)
    torch.cuda.synchronize()
    ix = built.export(ob.IndexData)
    ix.use_precomputed_table = 1
    finish_ivfpq(port, ix)
    Do, Io = port.search(ix, xq.cpu().numpy(), 50, 16)
    assert_parity(Do, Io, D.cpu().numpy(), I.cpu().numpy(), ob.L2, "builder index")
    _, gt = kb.ground_truth(spec, xq, 10)
    # ground truth itself against the oracle's exact search
    Dg, Ig = port.flat_search(ob.L2, built.vectors.cpu().numpy(), xq.cpu().numpy(), 10)
    assert (np.sort(gt.cpu().numpy(), 1) == np.sort(Ig, 1)).mean() > 0.999
    rec = (Ir.unsqueeze(2) == gt.unsqueeze(1)).any(2).float().mean().item()
    assert rec > 0.6, rec  # sanity floor (reference floors: tests/ut/test_gpu_search.cc:179-187)
    g.close()


def test_config_shapes_ivfsq8_ip_d768_and_ivfflat_d128(torch_cuda, port):
    # BASELINE.json configs[4] (IVF-SQ8 IP, d=768) and configs[1] (IVF-Flat L2, d=128) at sizes the
    # oracle finishes in seconds; "int8" inputs in Knowhere are converted to fp32 and SQ8-trained
    # (reference include/knowhere/index/index_factory.h:144-145, index_node_data_mock_wrapper.cc:55-61)
    r = np.random.default_rng(5)
    xb = r.integers(-128, 128, (6000, 768)).astype(np.float32)
    xq = r.integers(-128, 128, (25, 768)).astype(np.float32)
    ix = ob.make_index(port, ob.IVF_SQ8, ob.IP, xb, nlist=48)
    g = _gpu(ix)
    for k, nprobe in ((10, 16), (100, 48)):
        Do, Io = port.search(ix, xq, k, nprobe)
        D, I = g.search(xq, k, nprobe)
        assert_parity(Do, Io, D, I, ob.IP, f"SQ8 IP d=768 k={k}")
    g.close()
    xb, xq = gen_data(60000, 128, 42), gen_data(64, 128, 44)
    ix = ob.make_index(port, ob.IVF_FLAT, ob.L2, xb, nlist=128)
    g = _gpu(ix)
    Do, Io = port.search(ix, xq, 10, 64)
    D, I = g.search(xq, 10, 64)
    assert_parity(Do, Io, D, I, ob.L2, "IVF-Flat L2 d=128 nprobe=64")
    g.close()


def test_scale_properties_ivfpq_1m(torch_cuda):
    # size-independent properties at a size the oracle cannot run (1M x 128, nlist 1024):
    # (1) list-sharded search + merge == monolithic, bit for bit; (2) results sorted and ids unique;
    # (3) a bitset that removes the current top-1 makes the old top-2 the new top-1
    torch = torch_cuda
    from knowhere_amd import build as kb
    from knowhere_amd import index as kidx
    from knowhere_amd.index import merge_topk_device
    spec = kb.DataSpec(1_000_000, 128, ncenter=8192)
    built = kb.build_ivf(spec, kidx.IVF_PQ, kidx.L2, nlist=1024, M=32)
    xq = kb.queries(spec, 2000, "cuda:0")
    g = built.to_gpu_index()
    D, I = g.search_device(xq, 10, 32)
    torch.cuda.synchronize()
    assert (D[:, 1:] >= D[:, :-1]).all()
    assert all(len(set(r.tolist())) == 10 for r in I[:200].cpu())
    parts = []
    for r in range(3):
        own = (np.arange(1024) % 3) == r
        gs = built.to_gpu_index(owned_lists=own)
        parts.append(gs.search_device(xq, 10, 32))
        torch.cuda.synchronize()
        gs.close()
    Dm, Im = merge_topk_device(kidx.L2, torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    torch.cuda.synchronize()
    assert torch.equal(Dm, D) and torch.equal(Im, I)
    nbits = 1_000_000
    top1 = I[:, 0].clone()
    bsh = np.zeros((nbits + 7) // 8, np.uint8)
    for t in top1.cpu().numpy():
        bsh[t >> 3] |= 1 << (t & 7)
    D2, I2 = g.search_device(xq, 10, 32, bitset_t=torch.from_numpy(bsh).cuda(), nbits=nbits)
    torch.cuda.synchronize()
    removed = set(top1.cpu().numpy().tolist())
    I2c = I2.cpu().numpy()
    assert not any(int(i) in removed for i in I2c.ravel() if i >= 0)
    Ic = I.cpu().numpy()
    for q in range(0, 2000, 97):  # first surviving old result is the new top-1
        surv = [i for i in Ic[q] if int(i) not in removed]
        if surv:
            assert I2c[q, 0] == surv[0]
    g.close()


@pytest.mark.parametrize("kind,M", [(ob.IVF_FLAT, 0), (ob.IVF_PQ, 32), (ob.IVF_PQ, 16), (ob.IVF_SQ8, 0)],
                         ids=["ivfflat", "ivfpq32", "ivfpq16", "ivfsq8"])
def test_search_preassigned_equals_search(torch_cuda, port, kind, M):
    """IndexIVF::search == quantizer search + search_preassigned (IndexIVF.cpp:336-350): the split entry point
    used by the list-sharded deployment returns exactly what knhip_search_device returns"""
    torch = torch_cuda
    nb, nq, d, nlist, nprobe = 20000, 77, 64, 64, 12
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    ix = finish_ivfpq(port, ob.make_index(port, kind, ob.L2, xb, nlist=nlist, M=max(M, 1), nbits=8))
    g = _gpu(ix)
    xq_t = torch.from_numpy(xq).cuda()
    for k in (10, 100):
        D, I = g.search_device(xq_t, k, nprobe)
        cd, ck = g.coarse_search_device(xq_t, nprobe)
        D2, I2 = g.search_preassigned_device(xq_t, k, ck, cd)
        torch.cuda.synchronize()
        assert torch.equal(I, I2) and torch.equal(D.view(torch.int32), D2.view(torch.int32))
    # a dropped probe (key = -1) is simply not scanned
    cd, ck = g.coarse_search_device(xq_t, nprobe)
    ck2 = ck.clone()
    ck2[:, 1:] = -1
    D1, I1 = g.search_device(xq_t, 10, 1)
    D3, I3 = g.search_preassigned_device(xq_t, 10, ck2, cd)
    torch.cuda.synchronize()
    assert torch.equal(I1, I3)


@pytest.mark.parametrize("metric", [ob.L2, ob.IP])
def test_q4_scan_matches_v2_and_oracle(torch_cuda, port, monkeypatch, metric):
    """pq_scan_q4 (persistent workgroups, 4 queries per item, LUT computed in-kernel; d = 128, m = 32) returns
    exactly what the oracle and the 2-query kernel return: k = 10 (bulk launch only), k = 100 (rank-0 dump +
    select by pq_scan_v2, bulk by pq_scan_q4), bitset, ragged item tails (nq not a multiple of 4)"""
    nb, nq, d, nlist, nprobe = 60000, 203, 128, 128, 32
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    ix = finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=nlist, M=32, nbits=8))
    bs = _bitset(nb, 0.4, 1)
    monkeypatch.setenv("KNHIP_Q4", "0")  # read when the lists are attached
    g2 = _gpu(ix)
    monkeypatch.setenv("KNHIP_Q4", "1")
    g4 = _gpu(ix)
    g4.search(xq, 100, nprobe)  # (layouts built on first use are built now)
    bytes4 = g4.device_bytes
    # k = 128: the tie rule searches for 129 results -- R = 3 of both kernels (a caller's k = 128 must stay
    # on these kernels, not fall onto the systolic one and its second copy of the codes); k = 150: plain R = 3
    for k, np_, use_bs in ((10, nprobe, False), (100, nprobe, False), (10, 1, False), (64, 128, False), (10, nprobe, True),
                           (100, 8, True), (128, nprobe, False), (128, 8, True), (150, nprobe, False)):
        b, nbits = (bs, nb) if use_bs else (None, 0)
        Do, Io = port.search(ix, xq, k, np_, b, nbits)
        D2, I2 = g2.search(xq, k, np_, b, nbits)
        D4, I4 = g4.search(xq, k, np_, b, nbits)
        assert_parity(Do, Io, D4, I4, metric, f"q4 vs oracle k={k} nprobe={np_} bitset={use_bs}")
        assert np.array_equal(I2, I4) and np.array_equal(D2.view(np.uint32), D4.view(np.uint32))
    # few queries: most items hold fewer than 4 pairs
    for nq_small in (1, 3, 5):
        Do, Io = port.search(ix, xq[:nq_small], 10, nprobe)
        D4, I4 = g4.search(xq[:nq_small], 10, nprobe)
        assert_parity(Do, Io, D4, I4, metric, f"q4 nq={nq_small}")
    assert g4.device_bytes == bytes4, "a k = 128 search built another copy of the codes"
    g2.close()
    g4.close()


def test_q4_scan_residual_tables_and_ragged_lists(torch_cuda, port, monkeypatch):
    """pq_scan_q4 with per-list residual tables (precomputed table over the limit) and with empty / tiny lists"""
    monkeypatch.setenv("KNHIP_Q4", "1")
    nb, d = 30000, 128
    xb, xq = gen_data(nb, d, 42), gen_data(50, d, 44)
    ix = ob.make_index(port, ob.IVF_PQ, ob.L2, xb, nlist=64, M=32)
    ix.use_precomputed_table = 0
    ix.precomputed_table = None
    g = _gpu(ix, precomputed_table_max_bytes=1024)
    assert g.uses_precomputed_table == 0
    for k, nprobe in ((10, 8), (100, 64)):
        Do, Io = port.search(ix, xq, k, nprobe)
        D, I = g.search(xq, k, nprobe)
        assert_parity(Do, Io, D, I, ob.L2, "q4 residual tables")
    g.close()
    xb = gen_data(700, d, 42)
    xb[100:140] = xb[5]  # exact duplicates: distance ties
    ids = np.random.default_rng(5).permutation(700).astype(np.int64) * 3 + 1
    ix = sort_lists_by_id(finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, ob.L2, xb, nlist=17, M=32, ids=ids)))
    # lists emptied by hand: work items must skip them
    for l in (0, 3):
        ix.list_codes[l] = ix.list_codes[l][:0]
        ix.list_ids[l] = ix.list_ids[l][:0]
    g = _gpu(ix)
    for k, nprobe in ((10, 17), (128, 17), (3, 1)):
        Do, Io = port.search(ix, xq, k, nprobe)
        D, I = g.search(xq, k, nprobe)
        assert_parity(Do, Io, D, I, ob.L2, f"q4 ragged k={k} nprobe={nprobe}")
    g.close()
