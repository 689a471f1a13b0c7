"""GPU build tests (-m gpu): knhip_kmeans_device / knhip_index_train* / knhip_index_add* against the oracle's
restatements of faiss::Clustering and IndexIVF::train / add_core (oracle.c: orc_kmeans is pinned bit-for-bit against
the reference's own Clustering in tests/test_oracle.py).  Bar: centroids, codebooks, SQ ranges, assignments and codes
bit-equal; a search on the GPU-built index == the oracle's search on the same index bytes."""
import numpy as np
import pytest

from conftest import assert_parity, gen_data
from helpers import finish_ivfpq
from oracle import binding as ob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch


@pytest.mark.parametrize("n,d,k,metric,maxpts", [(3000, 16, 32, ob.L2, 256),   # LDS codebook path
                                                  (9000, 4, 256, ob.L2, 256),   # the PQ sub-quantizer shape
                                                  (6000, 128, 40, ob.L2, 256),  # coarse-quantizer path (d > 64)
                                                  (5000, 32, 24, ob.IP, 256),   # inner-product assigner
                                                  (4000, 96, 10, ob.L2, 64)])   # sub-sampled training set
def test_kmeans_equals_reference_restatement(torch_cuda, port, n, d, k, metric, maxpts):
    torch = torch_cuda
    from knowhere_amd.index import kmeans_device
    x = gen_data(n, d, 42, -3.0, 7.0)
    cen = kmeans_device(metric, torch.from_numpy(x).cuda(), k, niter=12, max_points=maxpts)
    torch.cuda.synchronize()
    co = port.kmeans(metric, x, k, niter=12, max_points=maxpts)
    assert cen.cpu().numpy().tobytes() == co.tobytes(), f"k-means centroids differ (n={n} d={d} k={k})"


def test_spherical_kmeans(torch_cuda, port):
    """Clustering::post_process_centroids with spherical = true (what IndexIVF sets for the inner product)"""
    torch = torch_cuda
    from knowhere_amd.index import kmeans_device
    for n, d, k in ((5000, 32, 24), (6000, 128, 40)):
        x = gen_data(n, d, 42, -3.0, 7.0)
        cen = kmeans_device(ob.IP, torch.from_numpy(x).cuda(), k, niter=7, spherical=True)
        torch.cuda.synchronize()
        co = port.kmeans(ob.IP, x, k, niter=7, spherical=True)
        assert cen.cpu().numpy().tobytes() == co.tobytes()
        assert np.allclose(np.linalg.norm(co, axis=1), 1.0, atol=1e-5)


def test_kmeans_codebook_larger_than_lds(torch_cuda, port):
    """k x d floats above the 160 KB of one workgroup's LDS (d = 64, k = 1024) take the coarse-quantizer path"""
    torch = torch_cuda
    from knowhere_amd.index import kmeans_device
    x = gen_data(5000, 64, 42, -3.0, 7.0)
    cen = kmeans_device(ob.L2, torch.from_numpy(x).cuda(), 1024, niter=3)
    torch.cuda.synchronize()
    assert cen.cpu().numpy().tobytes() == port.kmeans(ob.L2, x, 1024, niter=3).tobytes()


def _ivf_goldens():
    import os
    from helpers import golden_files
    from test_oracle import BLAS_NEAR_TIE  # (two fixtures whose reference training flips a near-tie inside sgemm)
    return [p for p in golden_files() if not os.path.basename(p).startswith("flat_")
            and os.path.basename(p)[:-4] not in BLAS_NEAR_TIE]


@pytest.mark.parametrize("path", _ivf_goldens(), ids=lambda p: p.split("/")[-1][:-4])
def test_train_add_rebuilds_the_reference_built_goldens(torch_cuda, path):
    """knhip_index_train + knhip_index_add at DEFAULT parameters on gen(nb, d, 42) == the bytes of the index the
    reference's own IndexIVF::train + add built from the same rows (tests/golden/make_golden.py): centroids (spherical
    for the inner product, 10 level-1 iterations), PQ codebooks, SQ8 ranges, list sizes, ids and codes"""
    from helpers import load_golden
    from knowhere_amd import GpuIndex
    ix, _, _ = load_golden(path)
    nb = int(np.load(path)["nb"])
    xb = gen_data(nb, ix.d, 42)
    g = GpuIndex(ix.kind, ix.metric, ix.d, nlist=ix.nlist, pq_m=ix.M)
    g.train(xb)
    assert g.get_coarse().tobytes() == ix.centroids.tobytes(), "coarse centroids"
    if ix.kind == ob.IVF_PQ:
        assert g.get_pq().tobytes() == ix.pq_centroids.tobytes(), "PQ codebooks"
    if ix.kind == ob.IVF_SQ8:
        assert g.get_sq().tobytes() == ix.sq_trained.tobytes(), "SQ8 ranges"
    g.add(xb)
    sizes, codes, ids = g.get_lists()
    assert (sizes == np.array([len(i) for i in ix.list_ids])).all(), "list sizes"
    assert (ids == np.concatenate(ix.list_ids)).all(), "ids"
    assert codes.tobytes() == np.concatenate([c.reshape(len(c), -1) for c in ix.list_codes]).tobytes(), "codes"
    g.close()


def test_kmeans_empty_cluster_split(torch_cuda, port):
    """duplicated points leave clusters empty: split_clusters (donor pick by the reference's RNG) must match"""
    torch = torch_cuda
    from knowhere_amd.index import kmeans_device
    base = gen_data(12, 8, 1)
    x = np.repeat(base, 40, axis=0)  # 480 points, 12 distinct: k = 32 forces empty clusters
    cen = kmeans_device(ob.L2, torch.from_numpy(x).cuda(), 32, niter=6)
    torch.cuda.synchronize()
    assert cen.cpu().numpy().tobytes() == port.kmeans(ob.L2, x, 32, niter=6).tobytes()


@pytest.mark.parametrize("kind,M,metric", [(ob.IVF_PQ, 8, ob.L2), (ob.IVF_PQ, 32, ob.L2), (ob.IVF_PQ, 16, ob.IP),
                                           (ob.IVF_PQ, 4, ob.L2), (ob.IVF_PQ, 2, ob.IP),  # (widths of pq_scan_any.hip)
                                           (ob.IVF_SQ8, 0, ob.L2), (ob.IVF_SQ8, 0, ob.IP), (ob.IVF_FLAT, 0, ob.L2),
                                           (ob.IVF_FLAT, 0, ob.IP)])
def test_train_add_search_pipeline(torch_cuda, port, kind, M, metric):
    """Train + two Adds on the device == the restated reference pipeline; Search on the result == oracle search"""
    from knowhere_amd import GpuIndex
    nb, d, nlist = 9000, 64 if M != 32 else 128, 24
    xb, xq = gen_data(nb, d, 42), gen_data(33, d, 44)
    g = GpuIndex(kind, metric, d, nlist=nlist, pq_m=M)
    g.train(xb[:6000], niter=8)
    cen, pq, sq = port.train_ivf(kind, metric, xb[:6000], nlist, M=M, niter=8)
    assert g.get_coarse().tobytes() == cen.tobytes(), "coarse centroids"
    if kind == ob.IVF_PQ:
        assert g.get_pq().tobytes() == pq.tobytes(), "PQ codebooks"
    if kind == ob.IVF_SQ8:
        assert g.get_sq().tobytes() == sq.tobytes(), "SQ8 ranges"
    g.add(xb[:5000])        # host rows
    import torch
    g.add(torch.from_numpy(xb[5000:]).cuda())  # device rows, second Add appends
    assert g.count == nb
    sizes, codes, ids = g.get_lists()
    # the same index built by the restated add path
    ix = ob.IndexData(kind, metric, d, nlist, M, 8)
    ix.centroids, ix.pq_centroids, ix.sq_trained = cen, pq, sq
    assign = port.assign(metric, cen, xb)
    resid = xb - cen[assign]
    if kind == ob.IVF_PQ:
        oc = port.pq_encode(d, M, 8, pq, np.ascontiguousarray(resid))
    elif kind == ob.IVF_SQ8:
        oc = port.sq8_encode(sq, np.ascontiguousarray(resid))
    else:
        oc = xb.view(np.uint8).reshape(nb, d * 4)
    ix.list_codes = [np.ascontiguousarray(oc[assign == l]) for l in range(nlist)]
    ix.list_ids = [np.nonzero(assign == l)[0].astype(np.int64) for l in range(nlist)]
    assert (sizes == np.array([len(i) for i in ix.list_ids])).all(), "list sizes (assignment)"
    assert (ids == np.concatenate(ix.list_ids)).all(), "ids"
    assert codes.tobytes() == np.concatenate(ix.list_codes).tobytes(), "codes"
    if kind == ob.IVF_PQ and metric == ob.L2:
        ix.use_precomputed_table = 1
        finish_ivfpq(port, ix)
    for k, nprobe in ((10, 8), (100, nlist)):
        Do, Io = port.search(ix, xq, k, nprobe)
        D, I = g.search(xq, k, nprobe)
        assert_parity(Do, Io, D, I, metric, f"search on the GPU-built index kind={kind} k={k}")
    g.close()


def test_encode_device_matches_add(torch_cuda, port):
    torch = torch_cuda
    from knowhere_amd import GpuIndex
    xb = gen_data(7000, 128, 42)
    ix = ob.make_index(port, ob.IVF_PQ, ob.L2, xb, nlist=32, M=32)
    g = GpuIndex.from_data(ix, device=0)
    a, c = g.encode_device(torch.from_numpy(xb[:2000]).cuda())
    torch.cuda.synchronize()
    ao = port.assign(ob.L2, ix.centroids, xb[:2000])
    co = port.pq_encode(128, 32, 8, ix.pq_centroids, np.ascontiguousarray(xb[:2000] - ix.centroids[ao]))
    assert (a.cpu().numpy() == ao).all() and c.cpu().numpy().tobytes() == co.tobytes()
    g.close()


def test_brute_force_repeated_add(torch_cuda, port):
    from knowhere_amd import GpuIndex
    xb, xq = gen_data(5000, 32, 42), gen_data(20, 32, 44)
    g = GpuIndex(0, ob.L2, 32)
    g.add(xb[:1234])
    g.add(xb[1234:])
    ix = ob.make_index(port, ob.FLAT, ob.L2, xb)
    Do, Io = port.search(ix, xq, 10)
    D, I = g.search(xq, 10)
    assert_parity(Do, Io, D, I, ob.L2, "brute force after two Adds")
    g.close()


@pytest.mark.parametrize("kind,d", [(ob.IVF_SQ8, 20), (ob.IVF_SQ8, 64), (ob.IVF_FLAT, 6), (ob.IVF_FLAT, 32)])
def test_large_lists_drop_the_aos_copy_and_rebuild_it(torch_cuda, port, monkeypatch, kind, d):
    """flat / SQ8 indexes above KNHIP_AOS_KEEP_MB keep only the interleaved layout; get_lists (Serialize) and a further
    Add rebuild the canonical bytes from it: same bytes, same search results (d not a multiple of 16 / 4 included)"""
    from knowhere_amd import GpuIndex
    monkeypatch.setenv("KNHIP_AOS_KEEP_MB", "0")
    nb, nlist = 7000, 16
    xb, xq = gen_data(nb, d, 42), gen_data(25, d, 44)
    g = GpuIndex(kind, ob.L2, d, nlist=nlist)
    g.train(xb[:4000], niter=5)
    g.add(xb[:3000])
    sizes0, codes0, ids0 = g.get_lists()       # rebuilt from the interleaved blocks
    g.add(xb[3000:])                           # merge needs the old canonical bytes again
    sizes, codes, ids = g.get_lists()
    monkeypatch.setenv("KNHIP_AOS_KEEP_MB", "100000")
    h = GpuIndex(kind, ob.L2, d, nlist=nlist)  # same build with the copy resident
    h.train(xb[:4000], niter=5)
    h.add(xb[:3000])
    s0, c0, i0 = h.get_lists()
    h.add(xb[3000:])
    s1, c1, i1 = h.get_lists()
    assert (sizes0 == s0).all() and codes0.tobytes() == c0.tobytes() and (ids0 == i0).all()
    assert (sizes == s1).all() and codes.tobytes() == c1.tobytes() and (ids == i1).all()
    D, I = g.search(xq, 10, 8)
    D2, I2 = h.search(xq, 10, 8)
    assert D.tobytes() == D2.tobytes() and (I == I2).all()
    g.close()
    h.close()


def test_train_errors(torch_cuda):
    from knowhere_amd import GpuIndex, KnhipError
    g = GpuIndex(ob.IVF_PQ, ob.L2, 32, nlist=64, pq_m=8)
    with pytest.raises(KnhipError):
        g.add(gen_data(10, 32, 1))  # not trained
    with pytest.raises(KnhipError):
        g.train(gen_data(20, 32, 1))  # fewer training vectors than centroids
    g.close()
