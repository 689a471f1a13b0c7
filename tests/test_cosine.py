"""a12 COSINE, CPU side: the oracle's restatement (oracle.c orc_normalize_vecs / orc_inverse_l2_norms /
orc_flat_cosine_search / the list_norms path of the IVF_FLAT scanner) against known answers produced by the classes the
reference's nodes instantiate -- IndexFlatCosine, IndexIVFFlatCosine, knowhere::NormalizeVecs (fixtures
tests/golden/cosine/*.npz from tests/golden/make_cosine_golden.py) -- and live against oracle/_ref/libknowhere_kref.so.
Bar: ids equal, distances bit-equal.  The GPU side is tests/test_gpu_cosine.py."""
import numpy as np
import pytest

from helpers import load_cosine_golden
from oracle import binding as ob


def test_normalize_matches_reference(port):
    zf, _, _ = load_cosine_golden()
    xn, norms = port.normalize(zf["xb"])
    assert xn.tobytes() == zf["normalized"].tobytes() and norms.tobytes() == zf["norms"].tobytes()
    assert norms[5] == 1.0 and (xn[5] == 0).all()                      # zero row untouched
    assert norms[6] == 1.0 and xn[6].tobytes() == zf["xb"][6].tobytes()  # unit row untouched (|1 - n^2| <= 1e-5)
    assert port.inverse_l2_norms(zf["xb"]).tobytes() == zf["inv_norms"].tobytes()


@pytest.mark.parametrize("k", [1, 10, 120])
def test_flat_cosine_matches_reference(port, k):
    zf, _, _ = load_cosine_golden()
    D, I = port.flat_cosine_search(zf["xb"], zf["xq"], k)
    assert (I == zf[f"flat_I_{k}"]).all() and D.tobytes() == zf[f"flat_D_{k}"].tobytes()


def test_flat_cosine_bitset(port):
    zf, _, _ = load_cosine_golden()
    D, I = port.flat_cosine_search(zf["xb"], zf["xq"], 10, zf["bitset"])
    assert (I == zf["flat_I_bs"]).all() and D.tobytes() == zf["flat_D_bs"].tobytes()


@pytest.mark.parametrize("k,nprobe", [(1, 1), (10, 4), (10, 16), (120, 16)])
def test_ivfflat_cosine_matches_reference(port, k, nprobe):
    zf, zi, ix = load_cosine_golden()
    qn, _ = port.normalize(zf["xq"])
    D, I = port.search(ix, qn, k, nprobe)
    assert (I == zi[f"I_{k}_{nprobe}"]).all() and D.tobytes() == zi[f"D_{k}_{nprobe}"].tobytes()


def test_ivfflat_cosine_bitset_and_stored_norm_semantics(port):
    zf, zi, ix = load_cosine_golden()
    qn, _ = port.normalize(zf["xq"])
    D, I = port.search(ix, qn, 10, 8, zf["bitset"], zf["xb"].shape[0])
    assert (I == zi["I_bs"]).all() and D.tobytes() == zi["D_bs"].tobytes()
    # rows are stored RAW with their norm beside them (IndexIVFFlat.cpp:516-524), ids ascending inside a list
    rows = zi["codes"].view(np.float32).reshape(-1, zf["xb"].shape[1])
    assert rows.tobytes() == zf["xb"][zi["ids"]].tobytes()
    assert zi["norms"].tobytes() == zf["norms"][zi["ids"]].tobytes()
    # and ip / norm is not the same float as the inner product with the normalised row: the semantics matter
    ip_norm = np.array([port.fvec_inner_product(qn[0], r) for r in rows[:400]], np.float32) / zi["norms"][:400]
    ip_unit = np.array([port.fvec_inner_product(qn[0], r) for r in zf["normalized"][zi["ids"][:400]]], np.float32)
    assert (ip_norm != ip_unit).any()


def test_oracle_equals_reference_live(port, kref):
    """other seeds, dimensions and list counts than the fixture"""
    r = np.random.default_rng(7)
    for d, nb, nlist in ((8, 900, 4), (33, 2500, 20)):
        xb = (r.random((nb, d), dtype=np.float32) * 4 - 2).astype(np.float32)
        xq = (r.random((11, d), dtype=np.float32) * 2 - 1).astype(np.float32)
        a, na = port.normalize(xb)
        b, nb_ = kref.normalize(xb)
        assert a.tobytes() == b.tobytes() and na.tobytes() == nb_.tobytes()
        D, I = port.flat_cosine_search(xb, xq, 7)
        D2, I2, inv = kref.flat_cosine_search(xb, xq, 7)
        assert (I == I2).all() and D.tobytes() == D2.tobytes()
        h = kref.ivfflat_create(d, nlist)
        kref.ivfflat_train(h, xb, niter=4)
        kref.ivfflat_add(h, xb)
        ix = ob.IndexData(ob.IVF_FLAT, ob.IP, d, nlist)
        ix.centroids = kref.ivfflat_centroids(h, d, nlist)
        ix.list_codes, ix.list_ids, ix.list_norms = kref.ivfflat_lists(h, d, nlist)
        qn, _ = port.normalize(xq)
        for k, nprobe in ((5, 2), (100, nlist)):
            Do, Io = port.search(ix, qn, k, nprobe)
            Dr, Ir = kref.ivfflat_search(h, xq, k, nprobe)
            assert (Io == Ir).all() and Do.tobytes() == Dr.tobytes()
        kref.ivfflat_destroy(h)
