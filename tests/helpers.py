"""tests/helpers.py -- fixture loading shared by CPU and GPU tests."""
import glob
import os

import numpy as np

from oracle import binding as ob

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files():
    return sorted(p for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if not os.path.basename(p).startswith("blob_"))


def golden_blobs():
    """index bytes written by the reference's faiss::write_index (tests/golden/make_golden_blobs.py)"""
    return sorted(glob.glob(os.path.join(GOLDEN, "blob_*.npz")))


def load_golden(path):
    z = np.load(path)
    kind, metric, d = int(z["kind"]), int(z["metric"]), int(z["d"])
    ix = ob.IndexData(kind, metric, d, int(z["nlist"]), int(z["M"]), int(z["nbits"]))
    ix.use_precomputed_table = int(z["use_precomputed_table"])
    if kind == ob.FLAT:
        ix.base = z["base"]
    else:
        ix.centroids = z["centroids"]
        sizes = z["list_sizes"]
        off = np.concatenate([[0], np.cumsum(sizes)])
        codes, ids = z["codes"], z["ids"]
        ix.list_codes = [np.ascontiguousarray(codes[off[l]:off[l + 1]]) for l in range(ix.nlist)]
        ix.list_ids = [np.ascontiguousarray(ids[off[l]:off[l + 1]]) for l in range(ix.nlist)]
        if kind == ob.IVF_PQ:
            ix.pq_centroids = z["pq_centroids"]
        if kind == ob.IVF_SQ8:
            ix.sq_trained = z["sq_trained"]
    cases = []
    for ci, (k, nprobe, use_bs) in enumerate(z["cases"]):
        cases.append(dict(k=int(k), nprobe=int(nprobe), bitset=z["bitset"] if use_bs else None,
                          nbits=int(z["nb"]) if use_bs else 0, D=z[f"D{ci}"], I=z[f"I{ci}"]))
    return ix, z["xq"], cases


def load_golden_range(path):
    """range-search cases of a golden fixture: (radius, [dict(max_empty, bitset, nbits, lims, ids, dis)])"""
    z = np.load(path)
    out = []
    for ri, (max_empty, use_bs) in enumerate(z["range_cases"]):
        out.append(dict(max_empty=int(max_empty), bitset=z["bitset"] if use_bs else None,
                        nbits=int(z["nb"]) if use_bs else 0, lims=z[f"RL{ri}"], ids=z[f"RI{ri}"], dis=z[f"RD{ri}"]))
    return float(z["range_radius"]), out


def finish_ivfpq(port, ix):
    """the precomputed term-2 table is derived data: recompute it with the restated formula"""
    if ix.kind == ob.IVF_PQ and ix.metric == ob.L2 and ix.use_precomputed_table == 1 and ix.precomputed_table is None:
        ix.precomputed_table = port.pq_precompute_table(ix.d, ix.M, ix.nbits, ix.centroids, ix.pq_centroids)
    return ix


def load_cosine_golden():
    """tests/golden/cosine/*.npz (tests/golden/make_cosine_golden.py): flat fixture, IVF_FLAT fixture as IndexData"""
    zf = np.load(os.path.join(GOLDEN, "cosine", "flat.npz"))
    zi = np.load(os.path.join(GOLDEN, "cosine", "ivfflat.npz"))
    d = zf["xb"].shape[1]
    sizes = zi["list_sizes"]
    nlist = len(sizes)
    off = np.concatenate([[0], np.cumsum(sizes)])
    ix = ob.IndexData(ob.IVF_FLAT, ob.IP, d, nlist)
    ix.centroids = zi["centroids"]
    ix.list_codes = [np.ascontiguousarray(zi["codes"][off[l]:off[l + 1]]) for l in range(nlist)]
    ix.list_ids = [np.ascontiguousarray(zi["ids"][off[l]:off[l + 1]]) for l in range(nlist)]
    ix.list_norms = [np.ascontiguousarray(zi["norms"][off[l]:off[l + 1]]) for l in range(nlist)]
    return zf, zi, ix


def sort_lists_by_id(ix):
    """every inverted list in ascending id order (codes follow): the storage order Knowhere's own Add produces (ids are the
    running row numbers, appended), and the order the library keeps its lists in -- which candidate of several tied at
    the k-th distance is returned depends on the scan order (conftest.assert_parity), so a fixture with shuffled ids is
    brought to that order on BOTH sides before it is compared"""
    for l in range(len(ix.list_ids)):
        o = np.argsort(ix.list_ids[l], kind="stable")
        ix.list_ids[l] = np.ascontiguousarray(ix.list_ids[l][o])
        ix.list_codes[l] = np.ascontiguousarray(ix.list_codes[l][o])
        if getattr(ix, "list_norms", None):
            ix.list_norms[l] = np.ascontiguousarray(ix.list_norms[l][o])
    return ix
