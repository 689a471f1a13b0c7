"""tests/golden/make_golden.py -- regenerate the golden fixtures FROM THE REFERENCE ITSELF.

Run in the dev container (needs oracle/_ref, i.e. /root/reference):
    python tests/golden/make_golden.py
Each fixture holds a small index TRAINED AND POPULATED BY THE REFERENCE's FAISS (k-means seed
1234, reference thirdparty/faiss/faiss/Clustering.h:24-77) as plain arrays, the queries, and the
(distances, ids) the reference returns when driven the way Knowhere drives it (one query per
search call; oracle/ref_driver.cpp).  The reference's own tests hold no golden vectors for this
path -- these pin the oracle and the HIP path on boxes without the reference.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def gen(n, d, seed):
    return (np.random.default_rng(seed).random((n, d), dtype=np.float32) * 100).astype(np.float32)


def main():
    ref = ob.Ref()
    nb, nq, d, nlist, M = 3000, 24, 32, 16, 8
    xb, xq = gen(nb, d, 42), gen(nq, d, 44)  # seeds: reference tests/ut/test_gpu_search.cc:64-65
    bitset = np.zeros((nb + 7) // 8, np.uint8)
    filt = np.random.default_rng(7).random(nb) < 0.4  # 40 % filtered, cf. test_gpu_search.cc:204-243
    for i in np.nonzero(filt)[0]:
        bitset[i >> 3] |= 1 << (i & 7)
    for kind, name in ((ob.FLAT, "flat"), (ob.IVF_FLAT, "ivfflat"), (ob.IVF_PQ, "ivfpq"), (ob.IVF_SQ8, "ivfsq8")):
        for metric, mname in ((ob.L2, "l2"), (ob.IP, "ip")):
            h = ref.create(kind, metric, d, nlist, M, 8)
            ref.train_add(h, xb)
            ix = ref.export(h, kind, metric, d, nlist, M, 8)
            arrs = dict(kind=kind, metric=metric, d=d, nlist=ix.nlist, M=ix.M, nbits=8, xq=xq, bitset=bitset,
                        nb=nb, use_precomputed_table=ix.use_precomputed_table)
            if kind == ob.FLAT:
                arrs["base"] = ix.base
            else:
                arrs["centroids"] = ix.centroids
                arrs["list_sizes"] = np.array([len(i) for i in ix.list_ids], np.int64)
                arrs["codes"] = np.concatenate([c.reshape(-1, ix.code_size) for c in ix.list_codes])
                arrs["ids"] = np.concatenate(ix.list_ids)
                if kind == ob.IVF_PQ:
                    arrs["pq_centroids"] = ix.pq_centroids
                if kind == ob.IVF_SQ8:
                    arrs["sq_trained"] = ix.sq_trained
            cases = []
            for ci, (k, nprobe, use_bs) in enumerate(((10, 4, False), (1, 1, False), (25, 16, False), (10, 8, True))):
                D, I = ref.search(h, xq, k, nprobe, bitset if use_bs else None, nb if use_bs else 0)
                arrs[f"D{ci}"], arrs[f"I{ci}"] = D, I
                cases.append((k, nprobe, int(use_bs)))
            arrs["cases"] = np.array(cases, np.int64)
            # range search cases (reference: IvfIndexNode::RangeSearch semantics, oracle/ref_driver.cpp
            # ref_range_search): radius = median 10th-best distance; (max_empty_result_buckets, bitset?)
            D10, _ = ref.search(h, xq, 10, ix.nlist if kind != ob.FLAT else 1)
            radius = np.float32(np.median(D10[:, 9]))
            arrs["range_radius"] = radius
            rcases = []
            for ri, (max_empty, use_bs) in enumerate(((2, False), (0, False), (1, True))):
                lims, rids, rdis = ref.range_search(h, xq, radius, max_empty, bitset if use_bs else None,
                                                    nb if use_bs else 0)
                arrs[f"RL{ri}"], arrs[f"RI{ri}"], arrs[f"RD{ri}"] = lims, rids, rdis
                rcases.append((max_empty, int(use_bs)))
            arrs["range_cases"] = np.array(rcases, np.int64)
            np.savez_compressed(os.path.join(OUT, f"{name}_{mname}.npz"), **arrs)
            ref.destroy(h)
            print("wrote", name, mname)
    headline_shapes(ref)


def headline_shapes(ref):
    """d = 128 fixtures that reach the kernels of the BASELINE configurations: IVF-PQ m = 32 (the staggered ADC scans
    pq_scan_q4 / pq_scan_v2, rank-0 dump phase at k = 100, range search), IVF-Flat and IVF-SQ8 through the MFMA
    prefilter (enough queries per list for the automatic switch)."""
    d, nq = 128, 48
    for kind, metric, name, nb, nlist, M, cases in (
            (ob.IVF_PQ, ob.L2, "h128_ivfpq32_l2", 12000, 32, 32, ((10, 8, False), (100, 32, False), (10, 8, True), (64, 4, False))),
            (ob.IVF_PQ, ob.IP, "h128_ivfpq32_ip", 12000, 32, 32, ((10, 8, False), (100, 32, False))),
            (ob.IVF_FLAT, ob.L2, "h128_ivfflat_l2", 4000, 16, 0, ((10, 8, False), (100, 16, False), (10, 8, True))),
            (ob.IVF_SQ8, ob.IP, "h128_ivfsq8_ip", 4000, 16, 0, ((10, 8, False), (100, 16, False), (10, 8, True)))):
        xb, xq = gen(nb, d, 42), gen(nq, d, 44)
        bitset = np.zeros((nb + 7) // 8, np.uint8)
        filt = np.random.default_rng(7).random(nb) < 0.4
        for i in np.nonzero(filt)[0]:
            bitset[i >> 3] |= 1 << (i & 7)
        h = ref.create(kind, metric, d, nlist, max(M, 1), 8)
        ref.train_add(h, xb)
        ix = ref.export(h, kind, metric, d, nlist, max(M, 1), 8)
        arrs = dict(kind=kind, metric=metric, d=d, nlist=ix.nlist, M=ix.M, nbits=8, xq=xq, bitset=bitset, nb=nb,
                    use_precomputed_table=ix.use_precomputed_table, centroids=ix.centroids,
                    list_sizes=np.array([len(i) for i in ix.list_ids], np.int64),
                    codes=np.concatenate([c.reshape(-1, ix.code_size) for c in ix.list_codes]),
                    ids=np.concatenate(ix.list_ids))
        if kind == ob.IVF_PQ:
            arrs["pq_centroids"] = ix.pq_centroids
        if kind == ob.IVF_SQ8:
            arrs["sq_trained"] = ix.sq_trained
        cl = []
        for ci, (k, nprobe, use_bs) in enumerate(cases):
            D, I = ref.search(h, xq, k, nprobe, bitset if use_bs else None, nb if use_bs else 0)
            arrs[f"D{ci}"], arrs[f"I{ci}"] = D, I
            cl.append((k, nprobe, int(use_bs)))
        arrs["cases"] = np.array(cl, np.int64)
        D10, _ = ref.search(h, xq, 10, ix.nlist)
        radius = np.float32(np.median(D10[:, 9]))
        arrs["range_radius"] = radius
        rcases = []
        for ri, (max_empty, use_bs) in enumerate(((2, False), (1, True))):
            lims, rids, rdis = ref.range_search(h, xq, radius, max_empty, bitset if use_bs else None, nb if use_bs else 0)
            arrs[f"RL{ri}"], arrs[f"RI{ri}"], arrs[f"RD{ri}"] = lims, rids, rdis
            rcases.append((max_empty, int(use_bs)))
        arrs["range_cases"] = np.array(rcases, np.int64)
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **arrs)
        ref.destroy(h)
        print("wrote", name)


if __name__ == "__main__":
    main()
