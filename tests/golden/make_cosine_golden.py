"""tests/golden/make_cosine_golden.py -- COSINE known answers from the classes Knowhere's nodes instantiate
(oracle/_ref/libknowhere_kref.so: IndexFlatCosine, IndexIVFFlatCosine, knowhere::NormalizeVecs, the fork's write_index).
Run in the dev container (needs /root/reference):  python tests/golden/make_cosine_golden.py
-> tests/golden/cosine/{flat,ivfflat}.npz"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import binding as ob  # noqa: E402

K = ob.KRef()
r = np.random.default_rng(2024)
d, nb, nq, nlist = 24, 3000, 16, 16
xb = (r.random((nb, d), dtype=np.float32) * 10 - 3).astype(np.float32)
xb[5] = 0                      # a zero row: inverse norm 1, stored norm 1
xb[6] = xb[6] / np.sqrt((xb[6].astype(np.float64) ** 2).sum())  # already unit: NormalizeVec leaves it alone
xb[100] = xb[50] * 3           # same direction, different length: a cosine tie
xq = (r.random((nq, d), dtype=np.float32) * 2 - 1).astype(np.float32)
bs = np.packbits(r.random(nb) < 0.5, bitorder="little")
out = dict(xb=xb, xq=xq, bitset=bs)
xn, norms = K.normalize(xb)
out["normalized"], out["norms"] = xn, norms
for k in (1, 10, 120):
    D, I, inv = K.flat_cosine_search(xb, xq, k)
    out[f"flat_D_{k}"], out[f"flat_I_{k}"] = D, I
D, I, inv = K.flat_cosine_search(xb, xq, 10, bs)
out["flat_D_bs"], out["flat_I_bs"], out["inv_norms"] = D, I, inv
out["flat_blob"] = K.flat_cosine_blob(xb)
np.savez_compressed(os.path.join(HERE, "cosine", "flat.npz"), **out)

h = K.ivfflat_create(d, nlist)
K.ivfflat_train(h, xb, niter=8)
K.ivfflat_add(h, xb[:2000])
K.ivfflat_add(h, xb[2000:])
cen = K.ivfflat_centroids(h, d, nlist)
codes, ids, lnorms = K.ivfflat_lists(h, d, nlist)
o2 = dict(centroids=cen, list_sizes=np.array([len(i) for i in ids], np.int64), codes=np.concatenate(codes),
          ids=np.concatenate(ids), norms=np.concatenate(lnorms))
for k, nprobe in ((1, 1), (10, 4), (10, 16), (120, 16)):
    D, I = K.ivfflat_search(h, xq, k, nprobe)
    o2[f"D_{k}_{nprobe}"], o2[f"I_{k}_{nprobe}"] = D, I
D, I = K.ivfflat_search(h, xq, 10, 8, bs, nb)
o2["D_bs"], o2["I_bs"] = D, I
o2["blob"] = K.ivfflat_blob(h, nb * d * 4 + nb * 16 + nlist * d * 4 + 65536)
np.savez_compressed(os.path.join(HERE, "cosine", "ivfflat.npz"), **o2)
K.ivfflat_destroy(h)
print("flat blob", out["flat_blob"][:4].tobytes(), len(out["flat_blob"]), "ivf blob", o2["blob"][:4].tobytes(), len(o2["blob"]))
