import os, sys
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch; torch.cuda.is_available()
from conftest import gen_data
from helpers import finish_ivfpq
from oracle import binding as ob
from knowhere_amd import GpuIndex
port = ob.Port()
os.environ["KNHIP_TIES"] = "canonical"
def clustered(n, d, ncenter, sigma, seed):
    rr = np.random.default_rng(seed)
    c = rr.random((ncenter, d), dtype=np.float32) * 10.0
    return (c[rr.integers(0, ncenter, n)] + sigma * rr.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
nb, d, nlist = 3000, 128, 40
xb = clustered(nb, d, 200, 0.4, 10)
r = np.random.default_rng(5)
xq = (xb[r.integers(0, nb, 130)] + 0.05 * r.standard_normal((130, d), dtype=np.float32)).astype(np.float32)
bs = np.packbits(r.random(nb) < 0.3, bitorder="little")
for metric in (0, 1):
    b = GpuIndex(2, metric, d, nlist, 32, 8, device=0); b.train(xb); b.add(xb)
    sizes, codes, ids = b.get_lists()
    ix = ob.IndexData(ob.IVF_PQ, metric, d, nlist, 32, 8)
    ix.centroids, ix.pq_centroids = b.get_coarse(), b.get_pq()
    pos = 0
    for l in range(nlist):
        n = int(sizes[l]); ix.list_codes.append(codes[pos:pos+n]); ix.list_ids.append(ids[pos:pos+n]); pos += n
    if metric == ob.L2: ix.use_precomputed_table = 1
    ix = finish_ivfpq(port, ix); b.close()
    os.environ.update({"KNHIP_PQF": "1", "KNHIP_PQF_GUARD": "0", "KNHIP_PQF_FORM": "decode"}); g = GpuIndex.from_data(ix, device=0)
    os.environ["KNHIP_PQF"] = "0"; g0 = GpuIndex.from_data(ix, device=0)
    for v in ("KNHIP_PQF", "KNHIP_PQF_GUARD", "KNHIP_PQF_FORM"): os.environ.pop(v, None)
    g.profile_enable(True)
    for k in (500, 200, 129, 128, 100):
        for nprobe in (2, 8):
            for b_ in (None, bs):
                Do, Io = port.search(ix, xq, k, nprobe, b_, nb if b_ is not None else 0)
                res = []
                for rep in range(3):
                    g.profile_reset()
                    D, I = g.search(xq, k, nprobe, b_, nb if b_ is not None else 0)
                    p = g.profile_get()
                    bad = np.flatnonzero((I != Io).any(1) | (D.view(np.uint32) != Do.view(np.uint32)).any(1))
                    res.append(len(bad))
                D0, I0 = g0.search(xq, k, nprobe, b_, nb if b_ is not None else 0)
                bad0 = int(((I0 != Io).any(1)).sum())
                print(f"metric {metric} k {k} nprobe {nprobe} bitset {b_ is not None}: prefilter path bad queries per run {res} (overflowed {p['mscan_overflow_queries']}, finished {p['mscan_queries']}); exact path bad {bad0}")
