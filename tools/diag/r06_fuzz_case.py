import os, sys, types
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch; torch.cuda.is_available()
from conftest import gen_data
from helpers import finish_ivfpq
from oracle import binding as ob
from knowhere_amd import GpuIndex
port = ob.Port()
seed = 10
r = np.random.default_rng(1000 + seed)
metric = int(r.integers(0, 2)); d = 128
nb = int(r.choice([3000, 20000, 90000, 250000]))
nlist = int(r.choice([4, 16, 64, 256])) if nb >= 20000 else int(r.choice([4, 16, 40]))
clustered = bool(r.integers(0, 2))
def _clustered(n, d, ncenter, sigma, seed):
    rr = np.random.default_rng(seed)
    c = rr.random((ncenter, d), dtype=np.float32) * 10.0
    return (c[rr.integers(0, ncenter, n)] + sigma * rr.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
xb = _clustered(nb, d, 200, 0.4, seed) if clustered else gen_data(nb, d, seed, -5.0, 5.0)
print("metric", metric, "nb", nb, "nlist", nlist, "clustered", clustered)
b = GpuIndex(2, metric, d, nlist, 32, 8, device=0); b.train(xb); b.add(xb)
sizes, codes, ids = b.get_lists()
ix = ob.IndexData(ob.IVF_PQ, metric, d, nlist, 32, 8)
ix.centroids, ix.pq_centroids = b.get_coarse(), b.get_pq()
pos = 0
for l in range(nlist):
    n = int(sizes[l]); ix.list_codes.append(codes[pos:pos+n]); ix.list_ids.append(ids[pos:pos+n]); pos += n
if metric == ob.L2: ix.use_precomputed_table = 1
ix = finish_ivfpq(port, ix)
b.close()
gs = {}
os.environ["KNHIP_PQF"] = "0"; gs["exact"] = GpuIndex.from_data(ix, device=0)
for form in ("decode", "half", "int8"):
    os.environ.update({"KNHIP_PQF": "1", "KNHIP_PQF_GUARD": "0", "KNHIP_PQF_FORM": form}); gs[form] = GpuIndex.from_data(ix, device=0)
os.environ.update({"KNHIP_PQF": "1", "KNHIP_PQF_GUARD": "0", "KNHIP_PQF_FORM": "decode", "KNHIP_TIES": "canonical"}); gs["decode-canonical-ties"] = GpuIndex.from_data(ix, device=0)
for v in ("KNHIP_PQF", "KNHIP_PQF_GUARD", "KNHIP_PQF_FORM", "KNHIP_TIES"): os.environ.pop(v, None)
for case in range(5):
    nq = int(r.choice([1, 3, 40, 130, 600])); k = int(r.choice([1, 10, 100, 128, 500, 1000])); nprobe = int(min(nlist, r.choice([1, 2, 8, 32, 256])))
    xq = (xb[r.integers(0, nb, nq)] + 0.05 * r.standard_normal((nq, d), dtype=np.float32)).astype(np.float32) if clustered else gen_data(nq, d, 77 + case, -5.0, 5.0)
    frac = float(r.choice([0.0, 0.0, 0.3, 0.9, 0.995]))
    bs = np.packbits(r.random(nb) < frac, bitorder="little") if frac > 0 else None
    if case != 4: continue
    Do, Io = port.search(ix, xq, k, nprobe, bs, nb if bs is not None else 0)
    print("case", case, "nq", nq, "k", k, "nprobe", nprobe, "frac", frac)
    for name, g in gs.items():
        g.profile_enable(True); g.profile_reset()
        D, I = g.search(xq, k, nprobe, bs, nb if bs is not None else 0)
        p = g.profile_get()
        bad = np.flatnonzero((I != Io).any(1) | (D.view(np.uint32) != Do.view(np.uint32)).any(1))
        print(name, "form", p["pq_filter_form"], "mscan q", p["mscan_queries"], "ovf", p["mscan_overflow_queries"], "ties", p["tie_queries"], "anom", p["tie_anomalies"], "queries differing from the ORACLE:", len(bad), bad[:8])
        if name == "decode-canonical-ties":
            os.environ["KNHIP_TIES"] = "canonical"
            D, I = g.search(xq, k, nprobe, bs, nb if bs is not None else 0)
            os.environ.pop("KNHIP_TIES")
            bad = np.flatnonzero((I != Io).any(1) | (D.view(np.uint32) != Do.view(np.uint32)).any(1))
            print("   with KNHIP_TIES=canonical at search time: differing", len(bad), bad[:8])
        if len(bad):
            q = bad[0]; j = np.flatnonzero(I[q] != Io[q])
            print("   q", q, "first differing rank", j[:6], "gpu ids", I[q][j[:6]], "oracle ids", Io[q][j[:6]], "gpu d", D[q][j[:6]], "oracle d", Do[q][j[:6]], "valid results oracle", int((Io[q] >= 0).sum()), "gpu", int((I[q] >= 0).sum()))
