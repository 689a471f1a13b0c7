"""tools/range_c3.py -- RangeSearch on the C3 index itself (IVF-PQ 100M x 128, nlist 16384), on the GPU box.

VERDICT round 2, item 7: "validate nlist 16384 (C3's own index) on hardware".  The host cannot walk 100M rows per
query in the oracle, so the check is three-fold:
  1. rank waves vs one pass over all 16384 lists (KNHIP_RANGE_NO_WAVES=1): lims, ids, order and distance bits equal;
  2. every reported distance is inside the radius, no id twice per query, and equal to the Search() distance of the same
     id where Search() (nprobe = nlist region covered) reports it;
  3. a query's hits are a prefix-consistent subset: with max_empty = 0 (all lists) the result contains the result of
     every early-stop setting.
Prints one JSON line with the timings and the ranks scanned per setting."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from knowhere_amd import build as kb, index as kidx

nb = int(os.environ.get("RANGE_NB", 100_000_000))
nlist = int(os.environ.get("RANGE_NLIST", 16384))
nq = int(os.environ.get("RANGE_NQ", 64))
d, m = 128, 32
dev = torch.device("cuda:0")
ncenter = 1 << max(4, int(round(np.log2(max(nb / 160.0, 16.0)))))
spec = kb.DataSpec(nb, d, kind="mixture", seed=42, ncenter=ncenter, sigma=0.35, latent=0)
t0 = time.time()
built = kb.build_ivf(spec, kidx.IVF_PQ, kidx.L2, nlist, m, device=str(dev), train_per_centroid=256, niter=10)
g = built.to_gpu_index(device=0)
xq = kb.queries(spec, nq, dev)
torch.cuda.synchronize()
build_s = time.time() - t0
D, I = g.search_device(xq, 100, 128)
D, I = D.cpu().numpy(), I.cpu().numpy()
radius = float(np.median(D[:, 20]))  # ~20 hits per query inside the probed region
xq_h = xq.cpu().numpy()
out = {"nb": nb, "nlist": nlist, "nq": nq, "radius": radius, "build_s": round(build_s, 1), "settings": []}


def same(a, b):
    return (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
            and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32)))


full = None
for max_empty in (0, 2, 16):
    t = time.time()
    got = g.range_search(xq_h, radius, max_empty)
    ms = (time.time() - t) * 1e3
    ranks = g.last_range_ranks()
    os.environ["KNHIP_RANGE_NO_WAVES"] = "1"
    t = time.time()
    one = g.range_search(xq_h, radius, max_empty)
    ms_one = (time.time() - t) * 1e3
    del os.environ["KNHIP_RANGE_NO_WAVES"]
    lims, ids, dis = got
    ok = same(got, one) and bool(np.all(dis < radius))
    for q in range(nq):
        seg = ids[lims[q]:lims[q + 1]]
        ok = ok and len(set(seg.tolist())) == len(seg)
        # Search() distances of the same ids
        pos = {int(i): float(x) for i, x in zip(I[q], D[q]) if i >= 0}
        for i, x in zip(seg.tolist(), dis[lims[q]:lims[q + 1]].tolist()):
            if i in pos:
                ok = ok and np.float32(pos[i]) == np.float32(x)
    if max_empty == 0:
        full = got
    else:
        for q in range(nq):
            ok = ok and set(ids[lims[q]:lims[q + 1]].tolist()) <= set(full[1][full[0][q]:full[0][q + 1]].tolist())
    out["settings"].append({"max_empty": max_empty, "hits": int(lims[-1]), "ranks_scanned": ranks,
                            "ms_waves": round(ms, 1), "ms_one_pass": round(ms_one, 1), "checks_ok": bool(ok)})
print(json.dumps(out))
assert all(s["checks_ok"] for s in out["settings"])
