"""tools/recall_sweep.py -- recall@10 and step time of IVF-PQ(+refine) vs refine_k / data shape."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from knowhere_amd import build as kb, index as kidx

ap = argparse.ArgumentParser()
ap.add_argument("--nb", type=int, default=10_000_000); ap.add_argument("--nlist", type=int, default=4096)
ap.add_argument("--nprobe", type=int, default=64); ap.add_argument("--nq", type=int, default=10000)
ap.add_argument("--ncenter", type=int, default=0); ap.add_argument("--latents", default="0")
ap.add_argument("--rks", default="10,16,32,48,64,100"); ap.add_argument("--ngt", type=int, default=1000)
ap.add_argument("--niter", type=int, default=10); ap.add_argument("--tpc", type=int, default=64)
a = ap.parse_args()
dev = "cuda:0"
ncenter = a.ncenter or 1 << int(round(np.log2(a.nb / 160.0)))
for latent in [int(x) for x in a.latents.split(",")]:
    spec = kb.DataSpec(a.nb, 128, ncenter=ncenter, sigma=0.35, latent=latent)
    t0 = time.time()
    built = kb.build_ivf(spec, kidx.IVF_PQ, kidx.L2, a.nlist, 32, keep_vectors=True, niter=a.niter, train_per_centroid=a.tpc)
    g = built.to_gpu_index()
    xq = kb.queries(spec, a.nq, dev)
    sizes = built.list_offsets[1:] - built.list_offsets[:-1]
    imb = float((sizes.astype(np.float64) ** 2).sum() * a.nlist / float(sizes.sum()) ** 2)
    _, gt = kb.ground_truth(spec, xq[:a.ngt], 10)
    print(json.dumps(dict(latent=latent, ncenter=ncenter, build_s=round(time.time() - t0, 1), imbalance=round(imb, 2))), flush=True)
    g.profile_enable(True)
    for rk in [int(x) for x in a.rks.split(",")]:
        def step():
            D, I = g.search_device(xq, rk, a.nprobe)
            if rk > 10:
                return kidx.refine_device(kidx.L2, built.vectors, xq, I, 10)
            return D, I
        D, I = step(); torch.cuda.synchronize()
        rec = (I[:a.ngt].unsqueeze(2) == gt.unsqueeze(1)).any(2).float().mean().item()
        g.profile_reset(); t1 = time.time()
        for _ in range(3): step()
        torch.cuda.synchronize(); dt = (time.time() - t1) / 3
        p = g.profile_get()
        print(json.dumps(dict(refine_k=rk, recall10=round(rec, 4), ms=round(dt * 1e3, 2), qps=round(a.nq / dt),
                              scan_ms=round(p["ms"][3] / 3, 2), scan_GBps=round(p["scan_bytes"] / p["ms"][3] / 1e6, 1))), flush=True)
    g.close(); del built, g
    torch.cuda.empty_cache()
