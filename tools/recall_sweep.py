"""tools/recall_sweep.py -- recall@10 of IVF-PQ(+refine) vs data intrinsic dimension / refine_k."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from knowhere_amd import build as kb, index as kidx

ap = argparse.ArgumentParser()
ap.add_argument("--nb", type=int, default=10_000_000); ap.add_argument("--nlist", type=int, default=4096)
ap.add_argument("--nprobe", type=int, default=64); ap.add_argument("--nq", type=int, default=2000)
ap.add_argument("--ncenter", type=int, default=65536)
a = ap.parse_args()
dev = "cuda:0"
for latent, sigma in ((0, 0.35), (32, 0.35), (16, 0.35), (8, 0.35)):
    spec = kb.DataSpec(a.nb, 128, ncenter=a.ncenter, sigma=sigma, latent=latent)
    t0 = time.time()
    built = kb.build_ivf(spec, kidx.IVF_PQ, kidx.L2, a.nlist, 32, keep_vectors=True)
    g = built.to_gpu_index()
    xq = kb.queries(spec, a.nq, dev)
    sizes = built.list_offsets[1:] - built.list_offsets[:-1]
    imb = float((sizes.astype(np.float64) ** 2).sum() * a.nlist / float(sizes.sum()) ** 2)
    _, gt = kb.ground_truth(spec, xq, 10)
    row = dict(latent=latent, sigma=sigma, build_s=round(time.time() - t0, 1), imbalance=round(imb, 2))
    for rk in (10, 100, 200, 400, 1000):
        D, I = g.search_device(xq, rk, a.nprobe)
        cand = (I.unsqueeze(2) == gt.unsqueeze(1)).any(1).float().mean().item() if rk >= 10 else 0
        if rk > 10:
            Dr, Ir = kidx.refine_device(kidx.L2, built.vectors, xq, I, 10)
        else:
            Ir = I
        rec = (Ir.unsqueeze(2) == gt.unsqueeze(1)).any(2).float().mean().item()
        row[f"rk{rk}"] = round(rec, 4)
        row[f"cand{rk}"] = round(cand, 4)
    print(json.dumps(row), flush=True)
    g.close(); del built, g
    torch.cuda.empty_cache()
