"""tools/profile_scan.py -- build a real (trained) IVF-PQ index once and run a few searches; meant to be
run under rocprofv3 (kernel stats / PMC passes)."""
import argparse, os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from knowhere_amd import build as kb, index as kidx
ap = argparse.ArgumentParser()
ap.add_argument("--nb", type=int, default=20_000_000); ap.add_argument("--nlist", type=int, default=4096)
ap.add_argument("--nprobe", type=int, default=64); ap.add_argument("--nq", type=int, default=10000)
ap.add_argument("--ks", default="10,100"); ap.add_argument("--iters", type=int, default=2)
a = ap.parse_args()
spec = kb.DataSpec(a.nb, 128, ncenter=1 << 17, sigma=0.35)
built = kb.build_ivf(spec, kidx.IVF_PQ, kidx.L2, a.nlist, 32)
g = built.to_gpu_index()
xq = kb.queries(spec, a.nq, "cuda:0")
g.profile_enable(True)
for k in [int(x) for x in a.ks.split(",")]:
    g.search_device(xq, k, a.nprobe); torch.cuda.synchronize(); g.profile_reset()
    for _ in range(a.iters):
        g.search_device(xq, k, a.nprobe)
    torch.cuda.synchronize()
    p = g.profile_get()
    print(json.dumps(dict(k=k, scan_ms=round(p["ms"][3] / a.iters, 3), scan_GBps=round(p["scan_bytes"] / p["ms"][3] / 1e6, 1),
                          coarse_ms=round(p["ms"][0] / a.iters, 3))), flush=True)
