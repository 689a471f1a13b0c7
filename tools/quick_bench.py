"""tools/quick_bench.py -- kernel-level timing on synthetic random codes (no training).

Scan cost does not depend on code values (the systolic ADC is conflict-free by construction), so
random codes / centroids / codebooks give representative per-stage timings at any scale without
building a real index.  Not a parity tool and not the contract bench (see bench.py).
"""
import argparse
import json
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from knowhere_amd import GpuIndex  # noqa: E402
from knowhere_amd import index as kidx  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="pq", choices=["pq", "flat", "sq8", "bf"])
    ap.add_argument("--nb", type=int, default=10_000_000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--nprobe", type=int, default=64)
    ap.add_argument("--nq", type=int, default=10000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--m", type=int, default=32)
    ap.add_argument("--metric", default="l2")
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g_ = torch.Generator(device=dev)
    g_.manual_seed(1)
    metric = kidx.L2 if a.metric == "l2" else kidx.IP
    kind = {"pq": kidx.IVF_PQ, "flat": kidx.IVF_FLAT, "sq8": kidx.IVF_SQ8, "bf": kidx.BRUTE_FORCE}[a.kind]
    t0 = time.time()
    q = torch.rand((a.nq, a.d), device=dev, generator=g_) * 100
    if kind == kidx.BRUTE_FORCE:
        g = GpuIndex(kind, metric, a.d)
        x = torch.rand((a.nb, a.d), device=dev, generator=g_) * 100
        g.add_vectors_device(x)
        del x
        code_size = a.d * 4
    else:
        g = GpuIndex(kind, metric, a.d, nlist=a.nlist, pq_m=a.m)
        cent = torch.rand((a.nlist, a.d), device=dev, generator=g_) * 100
        g.set_coarse_device(cent)
        if kind == kidx.IVF_PQ:
            cb = (torch.rand((a.m, 256, a.d // a.m), generator=torch.Generator().manual_seed(2)) * 20 - 10).numpy()
            g.set_pq(cb)
            code_size = a.m
        elif kind == kidx.IVF_SQ8:
            g.set_sq(np.full(a.d, -50, np.float32), np.full(a.d, 100, np.float32))
            code_size = a.d
        else:
            code_size = a.d * 4
        # mildly unbalanced lists
        w = np.random.default_rng(3).gamma(8.0, 1.0, a.nlist)
        sizes = np.floor(w / w.sum() * a.nb).astype(np.int64)
        sizes[0] += a.nb - sizes.sum()
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        if kind == kidx.IVF_FLAT:
            codes = (torch.rand((a.nb, a.d), device=dev, generator=g_) * 100).view(torch.uint8).reshape(a.nb, -1)
        else:
            codes = torch.randint(0, 256, (a.nb, code_size), device=dev, dtype=torch.uint8, generator=g_)
        ids = torch.arange(a.nb, device=dev, dtype=torch.int64)
        g.set_lists_device(off, codes, ids)
        del codes, ids
    torch.cuda.synchronize()
    build_s = time.time() - t0
    g.profile_enable(True)
    D, I = g.search_device(q, a.k, a.nprobe)  # warm-up (allocates scratch)
    torch.cuda.synchronize()
    g.profile_reset()
    t0 = time.time()
    for _ in range(a.iters):
        g.search_device(q, a.k, a.nprobe, out=(D, I))
    torch.cuda.synchronize()
    wall = (time.time() - t0) / a.iters
    p = g.profile_get()
    ms = [m / a.iters for m in p["ms"]]
    scan_bytes = p["scan_bytes"] / a.iters if kind != kidx.BRUTE_FORCE else float(a.nq) * a.nb * code_size
    res = dict(kind=a.kind, nb=a.nb, nlist=a.nlist, nprobe=a.nprobe, nq=a.nq, k=a.k, m=a.m, build_s=round(build_s, 2),
               wall_ms=round(wall * 1e3, 3), qps=round(a.nq / wall, 1),
               stage_ms=dict(coarse=round(ms[0], 3), group=round(ms[1], 3), lut=round(ms[2], 3),
                             scan=round(ms[3], 3), merge=round(ms[4], 3)),
               scan_bytes=scan_bytes, scan_algo_GBps=round(scan_bytes / (ms[3] * 1e-3) / 1e9, 1) if ms[3] > 0 else None,
               device_GB=round(g.device_bytes / 1e9, 2), coarse_fallback_queries=p["coarse_fallback_queries"], valid_ids=int((I >= 0).sum().item()))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
