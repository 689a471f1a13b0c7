"""tools/pq_f16_candidates.py -- how many rows would a HALF-PRECISION prefilter form of the IVF-PQ ADC scan let through?

DESIGN.md section 7 item 2: an approximate ADC pass (per-query table -2<q_m, cb> in f16, 32 v_pk_add_f16 per vector,
per-vector constant sum_m precomp[list][m][c_m]) runs the loop 2.1x faster (tools/ubench/adc_loop), but its error
bound is ~33 * 2^-11 * sum_m max_c |table_m|  (the kernel written afterwards uses an integer table whose half additions
are exact: half that bound -- the counts below are therefore upper bounds of what it lets through).  This script builds a real IVF-PQ index through the product ABI, replays
both the exact (fp32, reference order) and the f16 sums with torch for a sample of queries, and reports
  * the observed |approx - exact| against the bound,
  * rows with approx <= tau + bound (what the filter would hand to the exact finish) vs k, tau = exact k-th distance.
Measurement tooling only (not a product path).  --selftest runs the arithmetic on random tensors on the CPU."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def replay(q, cen, cb, codes_by_list, keys, k):
    """q [d]; cen [nlist, d]; cb [M, 256, dsub]; codes_by_list(l) -> uint8 [len, M]; keys: probed lists (coarse order).
    -> dict of per-query statistics"""
    M, ksub, dsub = cb.shape
    qm = q.view(M, dsub)
    T = torch.einsum("md,mcd->mc", qm, cb)                       # <q_m, cb[m][c]>
    Qh = (-2.0 * T).half()                                       # the per-query table of the prefilter
    cbn = (cb * cb).sum(2)                                       # ||cb||^2
    ar = torch.arange(M, device=q.device)
    exact_all, approx_all = [], []
    for l in keys.tolist():
        codes = codes_by_list(l).long()                          # [len, M]
        if codes.shape[0] == 0:
            continue
        c = cen[l]
        dis0 = ((q - c) ** 2).sum()
        P = cbn + 2.0 * torch.einsum("md,mcd->mc", c.view(M, dsub), cb)   # precomputed term-2 table row
        lut = P - 2.0 * T                                        # fp32 table of the exact scan
        ex = torch.zeros(codes.shape[0], device=q.device)
        acc_h = torch.zeros(codes.shape[0], device=q.device, dtype=torch.float16)
        for m in range(M):                                       # sequential in m, as the reference sums
            ex = ex + lut[m, codes[:, m]]
            acc_h = acc_h + Qh[m, codes[:, m]]                   # one half-precision rounding per addition
        psum = P[ar.unsqueeze(0), codes].sum(1)                  # per-vector constant (computed at add time)
        exact_all.append(dis0 + ex)
        approx_all.append(dis0 + psum + acc_h.float())
    ex = torch.cat(exact_all)
    ap = torch.cat(approx_all)
    kk = min(k, ex.numel())
    tau = torch.topk(ex, kk, largest=False).values[-1]
    bound = 33.0 * 2.0 ** -11 * Qh.float().abs().max(1).values.sum()
    err = (ap - ex).abs().max()
    return dict(rows=ex.numel(), tau=float(tau), bound=float(bound), max_err=float(err),
                pass_bound=int((ap <= tau + bound).sum()), pass_emp=int((ap <= tau + err).sum()),
                pass_fp32=int((ex <= tau).sum()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nb", type=int, default=10_000_000)
    ap.add_argument("--nlist", type=int, default=2048)
    ap.add_argument("--nprobe", type=int, default=16)
    ap.add_argument("--nq", type=int, default=100)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--data", default="mixture", choices=["mixture", "uniform"])
    ap.add_argument("--selftest", action="store_true")
    a = ap.parse_args()
    if a.selftest:
        g = torch.Generator().manual_seed(1)
        d, M, nlist = 128, 32, 8
        cen = torch.randn((nlist, d), generator=g)
        cb = torch.randn((M, 256, d // M), generator=g) * 0.3
        lists = [torch.randint(0, 256, (300 + 10 * l, M), generator=g, dtype=torch.uint8) for l in range(nlist)]
        r = replay(torch.randn(d, generator=g), cen, cb, lambda l: lists[l], torch.arange(4), 10)
        print("selftest", r)
        assert r["max_err"] <= r["bound"] and r["pass_bound"] >= r["pass_fp32"] == 10
        return
    from knowhere_amd import build as kb
    from knowhere_amd import index as kidx
    dev = torch.device("cuda:0")
    ncenter = 1 << max(4, int(round(np.log2(max(a.nb / 160.0, 16.0)))))
    spec = kb.DataSpec(a.nb, 128, kind=a.data, seed=42, ncenter=ncenter, sigma=0.35)
    built = kb.build_ivf(spec, kidx.IVF_PQ, kidx.L2, a.nlist, 32, device=str(dev))
    xq = kb.queries(spec, a.nq, dev)
    off = built.list_offsets
    stats = []
    for i in range(a.nq):
        q = xq[i]
        d2 = ((built.centroids - q) ** 2).sum(1)
        keys = torch.topk(d2, a.nprobe, largest=False).indices
        stats.append(replay(q, built.centroids, built.codebooks, lambda l: built.codes[off[l]:off[l + 1]], keys, a.k))
    med = lambda f: float(np.median([f(s) for s in stats]))  # noqa: E731
    mx = lambda f: float(np.max([f(s) for s in stats]))      # noqa: E731
    print(f"data={a.data} nb={a.nb} nlist={a.nlist} nprobe={a.nprobe} k={a.k} queries={a.nq}: rows scanned / query "
          f"{med(lambda s: s['rows']):.0f}")
    print(f"  bound / tau: median {med(lambda s: s['bound'] / max(s['tau'], 1e-30)):.4f}   "
          f"observed max error / bound: median {med(lambda s: s['max_err'] / s['bound']):.4f} max {mx(lambda s: s['max_err'] / s['bound']):.4f}")
    print(f"  rows passing approx <= tau + bound: median {med(lambda s: s['pass_bound']):.0f} max {mx(lambda s: s['pass_bound']):.0f}"
          f"   (with the observed error instead of the bound: median {med(lambda s: s['pass_emp']):.0f}); exact <= tau: {a.k}")


if __name__ == "__main__":
    main()
