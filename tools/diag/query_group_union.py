"""tools/diag/query_group_union.py -- what would a QUERY-GROUP-STATIONARY table cost at C3?

VERDICT round 4, item 1 proposes fixed groups of 16 queries (sorted by coarse key) whose 128 KB int8 table is written once
and kept while a workgroup walks the union of the group's probed lists.  The matrix instruction then spends 16 query slots
on every (list, group) it visits whether or not all 16 queries probe that list, so the work is
    sum_g |union of the probes of group g| x 16     against     nq x nprobe     for the (list, <= 16 queries) units of today.
This script measures that ratio on the bench's own coarse quantizer and query batch (no rows are added: only the trained
centroids and the queries' probe lists are needed), for the grouping the verdict names and for a greedy one that packs
queries by overlap of their probe sets."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

from knowhere_amd import build as kb
from knowhere_amd import index as kidx

nb, d, nlist, nprobe, nq = 100_000_000, 128, 16384, 128, 10000
dev = "cuda:0"
spec = kb.DataSpec(nb, d, kind="mixture", seed=42, ncenter=1 << int(round(np.log2(nb / 160.0))), sigma=0.35)
built = kb.build_ivf(spec, kidx.IVF_PQ, kidx.L2, nlist, 32, device=dev, train_only=True, train_per_centroid=256, niter=10)
g = built.gpu
xq = kb.queries(spec, nq, torch.device(dev), seed=44)
_, keys = g.coarse_search_device(xq, nprobe)
keys = keys.cpu().numpy()
print(f"coarse quantizer trained ({built.timings}); probes of {nq} queries: {keys.shape}", flush=True)

pairs = nq * nprobe
per_list = np.bincount(keys.ravel(), minlength=nlist)
units_today = int(((per_list + 15) // 16).sum())
print(f"today: (list, <= 16 queries) units = {units_today}, slots used {pairs / (units_today * 16):.3f} of 16 x units")


def union_cost(order, label):
    tot = 0
    sizes = []
    for s in range(0, nq, 16):
        u = np.unique(keys[order[s:s + 16]].ravel()).size
        sizes.append(u)
        tot += u
    sizes = np.asarray(sizes)
    print(f"{label}: groups {len(sizes)}, union of probed lists per group: mean {sizes.mean():.0f} (min {sizes.min()}, "
          f"max {sizes.max()}) of at most {16 * nprobe}; (list, group) visits {tot} = {tot / units_today:.2f} x today's units; "
          f"query slots doing work {pairs / (tot * 16):.3f}", flush=True)


union_cost(np.argsort(keys[:, 0], kind="stable"), "sorted by the closest list (the verdict's grouping)")
union_cost(np.lexsort((keys[:, 1], keys[:, 0])), "sorted by (closest, second closest) list")
# greedy: seed with the first free query, add the 15 free queries sharing most probes with it (exact, nq^2 / 16 set sizes)
member = np.zeros((nq, nlist), dtype=np.uint8)
member[np.arange(nq)[:, None], keys] = 1
mt = torch.from_numpy(member).to(dev).half()
free = torch.ones(nq, dtype=torch.bool, device=dev)
order = []
for _ in range(nq // 16):
    seed = int(torch.nonzero(free)[0])
    ov = (mt @ mt[seed]).float()
    ov[~free] = -1
    ov[seed] = 1e9
    pick = torch.topk(ov, 16).indices
    free[pick] = False
    order.extend(pick.cpu().tolist())
union_cost(np.asarray(order), "greedy by probe-set overlap with a seed query")
