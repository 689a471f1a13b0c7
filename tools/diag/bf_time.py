"""BRUTE_FORCE batch time at 1M x 128, nq 10k, k 10 (device-resident queries), and its bits against the exact row scan on a slice"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
from conftest import gen_data
from knowhere_amd import GpuIndex
nb, d, nq, k = 1_000_000, 128, 10000, 10
xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
g = GpuIndex(0, 0, d); g.add_vectors(xb)
xq_t = torch.from_numpy(xq).cuda()
for it in range(5):
    torch.cuda.synchronize(); t0 = time.time()
    D, I = g.search_device(xq_t, k)
    torch.cuda.synchronize(); t1 = time.time()
    print("BF 1M x 128 nq 10k ms", round((t1 - t0) * 1e3, 3), flush=True)
os.environ["KNHIP_BF"] = "exact"
g0 = GpuIndex(0, 0, d); g0.add_vectors(xb)
del os.environ["KNHIP_BF"]
D0, I0 = g0.search_device(xq_t[:500], k)
print("bits equal to the row scan (500 queries):", bool((D0.view(torch.int32) == D[:500].view(torch.int32)).all().item() and (I0 == I[:500]).all().item()))
