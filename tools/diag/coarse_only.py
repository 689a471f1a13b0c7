"""coarse stage alone at the C3 / C5 shapes (random centroids and queries): for rocprofv3 --kernel-trace --stats"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from knowhere_amd import GpuIndex
from knowhere_amd.index import IVF_FLAT
shape = sys.argv[1] if len(sys.argv) > 1 else "C3"
nlist, d, nq, nprobe, metric = (16384, 128, 10000, 128, 0) if shape == "C3" else (65536, 768, 10000, 256, 1)
g = torch.Generator(device="cuda").manual_seed(1)
cen = torch.randn((nlist, d), device="cuda", generator=g)
if metric == 1:
    cen = cen / cen.norm(dim=1, keepdim=True)
xq = torch.randn((nq, d), device="cuda", generator=g)
ix = GpuIndex(IVF_FLAT, metric, d, nlist=nlist)
ix.set_coarse_device(cen.contiguous())
for it in range(6):
    torch.cuda.synchronize(); t0 = time.time()
    D, I = ix.coarse_search_device(xq, nprobe)
    torch.cuda.synchronize(); t1 = time.time()
    print(shape, "coarse ms", round((t1 - t0) * 1e3, 3), flush=True)
