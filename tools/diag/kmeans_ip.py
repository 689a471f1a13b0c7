import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
from conftest import gen_data
from oracle import binding as ob
from knowhere_amd.index import kmeans_device
from knowhere_amd import GpuIndex
port = ob.Port()
xb = gen_data(4000, 128, 42)
xt = torch.from_numpy(xb).cuda()
for k in (16, 24):
    for it in (1, 2, 3, 10):
        a = kmeans_device(ob.IP, xt, k, niter=it, spherical=True).cpu().numpy()
        b = port.kmeans(ob.IP, xb, k, niter=it, spherical=True)
        print("k", k, "niter", it, "rows differing", int((a != b).any(1).sum()), flush=True)
# assignment check on the centroids after one iteration (identical on both sides)
for k in (16, 24):
    cen = port.kmeans(ob.IP, xb, k, niter=1, spherical=True)
    g = GpuIndex(ob.IVF_FLAT, ob.IP, 128, nlist=k)
    g.set_coarse(cen)
    dis, keys = g.coarse_search_device(xt, 1)
    torch.cuda.synchronize()
    keys = keys.cpu().numpy(); dis = dis.cpu().numpy()
    ao = port.assign(ob.IP, cen, xb)
    bad = np.nonzero(keys[:, 0] != ao)[0]
    print("k", k, "assign mismatches", len(bad), bad[:5], flush=True)
    for i in bad[:3]:
        d = np.array([port.fvec_inner_product(xb[i], c) for c in cen], np.float32)
        o = np.argsort(-d, kind="stable")[:3]
        print("  row", i, "gpu", keys[i, 0], dis[i, 0], "oracle", ao[i], "top", o, d[o], d[keys[i, 0]])
