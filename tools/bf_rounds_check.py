import os, sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import gen_data
from knowhere_amd import GpuIndex
nb, d, nq = 300_000, 128, 5000
xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
g1 = GpuIndex(0, 0, d); g1.add_vectors(xb)
os.environ["KNHIP_BF"] = "exact"
g0 = GpuIndex(0, 0, d); g0.add_vectors(xb)
del os.environ["KNHIP_BF"]
g1.profile_enable(True)
for k in (10,):
    D1, I1 = g1.search(xq, k); print("form", g1.profile_get()["pq_filter_form"])
    D0, I0 = g0.search(xq, k)
    bad = np.flatnonzero((I0 != I1).any(1))
    print("k", k, "queries differing", len(bad), bad[:10], "dist equal", np.array_equal(D0.view(np.uint32), D1.view(np.uint32)))
    if len(bad):
        q = bad[0]; print(I0[q], I1[q]); print(D0[q], D1[q])
