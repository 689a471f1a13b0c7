"""tools/bench_prims.py -- GB/s of the src/simd hook-table kernels (prims.hip) on one GPU: every entry is a streaming read
of y, algorithmic bytes = ny * d * sizeof(T) (+ the output), bound by HBM (8 TB/s, MI355X_MICROARCH.md).
usage (GPU box):  python tools/bench_prims.py > gpurun_out/prims.log"""
import ctypes as C
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from knowhere_amd import _lib  # noqa: E402

L = _lib.load()
HBM = 8000.0
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def timed(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


rows = []
for d in (128, 768):
    ny = (1 << 31) // (d * 4)  # 2 GiB of fp32 rows
    y = torch.rand((ny, d), device="cuda") * 2 - 1
    x = torch.rand(d, device="cuda")
    out = torch.empty(ny, device="cuda")
    idx = torch.empty(1, dtype=torch.int64, device="cuda")
    ysq = torch.empty(ny, device="cuda")
    cases = [
        ("fvec_L2sqr_ny", 4, lambda: L.knhip_fvec_L2sqr_ny(p(out), p(x), p(y), d, ny, None)),
        ("fvec_inner_products_ny", 4, lambda: L.knhip_fvec_inner_products_ny(p(out), p(x), p(y), d, ny, None)),
        ("fvec_L1_ny", 4, lambda: L.knhip_fvec_L1_ny(p(out), p(x), p(y), d, ny, None)),
        ("fvec_norms_L2sqr", 4, lambda: L.knhip_fvec_norms_L2sqr(p(out), p(y), d, ny, None)),
        ("fvec_norms_L2sqr_ref", 4, lambda: L.knhip_fvec_norms_L2sqr_ref(p(out), p(y), d, ny, None)),
        ("fvec_L2sqr_ny_nearest", 4, lambda: L.knhip_fvec_L2sqr_ny_nearest(p(out), p(x), p(y), d, ny, p(idx), None)),
        # the same buffer read as [d][d_offset = ny]
        ("fvec_L2sqr_ny_transposed", 4, lambda: L.knhip_fvec_L2sqr_ny_transposed(p(out), p(x), p(y), p(ysq), d, ny, ny, None)),
    ]
    yh = y.to(torch.float16)
    xh = x.to(torch.float16)
    yb = y.to(torch.bfloat16)
    xb = x.to(torch.bfloat16)
    yi = (y * 127).to(torch.int8)
    xi = (x * 127).to(torch.int8)
    cases += [
        ("fp16_vec_L2sqr (ny)", 2, lambda: L.knhip_typed_vec_ny(0, 0, p(out), p(xh), p(yh), d, ny, None)),
        ("bf16_vec_inner_product (ny)", 2, lambda: L.knhip_typed_vec_ny(1, 1, p(out), p(xb), p(yb), d, ny, None)),
        ("int8_vec_L2sqr (ny)", 1, lambda: L.knhip_typed_vec_ny(2, 0, p(out), p(xi), p(yi), d, ny, None)),
    ]
    for name, esz, fn in cases:
        ms = timed(fn)
        gb = (ny * d * esz + ny * 4) / 1e9
        rows.append({"entry": name, "d": d, "ny": ny, "ms": round(ms, 3), "GBps": round(gb / ms * 1e3, 1),
                     "hbm_frac": round(gb / ms * 1e3 / HBM, 3)})
        print(rows[-1], flush=True)
    n = ny * d
    a, b, c = y.view(-1), torch.rand(n, device="cuda"), torch.empty(n, device="cuda")
    for name, fn in (("fvec_madd", lambda: L.knhip_fvec_madd(n, p(a), -2.0, p(b), p(c), None)),
                     ("fvec_madd_and_argmin", lambda: L.knhip_fvec_madd_and_argmin(n, p(a), -2.0, p(b), p(c), p(idx), None))):
        ms = timed(fn)
        gb = n * 12 / 1e9
        rows.append({"entry": name, "d": d, "ny": n, "ms": round(ms, 3), "GBps": round(gb / ms * 1e3, 1),
                     "hbm_frac": round(gb / ms * 1e3 / HBM, 3)})
        print(rows[-1], flush=True)
    del y, yh, yb, yi, a, b, c
print(json.dumps(rows))
