"""a10 on the device (-m gpu): every entry of the src/simd hook table (reference src/simd/hook.h:33-123) through the C
ABI, against the known answers the REFERENCE's own scalar definitions produced (tests/golden/simd/table.npz, generator
tests/golden/make_simd_golden.py) and against the oracle restatement on other seeds.  Bar: bit-equal."""
import ctypes as C
import os

import numpy as np
import pytest

import simd_cases as sc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "simd", "table.npz")


class GpuTable:
    """the _SimdTable call shapes (oracle/binding.py) served by libknhip.so on cuda:0"""

    def __init__(self):
        import torch
        from knowhere_amd import _lib
        assert torch.cuda.is_available(), "-m gpu tests need a GPU"
        self.t, self._lib, self.L = torch, _lib, _lib.load()
        self._live = []

    def dev(self, a):
        a = np.ascontiguousarray(a)
        if a.dtype == np.uint16:  # bf16 bit patterns travel as int16 storage
            a = a.view(np.int16)
        t = self.t.from_numpy(a).cuda()
        self._live.append(t)  # inputs must outlive the asynchronous launch: released in host()
        return t

    @staticmethod
    def p(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    def out(self, n, dtype=None):
        return self.t.empty(max(n, 1), dtype=dtype or self.t.float32, device="cuda")

    def host(self, t, n):
        self.t.cuda.synchronize()
        r = t[:n].cpu().numpy()
        if len(self._live) > 64:
            self._live.clear()
        return r

    def simd_scalar(self, name, x, y=None):
        x = np.ascontiguousarray(x, np.float32)
        o = self.out(1)
        if name == "fvec_norm_L2sqr":
            self._lib.check(self.L.knhip_fvec_norms_L2sqr_ref(self.p(o), self.p(self.dev(x[None])), x.size, 1, None))
            return self.host(o, 1)[0]
        fn = {"fvec_inner_product": self.L.knhip_fvec_inner_products_ny, "fvec_L2sqr": self.L.knhip_fvec_L2sqr_ny,
              "fvec_L1": self.L.knhip_fvec_L1_ny, "fvec_Linf": self.L.knhip_fvec_Linf_ny}[name]
        yt = self.dev(np.ascontiguousarray(y, np.float32)[None])
        self._lib.check(fn(self.p(o), self.p(self.dev(x)), self.p(yt), x.size, 1, None))
        return self.host(o, 1)[0]

    def simd_ny(self, name, x, y):
        ny, d = y.shape
        o = self.out(ny)
        fn = self.L.knhip_fvec_L2sqr_ny if name == "fvec_L2sqr_ny" else self.L.knhip_fvec_inner_products_ny
        self._lib.check(fn(self.p(o), self.p(self.dev(x)), self.p(self.dev(y)), d, ny, None))
        return self.host(o, ny)

    def simd_ny_transposed(self, x, yt, y_sqlen, ny, nearest=False):
        d, d_offset = yt.shape
        o = self.out(ny)
        xd, yd, sd = self.dev(x), self.dev(yt), self.dev(y_sqlen)
        if nearest:
            idx = self.out(1, self.t.int64)
            self._lib.check(self.L.knhip_fvec_L2sqr_ny_nearest_y_transposed(self.p(o), self.p(xd), self.p(yd), self.p(sd),
                                                                           d, d_offset, ny, self.p(idx), None))
            return int(self.host(idx, 1)[0]), self.host(o, ny)
        self._lib.check(self.L.knhip_fvec_L2sqr_ny_transposed(self.p(o), self.p(xd), self.p(yd), self.p(sd), d, d_offset,
                                                              ny, None))
        return self.host(o, ny)

    def simd_ny_nearest(self, x, y):
        ny, d = y.shape
        o, idx = self.out(ny), self.out(1, self.t.int64)
        self._lib.check(self.L.knhip_fvec_L2sqr_ny_nearest(self.p(o), self.p(self.dev(x)), self.p(self.dev(y)), d, ny,
                                                           self.p(idx), None))
        return int(self.host(idx, 1)[0]), self.host(o, ny)

    def simd_madd(self, a, bf, b, argmin=False):
        c = self.out(a.size)
        ad, bd = self.dev(a), self.dev(b)
        if argmin:
            idx = self.out(1, self.t.int64)
            self._lib.check(self.L.knhip_fvec_madd_and_argmin(a.size, self.p(ad), bf, self.p(bd), self.p(c), self.p(idx),
                                                              None))
            return int(self.host(idx, 1)[0]), self.host(c, a.size)
        self._lib.check(self.L.knhip_fvec_madd(a.size, self.p(ad), bf, self.p(bd), self.p(c), None))
        return self.host(c, a.size)

    _DT = {np.dtype(np.float16): 0, np.dtype(np.uint16): 1, np.dtype(np.int8): 2}

    def simd_batch_4(self, is_l2, x, ys):
        o = self.out(4)
        rows = [self.dev(r) for r in ys]
        xd = self.dev(x)
        metric = 0 if is_l2 else 1
        if x.dtype == np.float32:
            self._lib.check(self.L.knhip_fvec_batch_4(metric, self.p(xd), *[self.p(r) for r in rows], x.size, self.p(o),
                                                      None))
        else:
            self._lib.check(self.L.knhip_typed_vec_batch_4(self._DT[x.dtype], metric, self.p(xd),
                                                           *[self.p(r) for r in rows], x.size, self.p(o), None))
        return self.host(o, 4)

    def simd_typed(self, op, x, y=None):
        o = self.out(1)
        dt = self._DT[np.ascontiguousarray(x).dtype]
        if op == 2:
            self._lib.check(self.L.knhip_typed_vec_ny(dt, 2, self.p(o), None, self.p(self.dev(x[None])), x.size, 1, None))
        else:
            self._lib.check(self.L.knhip_typed_vec_ny(dt, op, self.p(o), self.p(self.dev(x)), self.p(self.dev(y[None])),
                                                      x.size, 1, None))
        return self.host(o, 1)[0]

    def simd_ivec(self, is_l2, x, y):
        o = self.out(1, self.t.int32)
        self._lib.check(self.L.knhip_ivec_ny(0 if is_l2 else 1, self.p(o), self.p(self.dev(x)), self.p(self.dev(y[None])),
                                             x.size, 1, None))
        return int(self.host(o, 1)[0])


@pytest.fixture(scope="module")
def gpu_table():
    return GpuTable()


@pytest.mark.parametrize("d", sc.DIMS)
def test_gpu_equals_reference_known_answers(gpu_table, d):
    z = np.load(GOLD)
    got = sc.evaluate(gpu_table, d)
    for name, v in got.items():
        want = z[f"d{d}/{name}"]
        assert sc.same(np.asarray(v, want.dtype), want), f"{name} d={d}: {np.asarray(v).ravel()[:4]} vs {want.ravel()[:4]}"


@pytest.mark.parametrize("d", (5, 96, 512))
def test_gpu_equals_oracle_other_seeds(gpu_table, port, d):
    a, b = sc.evaluate(gpu_table, d, seed=3), sc.evaluate(port, d, seed=3)
    for name in b:
        assert sc.same(np.asarray(a[name], b[name].dtype), b[name]), f"{name} d={d}"


def test_row_entries_at_scale(gpu_table, port):
    """many rows (several workgroups, ragged tail) incl. the typed operands and the grid-wide argmin"""
    T = gpu_table
    r = np.random.default_rng(5)
    ny, d = 20011, 96
    x = (r.random(d, dtype=np.float32) * 2 - 1).astype(np.float32)
    y = (r.random((ny, d), dtype=np.float32) * 2 - 1).astype(np.float32)
    y[15000] = y[777]
    i, dis = T.simd_ny_nearest(y[15000], y)
    assert i == 777 and dis.tobytes() == port.fvec_L2sqr_ny(y[15000], y).tobytes()
    for dt, mk in ((np.float16, lambda a: a.astype(np.float16)), (np.uint16, sc.bf16_bits),
                   (np.int8, lambda a: (a * 127).astype(np.int8))):
        xx, yy = mk(x), mk(y)
        o = T.out(ny)
        for op in (0, 1, 2):
            T._lib.check(T.L.knhip_typed_vec_ny(T._DT[np.dtype(dt)], op, T.p(o), T.p(T.dev(xx)), T.p(T.dev(yy)), d, ny,
                                                None))
            got = T.host(o, ny)
            for row in (0, 1, 63, 64, 127, 128, 9999, ny - 1):
                want = port.simd_typed(op, xx, yy[row]) if op < 2 else port.simd_typed(2, yy[row])
                assert got[row] == want, (dt, op, row)
