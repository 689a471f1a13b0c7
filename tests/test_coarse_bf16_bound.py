"""CPU: the error bound behind the coarse quantizer's bf16 prefilter (knowhere_amd/csrc/coarse_gemm.hip, coarse_bf16_kernel).

Operands are split x = hi + lo + r into two bf16 terms (round to nearest even, as v_cvt_pk_bf16_f32), the product is
hi hi + hi lo + lo hi accumulated in fp32.  The certificate of coarse_rerank_kernel widens its threshold by
eps = (8 d 2^-24 + 2^-14) * (||q||^2 + max ||c||^2) (L2) / * ||q|| max ||c|| (IP); this replays the arithmetic in numpy --
every product rounded into the accumulator on its own, the least favourable order the hardware could take -- on random
and on adversarial inputs (every element's two roundings pushed the same way) and requires the observed error to stay
under the analytic bound (3 * 2^-16 + 3 d 2^-24) ||q|| ||c||, and that bound under the kernel's eps."""
import numpy as np
import pytest


def bf16_rn(x):
    """float32 -> the nearest bf16 (ties to even), returned as float32"""
    b = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((b + 0x7FFF + ((b >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def split(x):
    hi = bf16_rn(x)
    lo = bf16_rn((x - hi).astype(np.float32))
    return hi, lo


def approx_dot(q, c):
    qh, ql = split(q)
    ch, cl = split(c)
    acc = np.float32(0)
    for a, b in ((qh, ch), (qh, cl), (ql, ch)):
        for i in range(q.shape[0]):
            acc = np.float32(acc + np.float32(a[i] * b[i]))  # (bf16 x bf16 is exact in fp32)
    return acc


def adversarial(d, rng):
    """elements just above a bf16 rounding boundary twice over: hi and lo both round down by almost half a unit"""
    e = rng.integers(-3, 4, d)
    m = 1.0 + 2.0 ** -8 * (1 - 2.0 ** -9) + 2.0 ** -16 * (1 - 2.0 ** -7)
    return (m * 2.0 ** e).astype(np.float32)


@pytest.mark.parametrize("d", [8, 32, 100, 128, 768])
def test_split_bf16_dot_error_is_inside_the_certificates_eps(d):
    rng = np.random.default_rng(d)
    worst = 0.0
    cases = []
    for _ in range(40):
        cases.append((rng.standard_normal(d).astype(np.float32) * np.float32(10.0 ** rng.integers(-2, 3)),
                      rng.standard_normal(d).astype(np.float32) * np.float32(10.0 ** rng.integers(-2, 3))))
    for _ in range(10):
        cases.append((adversarial(d, rng), adversarial(d, rng)))
        cases.append((adversarial(d, rng), -adversarial(d, rng)))
    for q, c in cases:
        exact = float(np.dot(q.astype(np.float64), c.astype(np.float64)))
        err = abs(float(approx_dot(q, c)) - exact)
        nq, nc = float(np.linalg.norm(q.astype(np.float64))), float(np.linalg.norm(c.astype(np.float64)))
        bound = (3 * 2.0 ** -16 + 3 * d * 2.0 ** -24) * nq * nc
        assert err <= bound, (err, bound)
        worst = max(worst, err / bound)
        eps_rel = 8 * d * 2.0 ** -24 + 2.0 ** -14  # launch_coarse_rerank
        assert 2 * bound <= eps_rel * (nq * nq + nc * nc)  # L2: the distance carries twice the dot's error
        assert bound <= eps_rel * nq * nc                   # IP
    assert worst > 0.02  # (the adversarial inputs do come near: the bound is not vacuous)


def test_bf16_rounding_matches_the_definition():
    x = np.array([1.0, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -8 + 2.0 ** -20, 1.0 + 3 * 2.0 ** -8, -3.1415927, 65504.0, 1e-30],
                 np.float32)
    hi = bf16_rn(x)
    assert hi[0] == 1.0 and hi[1] == 1.0 and hi[2] == np.float32(1.0 + 2.0 ** -7)  # tie to even, then just above the tie
    assert hi[3] == np.float32(1.0 + 2.0 ** -6)  # tie to even upwards
    assert np.all(np.abs(x - hi) <= 2.0 ** -8 * np.abs(x))
    _, lo = split(x)
    assert np.all(np.abs(x - hi - lo) <= 2.0 ** -16 * np.abs(x))
