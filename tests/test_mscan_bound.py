"""CPU check of the error bounds the MFMA prefilter relies on (knowhere_amd/csrc/mfma_scan.hip).

The prefilter never decides a result: it only has to let every row through whose EXACT (reference-order fp32) distance
is within the query's bound, which holds as long as |approx - exact| <= eps.  Here the approximate arithmetic of the
kernels is replayed in numpy with the LEAST favourable rounding the hardware could use (every product added to the
fp32 accumulator one at a time, each addition rounded -- the matrix cores round less often), the exact distance is the
reference's sequential fp32 sum, and eps is the kernels' formula.  The margin is reported: eps must hold with room to
spare on random and on adversarial inputs (one-signed queries, saturated codes, tiny and huge value scales)."""
import numpy as np
import pytest

f32 = np.float32
U = f32(5.9604645e-8)


def _seq_sum(terms):
    """sequential fp32 accumulation of fp32 terms (one rounding per addition)"""
    acc = f32(0)
    for t in terms:
        acc = f32(acc + f32(t))
    return acc


def _sq8_case(d, scale, rng, mode):
    xq = (rng.standard_normal(d) * scale).astype(f32)
    xb = (rng.standard_normal((200, d)) * scale).astype(f32)
    if mode == "one_signed":
        xq, xb = np.abs(xq), np.abs(xb)
    vmin = xb.min(0).astype(f32)
    vdiff = (xb.max(0) - xb.min(0)).astype(f32)
    vdiff[vdiff == 0] = f32(1)
    codes = np.clip(np.floor((xb - vmin) / vdiff * f32(255)), 0, 255).astype(np.int64)
    if mode == "saturated":
        codes[:] = 255
    return xq, vmin, vdiff, codes


@pytest.mark.parametrize("d", [32, 128, 768])
@pytest.mark.parametrize("scale", [1e-3, 1.0, 100.0, 1e4])
@pytest.mark.parametrize("mode", ["random", "one_signed", "saturated"])
def test_sq8_ip_bound_holds_with_margin(d, scale, mode):
    rng = np.random.default_rng(d * 7 + int(np.log10(scale) * 3) + len(mode))
    q, vmin, vdiff, codes = _sq8_case(d, scale, rng, mode)
    dis0 = f32(rng.standard_normal() * scale * scale * d)
    tab = ((np.arange(256, dtype=f32) + f32(0.5)) / f32(255)).astype(f32)  # Codec8bit::decode_component
    inv255 = f32(1.0) / f32(255.0)
    # --- the kernel's query operand: y' scaled by a power of two, split into two halves -------------------------------
    yp_un = (q * vdiff * inv255).astype(f32)
    mx = np.abs(yp_un).max()
    ex = 0 if not (mx > 0) else int(np.clip(9 - int(np.floor(np.log2(mx))), -60, 60))
    sc = f32(2.0) ** ex
    yp = (q * vdiff * inv255 * sc).astype(f32)
    hi = yp.astype(np.float16)
    lo = (yp - hi.astype(f32)).astype(np.float16)
    A = f32(np.sum((q * (vmin + f32(0.5) * vdiff * inv255)).astype(f32), dtype=f32))
    W = f32(np.sum(np.abs(q) * (np.abs(vmin) + np.abs(vdiff)), dtype=f32))
    Yp = f32(np.sum(np.abs(yp_un), dtype=f32))
    HL = f32(np.sum(hi.astype(f32) + lo.astype(f32), dtype=f32))
    off = f32(1024.0) * HL
    e_mfma = (f32(2 * d) + f32(64)) * U * f32(1279) * Yp
    e_misc = f32(32) * U * (abs(A) + f32(1024) * Yp + abs(dis0))
    eps = f32(2) * (e_mfma + e_misc + (f32(d) + f32(8)) * U * W)
    worst = 0.0
    for row in codes[:40]:
        # approx: every product of S = sum (hi + lo)(1024 + c) added to the fp32 accumulator on its own
        a = (f32(1024) + row.astype(f32)).astype(f32)
        S = _seq_sum(np.concatenate([(hi.astype(f32) * a), (lo.astype(f32) * a)]))
        approx = f32(dis0 + A) + f32(f32(S - off) / sc)
        # exact: the reference's sequence (decode, multiply, add; one rounding each), accu0 added last
        x = (vmin + (tab[row] * vdiff).astype(f32)).astype(f32)
        exact = f32(dis0 + _seq_sum((q * x).astype(f32)))
        err = abs(float(approx) - float(exact))
        assert err <= float(eps), (err, float(eps))
        worst = max(worst, err / float(eps))
    assert worst < 0.5, f"the bound holds but with little room: {worst:.3f} of eps"


@pytest.mark.parametrize("d", [30, 128, 600])
@pytest.mark.parametrize("scale", [1e-2, 1.0, 100.0])
def test_flat_l2_bound_holds_with_margin(d, scale):
    rng = np.random.default_rng(d + int(scale * 10))
    q = (rng.random(d) * scale).astype(f32)
    xb = (rng.random((60, d)) * scale).astype(f32)
    qn = f32(np.sum(q * q, dtype=f32))
    xn = np.array([f32(np.sum(r * r, dtype=f32)) for r in xb], f32)
    eps = f32(16) * f32(d) * U * (qn + xn.max())
    worst = 0.0
    for r, n in zip(xb, xn):
        dot = _seq_sum((q * r).astype(f32))          # one rounded addition per product (the MFMA rounds less often)
        acc = f32(dot - f32(0.5) * n)                 # the accumulator starts at -||x||^2 / 2
        approx = f32(qn - f32(2) * acc)
        t = (q - r).astype(f32)
        exact = _seq_sum((t * t).astype(f32))         # src/simd/distances_ref.cc:30-37
        err = abs(float(approx) - float(exact))
        assert err <= float(eps), (err, float(eps))
        worst = max(worst, err / float(eps))
    assert worst < 0.5, f"the bound holds but with little room: {worst:.3f} of eps"
