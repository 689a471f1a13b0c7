"""CPU check of the error bound the half-precision ADC prefilter relies on (knowhere_amd/csrc/pq_filter.hip).

The prefilter never decides a result: it only has to let every row through whose EXACT distance (the reference's fp32
sum in m order) is within the query's bound, which holds as long as |approx - exact| <= eps.  Here the kernel's
arithmetic is replayed in numpy: per-query table scaled to integers of at most 2048 in all, 32 (exact) half additions in
the rotated order a lane walks (any start phase), the per-vector term-2 sum in fp32, the final fp32 combination; exact is
the reference's sequence; eps is the kernel's formula.  The bound must hold with room to spare on random and on
adversarial inputs (one-signed tables whose partial sums reach A_q, tiny and huge value scales, entries far below the
half range of the scaled table)."""
import numpy as np
import pytest

f32 = np.float32
f16 = np.float16
U = f32(2.0 ** -24)
UH = f32(2.0 ** -11)
M, KSUB, DSUB = 32, 256, 4


def _ip_chain(x, y):
    """fvec_inner_product scalar form: products added in dimension order from 0, one rounding per operation"""
    t = f32(0)
    for a, b in zip(x, y):
        t = f32(t + f32(a * b))
    return t


def _tables(q, cb, is_l2):
    T = np.zeros((M, KSUB), f32)
    for m in range(M):
        for c in range(KSUB):
            T[m, c] = _ip_chain(q[m * DSUB:(m + 1) * DSUB], cb[m, c])
    return (f32(-2.0) * T).astype(f32) if is_l2 else T


def _query_prep(Qf, pabs_max):
    """pqf_query_table_kernel: A = sum_m max_c |Qf|, sc = 2032 / A, integer table rint(Qf sc), eps_base"""
    A = f32(0)
    for m in range(M):
        A = f32(A + np.abs(Qf[m]).max())
    with np.errstate(over="ignore", divide="ignore"):
        sc = f32(f32(2032.0) / A) if A > 0 else f32(1.0)
    if not np.isfinite(sc):
        return A, f32(1.0), np.zeros_like(Qf, dtype=f16), f32(np.inf)  # (no bound: the exact kernels)
    Qh = np.rint((Qf * sc).astype(f32)).astype(f16)
    eps_base = f32(f32(16.5) / sc + f32(64.0) * U * f32(pabs_max + A))
    return A, sc, Qh, eps_base


def _case(rng, scale, mode):
    q = (rng.standard_normal(M * DSUB) * scale).astype(f32)
    cb = (rng.standard_normal((M, KSUB, DSUB)) * scale).astype(f32)
    if mode == "one_signed":
        q, cb = np.abs(q), np.abs(cb)
    if mode == "mixed_magnitudes":  # a few sub-quantizers dominate; the rest sit far below the scaled half range
        cb[4:] *= f32(1e-6)
    return q, cb


@pytest.mark.parametrize("is_l2", [True, False], ids=["l2", "ip"])
@pytest.mark.parametrize("scale", [1e-18, 1e-3, 1.0, 300.0])  # (1e-18: products of subnormal magnitude -> no bound)
@pytest.mark.parametrize("mode", ["random", "one_signed", "mixed_magnitudes"])
def test_half_adc_bound_holds_with_margin(is_l2, scale, mode):
    rng = np.random.default_rng(int(scale * 7) % 1000 + len(mode) + (3 if is_l2 else 0))
    q, cb = _case(rng, scale, mode)
    Qf = _tables(q, cb, is_l2)
    # term 2 of one list (precomputed table row): ||cb||^2 + 2 <c, cb>, any fp32 values serve -- both sides read them
    cen = (rng.standard_normal(M * DSUB) * scale).astype(f32)
    P = np.zeros((M, KSUB), f32)
    if is_l2:
        for m in range(M):
            P[m] = (cb[m] * cb[m]).sum(1) + f32(2) * (cb[m] * cen[m * DSUB:(m + 1) * DSUB]).sum(1)
    codes = rng.integers(0, KSUB, (48, M))
    if mode == "one_signed":
        codes[0] = np.abs(Qf).argmax(1)  # every entry at its column maximum: partial sums reach A
    ar = np.arange(M)
    pabs_max = f32(np.abs(P[ar, codes]).astype(f32).sum(1, dtype=f32).max()) if is_l2 else f32(0)
    A, sc, Qh, eps_base = _query_prep(Qf, pabs_max)
    isc = f32(1.0) / sc
    dis0 = f32(abs(rng.standard_normal()) * scale * scale * 40)
    tau = dis0  # (enters eps only through the roundings of the threshold)
    eps = f32(eps_base + f32(64.0) * U * f32(abs(dis0) + abs(tau)))
    assert np.isfinite(Qh.astype(f32)).all()
    if not np.isfinite(eps_base):
        return  # (a table of subnormal magnitude: the kernel hands the query to the exact path)
    worst = 0.0
    for row in codes:
        # exact: LUT entry = term2 + (-2 <q_m, cb>) rounded once (fvec_madd), summed from 0 in m order, dis0 last
        acc = f32(0)
        for m in range(M):
            lut = f32(P[m, row[m]] + Qf[m, row[m]]) if is_l2 else Qf[m, row[m]]
            acc = f32(acc + lut)
        exact = f32(dis0 + acc)
        # approx: half additions in a lane's rotated order, start phase = any of the 16
        for ph in (0, 5, 15):
            h = f16(0)
            for t in range(M):
                m = (t + ph) & 31
                h = f16(h + Qh[m, row[m]])
            assert np.isfinite(f32(h))
            ps = f32(0)
            for m in range(M):
                ps = f32(ps + P[m, row[m]])
            approx = f32(f32(h) * isc + f32(dis0 + ps)) if is_l2 else f32(f32(h) * isc + dis0)
            err = abs(float(approx) - float(exact))
            assert err <= float(eps), (err, float(eps))
            worst = max(worst, err / float(eps))
    assert worst < 0.75, f"the bound holds but with little room: {worst:.3f} of eps"


def test_integer_table_makes_the_half_additions_exact():
    """the per-m maxima of the rounded entries sum to at most 2048, every entry is an integer: every partial sum of a
    vector's 32 entries is an integer of magnitude <= 2048, which half precision represents exactly -- the half sum
    equals the integer sum for any order of the additions"""
    rng = np.random.default_rng(3)
    for scale in (1e-15, 1e-3, 1.0, 1e6, 1e15):
        for mode in ("one_signed", "random"):
            q, cb = _case(rng, scale, mode)
            Qf = _tables(q, cb, True)
            A, sc, Qh, eps = _query_prep(Qf, f32(0))
            assert np.isfinite(eps)
            Qi = Qh.astype(np.float64)
            assert np.array_equal(Qi, np.rint(Qi)) and np.abs(Qi).max(1).sum() <= 2048
            codes = rng.integers(0, KSUB, (20, M))
            codes[0] = np.abs(Qf).argmax(1)
            for row in codes:
                exact_int = int(Qi[np.arange(M), row].sum())
                for ph in (0, 7):
                    h = f16(0)
                    for t in range(M):
                        m = (t + ph) & 31
                        h = f16(h + Qh[m, row[m]])
                    assert float(h) == exact_int


def test_token_rotation_covers_every_subquantizer_once():
    """a lane's 32 steps of a window visit each m exactly once, and the 16 lanes an LDS gather is serviced together for
    sit on 16 different bank quads (m mod 16) at every step -- the phase map of kernels.h::pq_stream_phase"""
    def phase(lane):
        l = lane & 31
        return l if l < 4 else l + 4 if l < 12 else l - 8 if l < 16 else l - 16 if l < 20 else l - 12 if l < 28 else l - 24
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
              [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    for lane in range(64):
        assert sorted((t + phase(lane)) & 31 for t in range(32)) == list(range(32))
    for t in range(32):
        for g in groups:
            for base in (0, 32):
                quads = {((t + phase(base + l)) & 31) & 15 for l in g}
                assert len(quads) == 16
        for g0 in range(0, 64, 16):  # (also conflict-free for contiguous sixteenths)
            assert len({((t + phase(l)) & 31) & 15 for l in range(g0, g0 + 16)}) == 16
