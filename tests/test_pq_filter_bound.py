"""CPU check of the error bound the matrix-core ADC prefilter relies on (knowhere_amd/csrc/pq_filter.hip).

The prefilter never decides a result: it only has to let every row through whose EXACT distance (the reference's fp32
sum in m order) is within the query's bound, which holds as long as |approx - exact| <= eps.  Here the kernel's
arithmetic is replayed in numpy: per-query table scaled by a power of two and rounded to half precision, 32 products
with 1.0 accumulated in fp32 under the least favourable treatment the matrix core could give them (every addition
rounded on its own, in the lane's k-block order and in reverse), the per-vector term-2 sum in fp32, the final fp32
combination; exact is the reference's sequence; eps is the kernel's formula.  The bound must hold with room to spare on
random and on adversarial inputs (one-signed tables whose partial sums reach A_q, tiny and huge value scales, entries
far below the half range of the scaled table)."""
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
    """pqf_query_table_kernel: A = sum_m max_c |Qf|, sc = 2^(15 - e) with max |Qf| = f 2^e, table half(Qf sc), eps_base"""
    A = f32(0)
    gmax = f32(0)
    for m in range(M):
        a = np.abs(Qf[m]).max()
        A = f32(A + a)
        gmax = max(gmax, a)
    if not np.isfinite(A):
        return A, f32(1.0), np.zeros_like(Qf, dtype=f16), f32(np.inf)  # (no bound: the exact kernels)
    sc = f32(1.0)
    if gmax > 0:
        _, e = np.frexp(gmax)
        sc = f32(np.ldexp(1.0, int(np.clip(15 - e, -126, 126))))
    isc = f32(1.0) / sc
    with np.errstate(over="ignore", under="ignore"):
        Qh = (Qf * sc).astype(f32).astype(f16)
    eps_base = f32(f32(UH * A) * f32(1.001) + f32(2.0 ** -20) * isc + f32(128.0) * U * f32(pabs_max + A))
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
        # approx: the 32 halves (exact products with 1.0) added in fp32, one rounding per addition, in several orders
        for order in (list(range(M)), list(range(M - 1, -1, -1)), [(5 * t + 3) % M for t in range(M)]):
            h = f32(0)
            for m in order:
                h = f32(h + f32(Qh[m, row[m]]))
            ps = f32(0)
            for m in range(M):
                ps = f32(ps + P[m, row[m]])
            # kernel: L2 fma(h, isc, ps) <= (tau + eps) - dis0;  pessimistic value fma(h, isc, (dis0 +- eps) + ps)
            approx = f32(f32(h * isc + ps) + dis0) if is_l2 else f32(f32(h * isc) + dis0)
            err = abs(float(approx) - float(exact))
            assert err <= float(eps), (err, float(eps))
            worst = max(worst, err / float(eps))
    assert worst < 0.75, f"the bound holds but with little room: {worst:.3f} of eps"


def test_scaled_half_table_keeps_eleven_bits():
    """the power-of-two scale puts the largest magnitude into [2^14, 2^15): no entry overflows half precision and every
    normal entry is within 2^-11 relative of the fp32 value (2^-25 absolute, scaled, for the subnormal ones)"""
    rng = np.random.default_rng(3)
    for scale in (1e-15, 1e-3, 1.0, 1e6, 1e15):
        for mode in ("one_signed", "random", "mixed_magnitudes"):
            q, cb = _case(rng, scale, mode)
            Qf = _tables(q, cb, True)
            A, sc, Qh, eps = _query_prep(Qf, f32(0))
            assert np.isfinite(eps)
            big = np.abs(Qh.astype(np.float64)).max()
            assert 2.0 ** 14 <= big <= 2.0 ** 15
            x = Qf.astype(np.float64) * float(sc)
            err = np.abs(Qh.astype(np.float64) - x)
            assert (err <= np.maximum(np.abs(x) * 2.0 ** -11, 2.0 ** -25)).all()


def test_lane_map_covers_every_subquantizer_once_without_bank_conflicts():
    """pq_filter.hip::pf_lane_vec / pf_lane_m: the two lanes (k blocks 2 v, 2 v + 1) that fetch for one vector of a group
    visit its 32 sub-quantizers exactly once over the 16 steps, and the 16 lanes an LDS gather is serviced together for
    sit on 16 different bank quads (m mod 16) at every step -- the phase map of kernels.h::pq_stream_phase"""
    def phase(lane):
        l = lane & 31
        return l if l < 4 else l + 4 if l < 12 else l - 8 if l < 16 else l - 16 if l < 20 else l - 12 if l < 28 else l - 24

    def lane_vec(L):
        return 16 * (L >> 5) + (L & 15)

    def lane_m(L, s):
        return 16 * ((L >> 4) & 1) + ((phase(L) + s) & 15)
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
              [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    seen = {}
    for L in range(64):
        for s_ in range(16):
            seen.setdefault(lane_vec(L), []).append(lane_m(L, s_))
    assert sorted(seen) == list(range(32))
    for v, ms in seen.items():
        assert sorted(ms) == list(range(32)), v
    for s_ in range(16):
        for g in groups:
            for base in (0, 32):
                assert len({lane_m(base + l, s_) & 15 for l in g}) == 16
        for g0 in range(0, 64, 16):  # (also conflict-free for contiguous sixteenths)
            assert len({lane_m(l, s_) & 15 for l in range(g0, g0 + 16)}) == 16
    # selector operand: accumulator row i of column n sums element i & 7 of the k blocks 2 (i >> 3), 2 (i >> 3) + 1 -- the
    # lanes that fetch for vector 16 (i >> 3) + n
    for i in range(16):
        for kb in range(4):
            on = (kb >> 1) == (i >> 3)
            for n in range(16):
                L = 16 * kb + n
                assert (lane_vec(L) == 16 * (i >> 3) + n) == on


# ---- integer form: the integer fast path is a superset of the fp32 test ------------------------------------------------
def _int_threshold(t, inv0, is_l2):
    """pqi_kernel, per-pair constants: T in units of s0 (the float arithmetic of the kernel, step by step)"""
    tt = f32(t) if is_l2 else f32(-f32(t))
    with np.errstate(all="ignore"):  # (inf - inf / inf * x on the pass-nothing thresholds: NaN, clamped below as fmaxf does)
        x = f32(np.floor(f32(f32(tt + f32(f32(2.384185791015625e-7) * f32(abs(tt)))) * inv0)) + f32(2.0))
    lim = f32(2.0 * 536870912.0)
    if np.isnan(x):
        x = -lim  # fmaxf(NaN, lo) = lo
    x = min(max(x, -lim), lim)
    return int(x)


@pytest.mark.parametrize("is_l2", [True, False], ids=["l2", "ip"])
def test_integer_fast_path_is_a_superset_of_the_fp32_test(is_l2):
    """pq_filter.hip, integer form: with s_q = w s0 (s0 a power of two, w an integer weight in the selector operand) and the
    accumulator started at -T, the per-group test `w S - T <= ceil(-ps / s0)` (IP: `-w S - T <= 0`) must pass every
    (S, ps, t) the fp32 test `fma(S, s_q, ps) <= t` (IP: `S s_q >= t`) passes -- else a candidate would be lost.  Random and
    edge inputs over twelve orders of magnitude of the step; also: T stays inside the clamp, and the fp32 value rebuilt
    in the slow path from the integer accumulator equals the fma of the first version bit for bit."""
    rng = np.random.default_rng(7)
    n_pass = 0
    for trial in range(60000):
        e0 = int(rng.integers(-40, 20))
        s0 = f32(2.0 ** e0)
        inv0 = f32(1.0) / s0
        w = int(rng.integers(1, 128))
        s_q = f32(f32(w) * s0)  # exact: w < 2^7
        S = int(rng.integers(-4064, 4065))  # sum of 32 int8 entries
        scale = float(s_q) * 4064.0
        kind = trial % 6
        ps = f32(rng.normal(0.0, scale)) if is_l2 and kind != 0 else f32(0.0)
        if is_l2 and kind == 1:
            ps = f32(rng.normal(0.0, scale * 1e3))  # term-2 sums far above the table part
        # thresholds around the decision boundary, exactly on it, and far away
        val = np.float32(np.float64(S) * np.float64(s_q) + np.float64(ps))
        t = f32(val + f32(rng.normal(0.0, 1.0)) * f32(scale * 1e-3)) if kind < 4 else (
            f32(val) if kind == 4 else f32(rng.normal(0.0, scale * 10)))
        if abs(float(ps)) * float(inv0) >= 2.0 ** 29:  # (the table kernel refuses such batches: eps = inf)
            continue
        # fp32 test of the kernel's slow path: one fused multiply-add, then the compare
        v = np.float32(np.float64(f32(S)) * np.float64(s_q) + np.float64(ps))  # fma: one rounding
        passes_f = (v <= t) if is_l2 else (np.float32(np.float64(f32(S)) * np.float64(s_q)) >= t)
        # integer test
        T = _int_threshold(t, inv0, is_l2)
        assert abs(T) <= 2 ** 30
        Y = w * S
        negp = int(np.ceil(f32(f32(-ps) * inv0))) if is_l2 else 0
        x = (Y - T) if is_l2 else (-Y - T)
        passes_i = x <= negp
        if passes_f:
            n_pass += 1
            assert passes_i, (trial, e0, w, S, float(ps), float(t), T, negp)
        # the slow path's rebuilt operand: y = x + T (L2) / -(x + T) (IP) is w S exactly, and fma(y, s0, ps) == fma(S, s_q, ps)
        y = (x + T) if is_l2 else -(x + T)
        assert y == Y
        v2 = np.float32(np.float64(f32(y)) * np.float64(s0) + np.float64(ps if is_l2 else f32(0)))
        v1 = np.float32(np.float64(f32(S)) * np.float64(s_q) + np.float64(ps if is_l2 else f32(0)))
        assert v1.view(np.uint32) == v2.view(np.uint32)
    assert n_pass > 5000  # (the boundary cases really were exercised)
    # pass-nothing and NaN thresholds: nothing may pass the integer test by accident of the conversion
    for t in (-np.inf, np.nan) if is_l2 else (np.inf, np.nan):
        T = _int_threshold(f32(t), f32(1.0), is_l2)
        assert T == -2 ** 30


def _int8_prep(Qf, s0, pabs_max):
    """pqi_query_stats_kernel + pqi_query_table_kernel: per-m midranges, the lattice step w s0 >= R / 254, int8 table, eps"""
    hi, lo = Qf.max(1), Qf.min(1)
    mu = (f32(0.5) * hi + f32(0.5) * lo).astype(f32)
    A = f32(0)
    musum = f32(0)
    R = f32(0)
    for m in range(M):
        musum = f32(musum + mu[m])
        A = f32(A + max(abs(hi[m]), abs(lo[m])))
        R = max(R, f32(hi[m] - lo[m]))
    inv0 = f32(1.0) / s0
    w = f32(min(max(np.ceil(f32(f32(R / f32(254.0)) * inv0)), 1.0), 127.0))
    step = f32(w * s0)
    eps = f32(f32(16.4) * step + f32(128.0) * U * f32(pabs_max + A) + f32(64.0) * U * abs(musum))
    inv = f32(1.0) / step
    Qi = np.clip(np.rint(((Qf - mu[:, None]).astype(f32) * inv).astype(f32)), -127, 127).astype(np.int32)
    return int(w), step, musum, A, R, eps, Qi


def _base_step(rmax):
    """pqi_base_step: the power of two s0 with 127 s0 >= rmax / 254"""
    _, e = np.frexp(f32(rmax / f32(254.0 * 127.0)))
    return f32(np.ldexp(1.0, int(max(e, -100))))


@pytest.mark.parametrize("is_l2", [True, False], ids=["l2", "ip"])
@pytest.mark.parametrize("scale", [1e-3, 1.0, 300.0])
@pytest.mark.parametrize("mode", ["random", "one_signed", "mixed_magnitudes"])
def test_int8_adc_bound_holds_on_lattice_steps(is_l2, scale, mode):
    """integer form: |approx - exact| <= eps = 16.4 s_q + fp32 terms with s_q rounded UP to the batch's lattice (w s0); the
    batch's largest range comes from another query of up to 40 x the magnitude, so small weights (coarse lattice points)
    are exercised; the weight is an integer in [1, 127] and the table stays inside int8"""
    rng = np.random.default_rng(int(scale * 11) % 1000 + len(mode) + (5 if is_l2 else 0))
    q, cb = _case(rng, scale, mode)
    Qf = _tables(q, cb, is_l2)
    cen = (rng.standard_normal(M * DSUB) * scale).astype(f32)
    P = np.zeros((M, KSUB), f32)
    if is_l2:
        for m in range(M):
            P[m] = (cb[m] * cb[m]).sum(1) + f32(2) * (cb[m] * cen[m * DSUB:(m + 1) * DSUB]).sum(1)
    codes = rng.integers(0, KSUB, (48, M))
    codes[0] = (Qf - Qf.min(1, keepdims=True)).argmax(1)  # every entry at its column maximum
    ar = np.arange(M)
    pabs_max = f32(np.abs(P[ar, codes]).astype(f32).sum(1, dtype=f32).max()) if is_l2 else f32(0)
    R_q = f32((Qf.max(1) - Qf.min(1)).max())
    for blow in (1.0, 3.7, 40.0):  # the batch's widest query relative to this one
        s0 = _base_step(f32(R_q * f32(blow)))
        w, step, musum, A, R, eps_base, Qi = _int8_prep(Qf, s0, pabs_max)
        assert 1 <= w <= 127 and float(step) >= float(R) / 254.0 and np.abs(Qi).max() <= 127
        dis0 = f32(abs(rng.standard_normal()) * scale * scale * 40)
        eps = f32(eps_base + f32(64.0) * U * f32(abs(dis0) + abs(dis0) + abs(musum)))
        worst = 0.0
        for row in codes:
            acc = f32(0)
            for m in range(M):
                lut = f32(P[m, row[m]] + Qf[m, row[m]]) if is_l2 else Qf[m, row[m]]
                acc = f32(acc + lut)
            exact = f32(dis0 + acc)
            S = int(Qi[ar, row].sum())  # int32 accumulation: exact
            ps = f32(0)
            for m in range(M):
                ps = f32(ps + P[m, row[m]])
            # kernel: fma(S, s_q, ps) against ((tau + eps) - dis0) - musum; pessimistic value fma(S, s_q, (dis0 + musum) + ps)
            v = np.float32(np.float64(f32(S)) * np.float64(step) + np.float64(ps if is_l2 else f32(0)))
            approx = f32(f32(v + musum) + dis0)
            err = abs(float(approx) - float(exact))
            assert err <= float(eps), (blow, err, float(eps))
            worst = max(worst, err / float(eps))
        assert worst < 0.75, f"the bound holds but with little room: {worst:.3f} of eps (blow {blow})"


@pytest.mark.parametrize("is_l2", [True, False], ids=["l2", "ip"])
@pytest.mark.parametrize("scale", [1e-3, 1.0, 300.0])
@pytest.mark.parametrize("mode", ["random", "one_signed", "mixed_magnitudes"])
def test_sample_pass_value_is_pessimistic(is_l2, scale, mode):
    """pq_sample_kernel: dump = ((dis0 + psum) + sum_m Qf[m][code]) made pessimistic by
    slack = 128 u (pabs_max + A) + 64 u |dis0| + 64 u |value|: it must never be BETTER than the reference's exact distance
    (tau_q, the k-th best dump value, has to bound the k-th exact distance from the safe side), and it should stay close."""
    rng = np.random.default_rng(int(scale * 13) % 1000 + len(mode) + (7 if is_l2 else 0))
    q, cb = _case(rng, scale, mode)
    Qf = _tables(q, cb, is_l2)
    cen = (rng.standard_normal(M * DSUB) * scale).astype(f32)
    P = np.zeros((M, KSUB), f32)
    if is_l2:
        for m in range(M):
            P[m] = (cb[m] * cb[m]).sum(1) + f32(2) * (cb[m] * cen[m * DSUB:(m + 1) * DSUB]).sum(1)
    codes = rng.integers(0, KSUB, (64, M))
    ar = np.arange(M)
    pabs_max = f32(np.abs(P[ar, codes]).astype(f32).sum(1, dtype=f32).max()) if is_l2 else f32(0)
    A = f32(0)
    for m in range(M):
        A = f32(A + np.abs(Qf[m]).max())
    dis0 = f32(abs(rng.standard_normal()) * scale * scale * 40)
    slack0 = f32(f32(128.0) * U * f32(pabs_max + A) + f32(64.0) * U * abs(dis0))
    for row in codes:
        acc = f32(0)
        for m in range(M):
            lut = f32(P[m, row[m]] + Qf[m, row[m]]) if is_l2 else Qf[m, row[m]]
            acc = f32(acc + lut)
        exact = f32(dis0 + acc)
        ps = f32(0)
        for m in range(M):
            ps = f32(ps + P[m, row[m]])
        t = f32(0)
        for m in range(M):
            t = f32(t + Qf[m, row[m]])
        val = f32(f32(dis0 + ps) + t) if is_l2 else f32(dis0 + t)
        slack = f32(slack0 + f32(64.0) * U * abs(val))
        pess = f32(val + slack) if is_l2 else f32(val - slack)
        if is_l2:
            assert float(pess) >= float(exact), (float(pess), float(exact))
        else:
            assert float(pess) <= float(exact), (float(pess), float(exact))
        assert abs(float(pess) - float(exact)) <= 2.5 * float(slack) + 1e-30


# ---- the finish kernel's prune (mfma_scan.hip, KIND 2): its eps_max must dominate the eps of every emission --------------
@pytest.mark.parametrize("is_l2", [True, False], ids=["l2", "ip"])
@pytest.mark.parametrize("form", ["half", "int8"])
@pytest.mark.parametrize("mode", ["random", "one_signed", "mixed_magnitudes"])
def test_finish_prune_bound_dominates_every_emission(is_l2, form, mode):
    """mscan_finish_kernel prunes a candidate when its OPTIMISTIC distance `pess -+ 2 eps_max` is beyond the k-th best
    pessimistic distance tau2.  That is only sound while `exact >= pess - 2 eps_max` (L2; IP mirrored) for EVERY
    emitted candidate: the candidate's pess was built with the emission's own eps = eps_base + 64 u (|dis0| + |tau|
    [+ |sum mu|, integer form]) where dis0 is the probe's coarse distance (any sign for the inner product, the farthest
    probe has |dis0| = cmax) and tau is whatever bound the unit read -- the sample bound gthr or a tighter histogram
    edge, never beyond tau2.  Replay: several probes per query, tau swept over [gthr, tau2], pess as the kernel builds
    it, eps_max by the finish kernel's formula; the inequality must hold for each, and a candidate the prune DROPS
    must be strictly worse than k exact distances."""
    rng = np.random.default_rng(101 + 7 * len(mode) + (3 if is_l2 else 0) + (11 if form == "int8" else 0))
    scale = 30.0
    q, cb = _case(rng, scale, mode)
    Qf = _tables(q, cb, is_l2)
    ar = np.arange(M)
    nprobe, rows, k = 6, 40, 5
    # coarse distances: L2 positive and growing, IP of either sign with a large positive head (positive-score IP)
    dis0s = (np.sort(np.abs(rng.standard_normal(nprobe))) * scale * scale * 60).astype(f32)
    if not is_l2:
        dis0s = (dis0s[::-1] - f32(scale * scale * 20)).astype(f32)
    cmax = f32(np.abs(dis0s).max())
    P_all, codes_all = [], []
    for p in range(nprobe):
        cen = (rng.standard_normal(M * DSUB) * scale).astype(f32)
        P = np.zeros((M, KSUB), f32)
        if is_l2:
            for m in range(M):
                P[m] = (cb[m] * cb[m]).sum(1) + f32(2) * (cb[m] * cen[m * DSUB:(m + 1) * DSUB]).sum(1)
        P_all.append(P)
        codes_all.append(rng.integers(0, KSUB, (rows, M)))
    pabs_max = f32(max(np.abs(P_all[p][ar, c]).astype(f32).sum(dtype=f32) for p in range(nprobe) for c in codes_all[p])) \
        if is_l2 else f32(0)
    if form == "half":
        A, sc, Qh, eps_base = _query_prep(Qf, pabs_max)
        isc = f32(1.0) / sc
        musum = f32(0)
    else:
        R_q = f32((Qf.max(1) - Qf.min(1)).max())
        s0 = _base_step(f32(R_q * f32(3.7)))
        w, step, musum, A, R, eps_base, Qi = _int8_prep(Qf, s0, pabs_max)
    # exact distances of everything (the reference's sequence)
    exact = np.zeros((nprobe, rows), f32)
    for p in range(nprobe):
        for r, row in enumerate(codes_all[p]):
            acc = f32(0)
            for m in range(M):
                lut = f32(P_all[p][m, row[m]] + Qf[m, row[m]]) if is_l2 else Qf[m, row[m]]
                acc = f32(acc + lut)
            exact[p, r] = f32(dis0s[p] + acc)
    flat = np.sort(exact.ravel())
    kth = flat[k - 1] if is_l2 else flat[-k]
    # gthr: a loose sample bound; the emission's tau anywhere between it and (about) the final k-th value
    gthr = f32(kth + f32(abs(kth)) * f32(0.5) + f32(scale * scale)) if is_l2 else f32(kth - f32(abs(kth)) * f32(0.5) - f32(scale * scale))
    def emit(tau_of):
        pess_all, exact_all, eps_all = [], [], []
        for p in range(nprobe):
            dis0 = dis0s[p]
            for r, row in enumerate(codes_all[p]):
                tau = tau_of()
                eps = f32(eps_base + f32(64.0) * U * f32(f32(abs(dis0) + abs(tau)) + (abs(musum) if form == "int8" else f32(0))))
                ps = f32(0)
                for m in range(M):
                    ps = f32(ps + P_all[p][m, row[m]])
                if form == "half":
                    h = f32(0)
                    for m in range(M):
                        h = f32(h + f32(Qh[m, row[m]]))
                    pcs = f32(dis0 + eps) if is_l2 else f32(dis0 - eps)
                    pess = np.float32(np.float64(h) * np.float64(isc) + np.float64(f32(pcs + ps) if is_l2 else pcs))
                else:
                    S = int(Qi[ar, row].sum())
                    pcs = f32(f32(dis0 + musum) + eps) if is_l2 else f32(f32(dis0 + musum) - eps)
                    pess = np.float32(np.float64(f32(S)) * np.float64(step) + np.float64(f32(pcs + ps) if is_l2 else pcs))
                pess_all.append(pess)
                exact_all.append(exact[p, r])
                eps_all.append(eps)
        return np.array(pess_all, f32), np.array(exact_all, f32), np.array(eps_all, f32)

    # the loosest emission (every unit read the sample bound) gives a provisional k-th pessimistic value; a histogram edge
    # a unit can read lies between the sample bound and that value (k candidates must sit below it), and tightening tau
    # only moves the final tau2 further to the good side
    p0, _, _ = emit(lambda: gthr)
    s0_ = np.sort(p0)
    tau2_0 = s0_[k - 1] if is_l2 else s0_[-k]
    pess_all, exact_all, eps_all = emit(lambda: f32(gthr + (tau2_0 - gthr) * f32(rng.random())))
    # every candidate's pess is on the safe side of its exact distance
    assert ((pess_all >= exact_all) if is_l2 else (pess_all <= exact_all)).all()
    srt = np.sort(pess_all)
    tau2 = srt[k - 1] if is_l2 else srt[-k]
    # the finish kernel's formula
    eps_max = f32(eps_base + f32(64.0) * f32(5.9604645e-8) *
                  f32(f32(cmax + max(f32(abs(gthr)), f32(abs(tau2)))) + (abs(musum) if form == "int8" else f32(0))))
    opt = (pess_all - f32(2.0) * eps_max) if is_l2 else (pess_all + f32(2.0) * eps_max)
    assert ((opt <= exact_all) if is_l2 else (opt >= exact_all)).all(), "optimistic bound is not below the exact distance"
    dropped = (opt > tau2) if is_l2 else (opt < tau2)
    assert ((exact_all[dropped] > kth) if is_l2 else (exact_all[dropped] < kth)).all(), "the prune dropped a top-k row"
    # ... and it dominates the eps each emission used (what the prune's `2 eps_max` stands for)
    assert (eps_all <= eps_max).all()
    # the first version's formula (no |sum mu| term, |gthr| only) does NOT dominate them on the integer form
    eps_old = f32(eps_base + f32(64.0) * f32(5.9604645e-8) * f32(cmax + f32(abs(gthr))))
    if form == "int8" and abs(float(musum)) > 0:
        assert (eps_all > eps_old).any()


def test_sample_kernel_selection_finds_the_kth_smallest_key():
    """pq_sample_kernel selects tau_q itself: the ksel-th smallest of the dumped values as order-preserving integer keys, by
    cutting the interval [min, max] in four per step (three counters, one barrier).  The cut arithmetic replayed on integer
    sets with heavy ties, tiny and full-width spreads: the result is the ksel-th order statistic and the interval closes
    within the kernel's 18 steps."""
    rng = np.random.default_rng(77)
    worst_steps = 0
    for trial in range(3000):
        n = int(rng.integers(1, 400))
        kind = trial % 5
        if kind == 0:
            keys = rng.integers(0, 2 ** 32, n, dtype=np.uint64)
        elif kind == 1:
            keys = (np.uint64(0x42000000) + rng.integers(0, 2 ** 22, n, dtype=np.uint64))   # one exponent
        elif kind == 2:
            keys = (np.uint64(7) + rng.integers(0, 3, n, dtype=np.uint64))                 # ties everywhere
        elif kind == 3:
            keys = np.full(n, np.uint64(0xfffffffe))
        else:
            keys = np.concatenate([[0, 0xffffffff], rng.integers(0, 2 ** 32, max(n - 2, 0), dtype=np.uint64)]).astype(np.uint64)
        keys = keys.astype(np.uint64)
        n = len(keys)
        k = int(rng.integers(1, n + 1))
        lo, hi = int(keys.min()), int(keys.max())
        steps = 0
        while lo < hi:
            span = hi - lo
            q1, q2, q3 = lo + (span >> 2), lo + (span >> 1), lo + (span >> 2) + (span >> 1)
            assert lo <= q1 <= q2 <= q3 < hi
            t1, t2, t3 = int((keys <= q1).sum()), int((keys <= q2).sum()), int((keys <= q3).sum())
            if t1 >= k:
                hi = q1
            elif t2 >= k:
                lo, hi = q1 + 1, q2
            elif t3 >= k:
                lo, hi = q2 + 1, q3
            else:
                lo = q3 + 1
            steps += 1
            assert steps <= 18
        worst_steps = max(worst_steps, steps)
        assert lo == int(np.sort(keys)[k - 1]), (trial, k)
    assert worst_steps >= 14  # (the full-width cases really need most of the steps)
