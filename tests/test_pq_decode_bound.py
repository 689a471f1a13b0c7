"""CPU check of the error bound the DECODE form of the IVF-PQ prefilter relies on (knowhere_amd/csrc/pq_decode.hip).

The prefilter never decides a result: it only has to let every row through whose EXACT distance (the reference's fp32 sum in
m order, thirdparty/faiss/faiss/impl/pq_code_distance/pq_code_distance-inl.h:82-88 + IVFPQScanner_impl.h:147-150) is within
the query's bound, which holds as long as |approx - exact| <= eps.  Here the kernel's arithmetic is replayed in numpy:

    Q = half(sc_q q), Y = half(sc_y cb)       powers of two fixed per index (pqd_codebook_kernel)
    acc = -psum SC / 2 + sum_i Q_i Y_i        128 exact products accumulated in fp32 -- under the least favourable treatment
                                              the matrix core could give them: every addition rounded on its own, several
                                              orders; and with half subnormals FLUSHED to zero (the bound must hold either way)
    approx = dis0 + (-2 acc / SC)             (L2)          dis0 + acc / SC   (inner product)

against the exact sequence, with eps = pqd_query_prep_kernel's formula + the per-pair term.  The bound must hold with room
to spare on random and adversarial inputs (one-signed operands whose partial sums reach B_q, value scales from 1e-15 to
1e6, sub-quantizers far below the half range of the scaled codebook, queries far smaller / larger than the data)."""
import numpy as np
import pytest

f32 = np.float32
f16 = np.float16
U = f32(2.0 ** -24)
UH = f32(2.0 ** -11)
A_SUB = f32(2.0 ** -14)
M, KSUB, DSUB, D = 32, 256, 4, 128


def scale_for(amax):
    """pq_decode.hip::pqd_scale_for: the power of two that puts amax into [2^14, 2^15), exponent within +-60"""
    amax = float(amax)
    if not (amax > 0.0) or not np.isfinite(amax):
        return f32(1.0)
    _, e = np.frexp(amax)
    return f32(np.ldexp(1.0, int(np.clip(15 - e, -60, 60))))


def index_prep(cb, cmax):
    """pqd_codebook_kernel: scales and constants of an index"""
    ymax = f32(np.abs(cb).max())
    sc_y = scale_for(ymax)
    ysum = f32(0)
    for m in range(M):
        ysum = f32(ysum + np.abs(cb[m]).astype(f32).sum(1, dtype=f32).max())
    ysum = f32(ysum * f32(1.0001))
    sc_q = scale_for(f32(4.0) * f32(cmax + ymax))
    with np.errstate(over="ignore", under="ignore"):
        Y = (cb * sc_y).astype(f32).astype(f16)
    return dict(sc_y=sc_y, inv_y=f32(1) / sc_y, ymax=ymax, ysum=ysum, sc_q=sc_q, inv_q=f32(1) / sc_q, SC=f32(sc_q * sc_y),
                inv_sc=f32(1) / f32(sc_q * sc_y), Y=Y)


def query_prep(q, cb, ix, pabs_max, is_l2):
    """pqd_query_prep_kernel: B_q, ||q||_1, eps_base, the query as halves (None: eps = inf, the exact path)"""
    B = f32(0)
    for m in range(M):
        v = (np.abs(cb[m]) * np.abs(q[m * DSUB:(m + 1) * DSUB])[None, :]).astype(f32).sum(1, dtype=f32).max()
        B = f32(B + v)
    B = f32(B * f32(1.0001))
    q1 = f32(np.abs(q).sum(dtype=f32) * f32(1.0001))
    fits = bool((np.abs(q).astype(f32) * ix["sc_q"] < f32(65504.0)).all())
    if not fits or not np.isfinite(B):
        return None
    F = f32(2.0 if is_l2 else 1.0)
    e_prod = f32((f32(2) * UH + UH * UH) * B)
    e_sub = f32(A_SUB * (f32(1) + UH) * f32(ix["ysum"] * ix["inv_q"] + q1 * ix["inv_y"]) + f32(128) * A_SUB * A_SUB * ix["inv_sc"])
    e_acc = f32(f32(136) * U * f32((f32(1) + f32(3) * UH) * B + f32(0.5) * pabs_max))
    eps = f32(f32(F * f32(e_prod + e_sub + e_acc) + f32(128) * U * f32(pabs_max + F * B)) * f32(1.001))
    with np.errstate(over="ignore", under="ignore"):
        Q = (q * ix["sc_q"]).astype(f32).astype(f16)
    return dict(B=B, eps=eps, Q=Q)


def flush_subnormal(h):
    """half values below the normal range -> 0 (what a matrix pipe that flushes would see)"""
    x = h.astype(f32)
    return np.where(np.abs(x) < f32(2.0 ** -14), f32(0), x).astype(f32)


def _case(rng, scale, mode):
    q = (rng.standard_normal(D) * scale).astype(f32)
    cb = (rng.standard_normal((M, KSUB, DSUB)) * scale).astype(f32)
    cen = (rng.standard_normal(D) * scale * 3).astype(f32)
    if mode == "one_signed":
        q, cb = np.abs(q), np.abs(cb)
    if mode == "mixed_magnitudes":  # a few sub-quantizers dominate; the rest sit far below the scaled half range
        cb[4:] *= f32(1e-6)
    if mode == "small_query":       # a query a million times smaller than the data
        q *= f32(1e-6)
    if mode == "large_query":       # ... seven times the data's largest coordinate (still inside the half range)
        q = (q / np.abs(q).max() * (np.abs(cen).max() + np.abs(cb).max()) * f32(7.0)).astype(f32)
    return q, cb, cen


@pytest.mark.parametrize("is_l2", [True, False], ids=["l2", "ip"])
@pytest.mark.parametrize("scale", [1e-15, 1e-3, 1.0, 300.0, 1e6])
@pytest.mark.parametrize("mode", ["random", "one_signed", "mixed_magnitudes", "small_query", "large_query"])
def test_decode_form_bound_holds_with_margin(is_l2, scale, mode):
    rng = np.random.default_rng(int(np.log10(scale) * 7 + 200) + len(mode) + (3 if is_l2 else 0))
    q, cb, cen = _case(rng, scale, mode)
    ix = index_prep(cb, f32(np.abs(cen).max()))
    # term 2 of one list (a row of the precomputed table): ||cb||^2 + 2 <c, cb>
    P = np.zeros((M, KSUB), f32)
    if is_l2:
        for m in range(M):
            P[m] = ((cb[m] * cb[m]).astype(f32).sum(1, dtype=f32) +
                    f32(2) * (cb[m] * cen[m * DSUB:(m + 1) * DSUB]).astype(f32).sum(1, dtype=f32)).astype(f32)
    # the reference's per-query table: -2 <q_m, cb> (L2 with the precomputed table) / <q_m, cb> (inner product), a chain of
    # products added from 0 in dimension order
    T = np.zeros((M, KSUB), f32)
    for m in range(M):
        t = np.zeros(KSUB, f32)
        for i in range(DSUB):
            t = (t + (cb[m, :, i] * q[m * DSUB + i]).astype(f32)).astype(f32)
        T[m] = (f32(-2.0) * t).astype(f32) if is_l2 else t
    codes = rng.integers(0, KSUB, (40, M))
    if mode == "one_signed":
        codes[0] = [int(np.argmax((np.abs(cb[m]) * np.abs(q[m * DSUB:(m + 1) * DSUB])[None, :]).sum(1))) for m in range(M)]
    ar = np.arange(M)
    pabs_max = f32(np.abs(P[ar, codes]).astype(f32).sum(1, dtype=f32).max()) if is_l2 else f32(0)
    qp = query_prep(q, cb, ix, pabs_max, is_l2)
    if qp is None:
        assert mode == "large_query" or scale >= 1e6 or scale <= 1e-15  # (out of the half range: the exact kernels)
        return
    dis0 = f32(abs(rng.standard_normal()) * scale * scale * 40)
    tau = dis0  # (enters eps only through the roundings of the threshold)
    eps = f32(qp["eps"] + f32(64.0) * U * f32(abs(dis0) + abs(tau)))
    if not np.isfinite(eps):
        return
    Qn, Yn = qp["Q"].astype(f32), ix["Y"].astype(f32)
    Qf, Yf = flush_subnormal(qp["Q"]), flush_subnormal(ix["Y"])
    worst = 0.0
    for row in codes:
        # exact: LUT entry = term2 + table entry rounded once (fvec_madd), summed from 0 in m order, dis0 last
        acc = f32(0)
        for m in range(M):
            lut = f32(P[m, row[m]] + T[m, row[m]]) if is_l2 else T[m, row[m]]
            acc = f32(acc + lut)
        exact = f32(dis0 + acc)
        ps = f32(0)
        for m in range(M):
            ps = f32(ps + P[m, row[m]])
        start = f32(ps * f32(-0.5) * ix["SC"]) if is_l2 else f32(0)  # (a power-of-two multiple: exact)
        for Qs, Ys in ((Qn, Yn), (Qf, Yf)):
            y = np.concatenate([Ys[m, row[m]] for m in range(M)])  # the decoded row, dimension order
            prod = (Qs.astype(np.float64) * y.astype(np.float64))   # products of halves: exact in fp32
            assert (prod == prod.astype(f32)).all()
            for order in (np.arange(D), np.arange(D)[::-1], (5 * np.arange(D) + 3) % D, np.argsort(-np.abs(prod))):
                a = start
                for i in order:
                    a = f32(a + f32(prod[i]))
                x = f32(a * ix["inv_sc"])
                approx = f32(dis0 - f32(2.0) * x) if is_l2 else f32(dis0 + x)
                err = abs(float(approx) - float(exact))
                assert err <= float(eps), (err, float(eps), mode, scale)
                worst = max(worst, err / float(eps))
    assert worst < 0.8, f"the bound holds but with little room: {worst:.3f} of eps"


def test_threshold_in_accumulator_units_is_a_superset_of_the_distance_test():
    """pqd_kernel's pair prologue: acc >= SC ((dis0 - tau - eps) / 2) (L2) / acc >= SC (tau - eps - dis0) (inner product) must
    pass whenever the pessimistic distance the finish later sees (c -+ 2 acc / SC with c = dis0 +- eps) is within tau + ...:
    every row the distance test would keep, the accumulator test keeps."""
    rng = np.random.default_rng(11)
    for trial in range(20000):
        SC = f32(np.ldexp(1.0, int(rng.integers(-10, 40))))
        inv = f32(1) / SC
        dis0 = f32(rng.standard_normal() * 100)
        tau = f32(dis0 + rng.standard_normal() * 10)
        eps = f32(abs(rng.standard_normal()) * 0.1 + 64 * float(U) * (abs(dis0) + abs(tau)))
        x = f32(rng.standard_normal() * 8)          # <q, y> - psum / 2 in true units
        acc = f32(x * SC)
        # L2: approx = dis0 - 2 x <= tau + eps'  with eps' the part of eps that is not threshold arithmetic
        eps_arith = f32(64 * float(U) * (abs(dis0) + abs(tau)))
        if float(dis0) - 2.0 * float(x) <= float(tau) + float(eps) - float(eps_arith):
            t = f32(SC * f32(f32(f32(dis0 - tau) - eps) * f32(0.5)))
            assert acc >= t, (trial, "l2")
        if float(dis0) + float(x) >= float(tau) - float(eps) + float(eps_arith):
            t = f32(SC * f32(f32(tau - eps) - dis0))
            assert acc >= t, (trial, "ip")
        assert f32(acc * inv) == x  # (power-of-two scale: exact)


def test_lane_map_covers_every_dimension_once():
    """pq_decode.hip: lane (r, h) of step s holds the entries of sub-quantizers 16 h + 2 s, + 1 (bytes 2 s, 2 s + 1 of its 16
    contiguous code bytes) as the A operand's k block h; the B operand's lane (n, h) holds the query's dimensions
    64 h + 8 s .. + 8: over the eight steps every dimension meets its own partner exactly once"""
    seen = np.zeros(D, int)
    for s_ in range(8):
        for h in range(2):
            a_dims = [4 * (16 * h + 2 * s_ + e) + i for e in range(2) for i in range(4)]   # decoded entries, in k order
            b_dims = [64 * h + 8 * s_ + i for i in range(8)]                                # query halves, in k order
            assert a_dims == b_dims
            seen[a_dims] += 1
    assert (seen == 1).all()
    # accumulator layout of v_mfma_f32_32x32x16_f16: element i of lane (n, h) is row (i & 3) + 8 (i >> 2) + 4 h -- what the
    # parked record's first row (32 t + 4 h) and the flush's `+ (r & 3) + 8 (r >> 2)` assume; and the start values' float4
    # loads of the first version: rows 8 j + 4 h .. + 4 = elements 4 j .. 4 j + 3
    rows = sorted((i & 3) + 8 * (i >> 2) + 4 * h for h in range(2) for i in range(16))
    assert rows == list(range(32))
