"""Quantised refine store (refine_type = fp16 / bf16 / sq8 / sq6 / int8 / sq4u): the oracle's restatement pinned against the reference.

Knowhere builds IndexRefine(base, faiss::IndexScalarQuantizer(d, QT_fp16 / QT_bf16 / QT_8bit / QT_6bit /
QT_8bit_direct_signed, metric)) for these refine
types (reference src/index/refine/refine_utils.cc:150-185).  oracle.c restates the quantizer (train / encode / decode) and
the refine distance computer; here the restatement must produce the reference's code bytes, trained ranges and search
results bit for bit (oracle/_ref = the reference's own sources, SIMDLevel::NONE)."""
import numpy as np
import pytest

from conftest import gen_data
from oracle import binding as ob

ROW_TYPES = [(1, "fp16"), (2, "bf16"), (3, "sq8"), (4, "sq6"), (5, "int8"), (6, "sq4u")]
TRAINED = (3, 4, 6)  # types with trained ranges (3, 4: per dimension; 6: one for all, from quantiles for L2)


def _train(port, row_type, x, metric=ob.L2):
    if row_type == 6:
        return port.rows_train_uniform(metric, x)
    return port.rows_train(x) if row_type in TRAINED else None


def _data(row_type, *a, **kw):
    """int8 stores hold integer values in [-128, 127] (Knowhere's int8 data format); everything else takes the floats"""
    x = gen_data(*a, **kw)
    if row_type == 5:
        x = np.clip(np.rint((x - x.mean()) / (x.std() + 1e-9) * 40.0), -128, 127).astype(np.float32)
    return x


def _nasty(d, seed):
    """values that exercise the rounding rules: half ulps (ties), subnormal halves, overflow to inf, signed zeros"""
    rng = np.random.default_rng(seed)
    v = [0.0, -0.0, 1.0, -1.0, 65504.0, 65519.9, 65520.0, 70000.0, -70000.0, 1e-8, 2.0 ** -25, 2.0 ** -24, 3 * 2.0 ** -25,
         2.0 ** -14, 2.0 ** -14 - 2.0 ** -25, 1 + 2.0 ** -11, 1 + 3 * 2.0 ** -11, 1 + 2.0 ** -11 + 2.0 ** -20, 1 + 2.0 ** -8,
         1 + 3 * 2.0 ** -8, 3.3895314e38, 1e-40, -1e-40, 1234.5678, np.pi]
    x = np.concatenate([np.array(v, np.float32), (rng.standard_normal(4000) * 10 ** rng.uniform(-9, 5, 4000)).astype(np.float32)])
    n = (len(x) + d - 1) // d * d
    x = np.concatenate([x, np.zeros(n - len(x), np.float32)])
    return np.ascontiguousarray(x.reshape(-1, d))


@pytest.mark.parametrize("row_type,name", ROW_TYPES, ids=[r[1] for r in ROW_TYPES])
def test_rows_encode_equals_the_reference(port, ref, row_type, name):
    d = 24
    sets = (_data(row_type, 700, d, 5), _data(row_type, 300, d, 6, -3.0, 3.0)) + (() if row_type == 5 else (_nasty(d, 7),))
    for x in sets:
        if row_type in TRAINED:
            x = x[np.isfinite(x).all(1)]
        for metric in ((ob.L2, ob.IP) if row_type == 6 else (ob.L2,)):  # (sq4u: quantile range for L2, min / max else)
            codes_r, tr_r = ref.sq_rows(row_type, metric, x)
            tr = _train(port, row_type, x, metric)
            if row_type in TRAINED:
                assert tr.tobytes() == tr_r.tobytes(), f"{name} ranges"
            assert port.rows_encode(row_type, x, tr).tobytes() == codes_r.tobytes(), f"{name} code bytes"
        codes = port.rows_encode(row_type, x, tr)
        assert codes.tobytes() == codes_r.tobytes(), f"{name} code bytes"


def test_sq8_constant_column_and_clamp(port, ref):
    """vdiff == 0 encodes 0; rows outside the trained range clamp (encode with ranges trained on OTHER rows)"""
    d = 8
    x = gen_data(200, d, 3)
    x[:, 2] = 7.5
    codes_r, tr_r = ref.sq_rows(3, ob.L2, x)
    tr = port.rows_train(x)
    assert tr.tobytes() == tr_r.tobytes() and tr[d + 2] == 0
    assert port.rows_encode(3, x, tr).tobytes() == codes_r.tobytes()
    wide = gen_data(50, d, 4, -100.0, 300.0)
    c = port.rows_encode(3, wide, tr)
    assert c.min() == 0 and c.max() == 255
    back = port.rows_decode(3, d, c, tr)
    assert (back[:, 2] == 7.5).all()


@pytest.mark.parametrize("row_type,name", ROW_TYPES, ids=[r[1] for r in ROW_TYPES])
@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
@pytest.mark.parametrize("kind", [ob.IVF_PQ, ob.IVF_SQ8], ids=["ivfpq", "ivfsq8"])
def test_refine_rows_equals_index_refine_over_a_scalar_quantizer(port, ref, kind, metric, row_type, name):
    nb, nq, d, nlist, M = 2500, 24, 48, 20, 12
    xb, xq = _data(row_type, nb, d, 42), _data(row_type, nq, d, 44)
    h = ref.create(kind, metric, d, nlist, M, 8)
    try:
        ref.train_add(h, xb)
        ix = ref.export(h, kind, metric, d, nlist, M, 8)
        tr = _train(port, row_type, xb, metric)
        codes = port.rows_encode(row_type, xb, tr)
        for k, kf, nprobe in ((5, 4.0, 18), (1, 8.0, 3), (10, 1.0, 20)):
            Dr, Ir = ref.search_refine_sq(h, row_type, xb, xq, k, kf, nprobe)
            kb = int(k * kf)
            _, Ib = port.search(ix, xq, kb, nprobe)
            Dp, Ip = port.refine_rows(metric, row_type, d, codes, tr, xq, Ib, k)
            assert Dr.tobytes() == Dp.tobytes(), f"{name} k={k} k_factor={kf}: distances"
            assert (Ir == Ip).all(), f"{name} k={k} k_factor={kf}: ids"
    finally:
        ref.destroy(h)


@pytest.mark.parametrize("row_type,name", ROW_TYPES, ids=[r[1] for r in ROW_TYPES])
def test_decode_round_trip_properties(port, row_type, name):
    d = 16
    x = _data(row_type, 500, d, 9, -50.0, 50.0)
    tr = _train(port, row_type, x)
    c = port.rows_encode(row_type, x, tr)
    y = port.rows_decode(row_type, d, c, tr)
    # idempotence: re-encoding the decoded rows gives the same codes
    assert port.rows_encode(row_type, y, tr).tobytes() == c.tobytes()
    if row_type == 1:
        # the reference's scalar encode_fp16 rounds exact ties UP; everywhere else it is IEEE round-to-nearest (= numpy's half)
        rne = x.astype(np.float16).astype(np.float32)
        tie = (x.view(np.uint32) & 0x1fff) == 0x1000
        assert np.array_equal(y[~tie], rne[~tie])
        assert (np.abs(y) >= np.abs(rne)).all()
    elif row_type == 2:
        assert np.abs(y - x).max() <= np.abs(x).max() * 2.0 ** -8
    elif row_type == 5:
        assert np.array_equal(y, x)  # (integer values in range: lossless)
    elif row_type == 6:
        inside = (x >= tr[0]) & (x <= tr[0] + tr[1])  # (2 % of the values lie outside the quantile range: clamped)
        assert 0.97 < inside.mean() < 0.99
        assert (np.abs(y - x)[inside] <= tr[1] / 15.0 * 0.5001 + 1e-5).all()
    else:
        levels = 255.0 if row_type == 3 else 63.0
        assert (np.abs(y - x) <= tr[d:] / levels * 0.5001 + 1e-6).all()


def _encode_fp16_int(bits):
    """the kernel's integer form of the reference's scalar encode_fp16 (refine.hip rows_encode16_kernel), in numpy"""
    bits = bits.astype(np.int64)
    sign = (bits >> 16) & 0x8000
    fint = bits & 0x7fffffff
    t = fint & 0xfffff000
    E = t >> 23
    mant = (t & 0x7fffff) | np.where(E > 0, 0x800000, 0)
    s = np.clip(113 - E, 0, 62)
    sub = np.where(s <= 12, mant >> s, 0)
    b = np.where(E >= 113, t - (112 << 23), sub)
    b = np.minimum(b, (31 << 23) - 0x1000)
    o = (b + 0x1000) >> 13
    o = np.where(fint > 0x7f800000, 0x7e00, np.where(fint == 0x7f800000, 0x7c00, o))
    return (o | sign).astype(np.uint16)


def test_fp16_encode_integer_form_is_exhaustively_the_reference(port):
    """encode_fp16 depends on the top 20 bits of |x| only (the low 12 are masked first): all 2^19 classes x both signs,
    with the low bits clear, set and random"""
    hi = np.arange(1 << 19, dtype=np.uint32) << 12
    rng = np.random.default_rng(1)
    for low in (0, 0xfff, rng.integers(0, 0x1000, hi.size, dtype=np.uint32)):
        for sgn in (0, 0x80000000):
            bits = (hi | low | sgn).astype(np.uint32)
            x = bits.view(np.float32).reshape(-1, 64)
            want = port.rows_encode(1, x, None).view(np.uint16).ravel()
            got = _encode_fp16_int(bits)
            assert np.array_equal(want, got)


@pytest.mark.parametrize("row_type,name", ROW_TYPES, ids=[r[1] for r in ROW_TYPES])
@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_refine_rows_ties_equal_the_reference(port, ref, metric, row_type, name):
    """duplicated raw rows encode to the same code and tie exactly after the re-rank: which of them the reference returns
    depends on where they stood in the first stage's result (reorder_2_heaps pushes them in candidate order) -- the
    restatement must make the same choice"""
    rng = np.random.default_rng(11)
    d, nb, nq = 24, 3000, 40
    proto = (rng.integers(-3, 4, (30, d)) * 7.0).astype(np.float32)
    xb = np.ascontiguousarray(proto[rng.integers(0, 30, nb)])
    xq = np.ascontiguousarray(proto[rng.integers(0, 30, nq)] + rng.integers(0, 2, (nq, d)).astype(np.float32))
    h = ref.create(ob.IVF_SQ8, metric, d, 16, 0, 8)
    try:
        ref.train_add(h, xb)
        ix = ref.export(h, ob.IVF_SQ8, metric, d, 16, 0, 8)
        tr = _train(port, row_type, xb, metric)
        codes = port.rows_encode(row_type, xb, tr)
        ties = 0
        for k, kf, nprobe in ((6, 10.0, 9), (10, 3.0, 16), (4, 1.0, 5)):
            Dr, Ir = ref.search_refine_sq(h, row_type, xb, xq, k, kf, nprobe)
            _, Ib = port.search(ix, xq, int(k * kf), nprobe)
            Dp, Ip = port.refine_rows(metric, row_type, d, codes, tr, xq, Ib, k)
            assert Dr.tobytes() == Dp.tobytes() and (Ir == Ip).all(), f"{name} k={k} k_factor={kf}"
            ties += int((Dr[:, :-1] == Dr[:, 1:]).sum())
        assert ties > 0, "no tied distances in the results: the data tests nothing"
    finally:
        ref.destroy(h)


def test_sq6_packing_and_the_double_product(port, ref):
    """four 6-bit codes per three bytes, a ragged last group (d = 10: 8 bytes), clamping, a constant column; and the
    encoder's `x * 63.0` is a double product: values whose float product would round up to the next integer keep the
    lower code"""
    d = 10
    x = gen_data(400, d, 13, -5.0, 5.0)
    x[:, 3] = -2.25
    codes_r, tr_r = ref.sq_rows(4, ob.L2, x)
    tr = port.rows_train(x)
    assert tr.tobytes() == tr_r.tobytes() and tr[d + 3] == 0
    c = port.rows_encode(4, x, tr)
    assert c.shape == (400, 8) and c.tobytes() == codes_r.tobytes()
    y = port.rows_decode(4, d, c, tr)
    assert (y[:, 3] == -2.25).all()
    wide = gen_data(60, d, 14, -100.0, 100.0)
    back = port.rows_decode(4, d, port.rows_encode(4, wide, tr), tr)
    # (a code decodes to the middle of its cell: the top code lands half a cell above the trained maximum)
    assert (back >= tr[:d] - 1e-6).all() and (back <= tr[:d] + tr[d:] * (63.5 / 63.0) + 1e-5).all()
    # one column with range exactly [0, 1]: xi = x; the largest floats below j / 63 must encode to j - 1
    j = np.arange(1, 64, dtype=np.float64)
    below = np.nextafter((j / 63.0).astype(np.float32), np.float32(0))
    col = np.concatenate([[0.0, 1.0], below, (j / 63.0).astype(np.float32)]).astype(np.float32)
    xx = np.zeros((col.size, 4), np.float32)
    xx[:, 0] = col
    cr, trr = ref.sq_rows(4, ob.L2, xx)
    t2 = port.rows_train(xx)
    assert t2.tobytes() == trr.tobytes() and t2[0] == 0 and t2[4] == 1
    assert port.rows_encode(4, xx, t2).tobytes() == cr.tobytes()
