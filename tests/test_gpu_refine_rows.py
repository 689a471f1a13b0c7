"""GPU tests (-m gpu): the quantised refine store (knhip_rows; Knowhere's refine_type = fp16 / bf16 / sq8 / sq6 / int8 / sq4u).

The reference re-ranks the first stage's candidates against a faiss::IndexScalarQuantizer of the raw rows (reference
src/index/refine/refine_utils.cc:150-185, thirdparty/faiss/faiss/cppcontrib/knowhere/IndexRefine.cpp:66-165).  The oracle's
restatement is pinned against the reference in tests/test_refine_rows.py; here the device side -- range training, the five
encoders, append, and knhip_search_refine_rows -- must equal it bit for bit."""
import numpy as np
import pytest

from conftest import assert_parity, gen_data
from helpers import finish_ivfpq
from oracle import binding as ob
from test_refine_rows import ROW_TYPES, TRAINED, _data, _nasty, _train

pytestmark = pytest.mark.gpu


def _store(rt, xb, chunks=1, metric=ob.L2):
    from knowhere_amd import RowStore
    rows = RowStore(rt, xb.shape[1], device=0)
    if rt == 6:  # (sq4u: one range for all dimensions; Knowhere trains it from the 1 % / 99 % quantiles for L2)
        rows.train_uniform(xb, 2 if metric == ob.L2 else 0, 0.01 if metric == ob.L2 else 0.0)
    else:
        rows.train(xb)
    for part in np.array_split(xb, chunks):
        rows.add(part)
    return rows


@pytest.mark.parametrize("row_type,name", ROW_TYPES, ids=[r[1] for r in ROW_TYPES])
def test_device_encoders_write_the_reference_code_bytes(port, row_type, name):
    d = 24
    sets = (_data(row_type, 5000, d, 5), _data(row_type, 300, d, 6, -3.0, 3.0)) + (() if row_type == 5 else (_nasty(d, 7),))
    for x in sets:
        if row_type in TRAINED:
            x = np.ascontiguousarray(x[np.isfinite(x).all(1)])
        for metric in ((ob.L2, ob.IP) if row_type == 6 else (ob.L2,)):
            rows = _store(row_type, x, chunks=3, metric=metric)
            tr = _train(port, row_type, x, metric)
            if row_type in TRAINED:
                assert rows.trained().tobytes() == tr.tobytes(), "ranges (column minimum / maximum - minimum; sq4u: quantiles)"
            assert rows.count() == len(x)
            assert rows.codes().tobytes() == port.rows_encode(row_type, x, tr).tobytes(), f"{name} code bytes"
            rows.close()


def test_sq8_store_constant_column_and_rows_outside_the_trained_range(port):
    from knowhere_amd import RowStore
    d = 8
    x = gen_data(200, d, 3)
    x[:, 2] = 7.5
    rows = RowStore(3, d, device=0)
    rows.train(x)
    tr = port.rows_train(x)
    assert rows.trained().tobytes() == tr.tobytes()
    wide = gen_data(50, d, 4, -100.0, 300.0)
    rows.add(wide)  # (Add after Train with rows the ranges have not seen: clamped to 0 / 255)
    assert rows.codes().tobytes() == port.rows_encode(3, wide, tr).tobytes()
    rows.close()


def test_store_contract_errors():
    from knowhere_amd import KnhipError, RowStore
    with pytest.raises(KnhipError):
        RowStore(9, 8, device=0)
    rows = RowStore(3, 8, device=0)
    with pytest.raises(KnhipError):  # sq8 before its ranges are trained
        rows.add(gen_data(4, 8, 1))
    rows.close()


KINDS = [(ob.IVF_PQ, "ivfpq", dict(nlist=24, M=8)), (ob.IVF_SQ8, "ivfsq8", dict(nlist=24)), (ob.IVF_FLAT, "ivfflat", dict(nlist=24))]


@pytest.mark.parametrize("row_type,name", ROW_TYPES, ids=[r[1] for r in ROW_TYPES])
@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
@pytest.mark.parametrize("kind,kname,kw", KINDS, ids=[k[1] for k in KINDS])
def test_search_refine_rows_equals_index_refine_over_a_scalar_quantizer(port, kind, kname, kw, metric, row_type, name):
    from knowhere_amd import GpuIndex
    nb, nq, d = 6000, 64, 32
    xb, xq = _data(row_type, nb, d, 42, -20.0, 80.0), _data(row_type, nq, d, 44, -20.0, 80.0)
    ix = ob.make_index(port, kind, metric, xb, **kw)
    if kind == ob.IVF_PQ:
        finish_ivfpq(port, ix)
    g = GpuIndex.from_data(ix, device=0)
    rows = _store(row_type, xb, chunks=2, metric=metric)
    tr = _train(port, row_type, xb, metric)
    codes = port.rows_encode(row_type, xb, tr)
    bs = np.packbits(np.random.default_rng(5).random(nb) < 0.3, bitorder="little")
    for k, kb, nprobe in ((10, 40, 8), (1, 16, 3), (7, 7, 9), (20, 200, 24)):
        for bitset, nbits in ((None, 0), (bs, nb)):
            _, Ib = port.search(ix, xq, kb, nprobe, bitset, nbits)
            Do, Io = port.refine_rows(metric, row_type, d, codes, tr, xq, Ib, k)
            D, I = g.search_refine_rows(rows, xq, k, kb, nprobe, bitset, nbits)
            assert_parity(Do, Io, D, I, metric, f"{kname} {name} k={k} k_base={kb} bitset={bitset is not None}")
    rows.close()
    g.close()


@pytest.mark.parametrize("row_type,name", ROW_TYPES, ids=[r[1] for r in ROW_TYPES])
@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_refine_rows_ties_follow_reorder_2_heaps(port, metric, row_type, name):
    """duplicated raw rows encode to the same code and tie exactly after the re-rank: which of them are returned depends on
    where they stood in the first stage's result (reorder_2_heaps) -- as for the fp32 store (tests/test_gpu_ties.py)"""
    from knowhere_amd import GpuIndex
    from test_gpu_ties import _dup_data
    xb, xq = _dup_data(6000, 32, 40, 11)
    ix = ob.make_index(port, ob.IVF_SQ8, metric, xb, nlist=24)
    g = GpuIndex.from_data(ix, device=0)
    rows = _store(row_type, xb, metric=metric)
    tr = _train(port, row_type, xb, metric)
    codes = port.rows_encode(row_type, xb, tr)
    kbase, k, nprobe = 60, 6, 9
    _, Ib = port.search(ix, xq, kbase, nprobe)
    Do, Io = port.refine_rows(metric, row_type, 32, codes, tr, xq, Ib, k)
    D, I = g.search_refine_rows(rows, xq, k, kbase, nprobe)
    assert_parity(Do, Io, D, I, metric, f"{name} refine over duplicates")
    rows.close()
    g.close()


def test_refine_rows_at_scale_properties(port):
    """100k x 128 rows, batch 2000: (a) every returned id is among the first stage's candidates, (b) the distances are the
    decoded rows' distances recomputed on the host, sorted, (c) equal to the oracle on a sample of the queries"""
    from knowhere_amd import GpuIndex
    nb, nq, d, k, kb, nprobe = 100_000, 2000, 128, 10, 100, 16
    xb, xq = gen_data(nb, d, 1, -1.0, 1.0), gen_data(nq, d, 2, -1.0, 1.0)
    ix = ob.make_index(port, ob.IVF_PQ, ob.L2, xb, nlist=256, M=32)
    finish_ivfpq(port, ix)
    g = GpuIndex.from_data(ix, device=0)
    for rt in (1, 3):
        rows = _store(rt, xb, chunks=4)
        tr = rows.trained() if rt == 3 else None
        codes = rows.codes()
        _, Ib = g.search(xq, kb, nprobe)
        D, I = g.search_refine_rows(rows, xq, k, kb, nprobe)
        assert all(set(I[q][I[q] >= 0]) <= set(Ib[q]) for q in range(0, nq, 37))
        assert (np.diff(D, axis=1) >= 0).all()
        dec = port.rows_decode(rt, d, codes[I[:50].ravel()], tr).reshape(50, k, d)
        ref_d = ((xq[:50, None, :].astype(np.float64) - dec) ** 2).sum(-1)
        assert np.allclose(D[:50], ref_d, rtol=1e-4)
        Do, Io = port.refine_rows(ob.L2, rt, d, codes, tr, xq[:100], Ib[:100], k)
        assert_parity(Do, Io, D[:100], I[:100], ob.L2, f"row type {rt} at scale")
        rows.close()
    g.close()


def test_sq6_ragged_dimension_and_boundary_values(port):
    """d = 10 (the last group of codes owns two of its three bytes), d = 7; values right below each of the 63 cell
    boundaries (the encoder's product is a double one: tests/test_refine_rows.py); a constant column; rows outside the
    trained range; then a refine over the store with d not a multiple of four"""
    from knowhere_amd import GpuIndex, RowStore
    for d in (10, 7):
        x = gen_data(600, d, 13, -5.0, 5.0)
        x[:, 3] = -2.25
        rows = _store(4, x, chunks=2)
        tr = port.rows_train(x)
        assert rows.trained().tobytes() == tr.tobytes()
        assert rows.codes().tobytes() == port.rows_encode(4, x, tr).tobytes()
        wide = gen_data(60, d, 14, -100.0, 100.0)
        r2 = RowStore(4, d, device=0)
        r2.set_trained(tr)
        r2.add(wide)
        assert r2.codes().tobytes() == port.rows_encode(4, wide, tr).tobytes()
        r2.close()
        rows.close()
    j = np.arange(1, 64, dtype=np.float64)
    below = np.nextafter((j / 63.0).astype(np.float32), np.float32(0))
    col = np.concatenate([[0.0, 1.0], below, (j / 63.0).astype(np.float32)]).astype(np.float32)
    xx = np.zeros((col.size, 4), np.float32)
    xx[:, 0] = col
    rows = _store(4, xx)
    assert rows.codes().tobytes() == port.rows_encode(4, xx, port.rows_train(xx)).tobytes()
    rows.close()
    nb, nq, d = 4000, 40, 10
    xb, xq = gen_data(nb, d, 42, -20.0, 80.0), gen_data(nq, d, 44, -20.0, 80.0)
    for metric in (ob.L2, ob.IP):
        ix = ob.make_index(port, ob.IVF_FLAT, metric, xb, nlist=16)
        g = GpuIndex.from_data(ix, device=0)
        rows = _store(4, xb)
        tr = port.rows_train(xb)
        codes = port.rows_encode(4, xb, tr)
        _, Ib = port.search(ix, xq, 50, 8)
        Do, Io = port.refine_rows(metric, 4, d, codes, tr, xq, Ib, 10)
        D, I = g.search_refine_rows(rows, xq, 10, 50, 8)
        assert_parity(Do, Io, D, I, metric, "sq6 d=10")
        rows.close()
        g.close()


def test_sq4u_quantile_range_at_scale_and_odd_dimension(port):
    """the radix select behind RS_quantiles on 2 x 10^7 values (ranks 2 x 10^5 from either end) against numpy's order
    statistics; d = 7: the last byte of a row holds one code; a constant data set (vdiff = 0) encodes zeros"""
    from knowhere_amd import RowStore
    n, d = 156_250, 128
    x = gen_data(n, d, 21, -3.0, 9.0)
    rows = RowStore(6, d, device=0)
    rows.train_uniform(x, 2, 0.01)
    tr = rows.trained()
    N = n * d
    o = int(np.float32(0.01) * np.float32(N))
    part = np.partition(x.ravel(), (o, N - 1 - o))
    assert tr[0] == part[o] and tr[1] == np.float32(part[N - 1 - o] - part[o])
    rows.close()
    for dd in (7, 1):
        y = gen_data(300, dd, 22, -1.0, 1.0)
        r2 = _store(6, y, chunks=2, metric=ob.IP)
        t2 = port.rows_train_uniform(ob.IP, y)
        assert r2.trained().tobytes() == t2.tobytes()
        assert r2.codes().shape[1] == (dd + 1) // 2 and r2.codes().tobytes() == port.rows_encode(6, y, t2).tobytes()
        r2.close()
    z = np.full((50, 6), 2.5, np.float32)
    r3 = _store(6, z)
    assert r3.trained()[1] == 0 and not r3.codes().any()
    r3.close()
