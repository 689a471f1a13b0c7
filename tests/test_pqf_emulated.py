"""The SOURCE of the half-precision ADC prefilter kernels (knowhere_amd/csrc/pq_filter.hip) and of the sample-plan / tau /
exact-finish kernels of mfma_scan.hip, executed on the CPU through the host stand-in of tests/hipemu (one OS thread
per GPU thread, barriers for __syncthreads and for the cross-lane operations), against the oracle.

Why: the kernels were written when no GPU was at hand.  The numpy model (tests/test_pqf_model.py) pins the layouts and
formulas; this test runs the kernel files themselves -- their index arithmetic, the persistent unit protocol (per-XCD
counters, mailbox, barriers), the per-pair constants, window loop, epilogues, candidate lists and the exact finish --
with only the hardware-specific lines rewritten (tests/hipemu/emu_build.py lists them: LDS byte-offset addressing, the
SDWA shift, dynamic LDS declarations).  It cannot see hardware hazards; it does catch wrong indices, protocol
deadlocks and wrong arithmetic.  Results must equal the oracle's bit for bit for every query that did not overflow its
candidate list."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import gen_data
from oracle import binding as ob

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"))

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="needs the host clang++")

M, KSUB, DSUB = 32, 256, 4


@pytest.fixture(scope="module")
def emu():
    import emu_build
    lib = C.CDLL(emu_build.build())
    lib.emu_pqf_search.restype = C.c_int
    return lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def run_emulated(lib, port, ix, xq, k, nprobe, cap=4096, bitset=None, use_hist=1, retry=0):
    is_l2 = ix.metric == ob.L2
    nq = xq.shape[0]
    cdis, keys = port.coarse_search(ix, xq, nprobe)
    lens = np.array([len(c) for c in ix.list_codes], np.int64)
    row_off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    codes = np.ascontiguousarray(np.concatenate([c.reshape(-1, M) for c in ix.list_codes]).astype(np.uint8))
    ids = np.ascontiguousarray(np.concatenate(ix.list_ids).astype(np.int64))
    pre = np.ascontiguousarray(ix.precomputed_table, np.float32) if is_l2 else None
    cb = np.ascontiguousarray(ix.pq_centroids, np.float32)
    cen = np.ascontiguousarray(ix.centroids, np.float32)
    xq = np.ascontiguousarray(xq, np.float32)
    keys = np.ascontiguousarray(keys, np.int64)
    cdis = np.ascontiguousarray(cdis, np.float32)
    D = np.zeros((nq, k), np.float32)
    I = np.zeros((nq, k), np.int64)
    cnt = np.zeros(nq, np.int32)
    ovf = np.zeros(nq, np.int32)
    tau = np.zeros(nq, np.float32)
    nunits = np.zeros(1, np.int64)
    nbits = 0 if bitset is None else int(lens.sum())
    rc = lib.emu_pqf_search(C.c_int64(ix.nlist), _p(lens, C.c_int64), _p(row_off, C.c_int64), _p(codes, C.c_uint8),
                            _p(ids, C.c_int64), _p(pre, C.c_float), _p(cb, C.c_float), _p(cen, C.c_float),
                            _p(xq, C.c_float), C.c_int64(nq), C.c_int(nprobe), _p(keys, C.c_int64), _p(cdis, C.c_float),
                            C.c_int(k), C.c_int(1 if is_l2 else 0), C.c_int(cap), _p(bitset, C.c_uint8), C.c_int64(nbits),
                            C.c_int(use_hist), C.c_int(retry), _p(D, C.c_float), _p(I, C.c_int64), _p(cnt, C.c_int32), _p(ovf, C.c_int32),
                            _p(tau, C.c_float), _p(nunits, C.c_int64))
    assert rc == 0, f"emulated pipeline failed at stage {rc}"
    return D, I, cnt, ovf, tau, int(nunits[0])


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_emulated_kernels_return_the_oracles_bits(emu, port, metric):
    nb, d, nlist, nq, k, nprobe = 2600, 128, 5, 11, 10, 3  # 11 queries: units of 8 and ragged ones, two "CUs"
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    ix = ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=nlist, M=32)
    Do, Io = port.search(ix, xq, k, nprobe)
    D, I, cnt, ovf, tau, nunits = run_emulated(emu, port, ix, xq, k, nprobe)
    assert nunits >= nlist - 1
    assert not ovf.any(), (ovf, cnt)
    assert np.array_equal(I, Io), (I, Io)
    assert np.array_equal(D.view(np.uint32), Do.view(np.uint32))
    scanned = sum(len(c) for c in ix.list_codes) * nprobe / nlist
    assert 0 < cnt.max() < 0.6 * scanned, (cnt, scanned)


@pytest.mark.timeout(1500)
def test_emulated_bitset_and_overflow_flags(emu, port):
    nb, d, nlist, nq, k, nprobe = 2000, 128, 4, 6, 5, 2
    xb, xq = gen_data(nb, d, 52), gen_data(nq, d, 54)
    ix = ob.make_index(port, ob.IVF_PQ, ob.L2, xb, nlist=nlist, M=32)
    filt = np.random.default_rng(3).random(nb) < 0.4
    bs = np.packbits(filt, bitorder="little")
    Do, Io = port.search(ix, xq, k, nprobe, bs, nb)
    D, I, cnt, ovf, tau, _ = run_emulated(emu, port, ix, xq, k, nprobe, bitset=bs, use_hist=0)
    assert not ovf.any()
    assert np.array_equal(I, Io) and np.array_equal(D.view(np.uint32), Do.view(np.uint32))
    # a capacity of 8 candidates: a query that gathers more is flagged 1 by the filter kernel; the finish kernel's first
    # pass then prepares its RETRY (flag 2, candidate list emptied, bound = exact k-th of the 8 gathered rows -- the
    # retry and exact rounds themselves belong to the product's orchestration, not to this harness); the queries that
    # stayed within the capacity still equal the oracle
    D2, I2, cnt2, ovf2, _, _ = run_emulated(emu, port, ix, xq, k, nprobe, cap=8, bitset=bs, use_hist=0)
    assert (ovf2 == 2).sum() >= 1 and set(ovf2.tolist()) <= {0, 2}, ovf2
    assert (cnt2[ovf2 == 2] == 0).all()
    ok = ovf2 == 0
    assert np.array_equal(I2[ok], Io[ok])
    # ... and with the retry round (one-query units under the tightened bound, second finish pass): a capacity of 64
    # overflows in the first round and fits in the second -- every query ends with the oracle's result
    D3, I3, cnt3, ovf3, _, _ = run_emulated(emu, port, ix, xq, k, nprobe, cap=64, bitset=bs, use_hist=0, retry=1)
    assert not ovf3.any(), (ovf3, cnt3)
    assert np.array_equal(I3, Io) and np.array_equal(D3.view(np.uint32), Do.view(np.uint32))


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_emulated_ragged_lists_and_k(emu, port, metric):
    """an empty list, a list shorter than one group of 64, one of exactly 128 rows, ids that are not row numbers;
    k = 1 and k above a wave; every list probed"""
    nb, d, nlist, nq = 1500, 128, 6, 9
    xb, xq = gen_data(nb, d, 62), gen_data(nq, d, 64)
    ids = np.random.default_rng(5).permutation(nb).astype(np.int64) * 3 + 1
    ix = ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=nlist, M=32, ids=ids)
    ix.list_codes[1], ix.list_ids[1] = ix.list_codes[1][:0], ix.list_ids[1][:0]
    ix.list_codes[2], ix.list_ids[2] = ix.list_codes[2][:17], ix.list_ids[2][:17]
    ix.list_codes[3], ix.list_ids[3] = ix.list_codes[3][:128], ix.list_ids[3][:128]
    for k, nprobe in ((1, 2), (70, nlist), (10, 4)):
        Do, Io = port.search(ix, xq, k, nprobe)
        D, I, cnt, ovf, tau, _ = run_emulated(emu, port, ix, xq, k, nprobe)
        ok = ovf == 0
        assert ok.all(), (k, nprobe, ovf, cnt)
        assert np.array_equal(I, Io), (k, nprobe)
        assert np.array_equal(D.view(np.uint32), Do.view(np.uint32)), (k, nprobe)


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("is_l2", [True, False], ids=["l2", "ip"])
def test_emulated_row_select_above_4096_keys(emu, is_l2):
    """topk.hip::row_select_kernel with k up to 16384 (the sort of the selected keys fills 128 KB of LDS) and, sorted in
    a global scratch row, up to 65536: the shapes behind nprobe > 4096 and range search on more than 4096 lists.
    Canonical order with ties on the value."""
    emu.emu_row_select.restype = C.c_int
    rng = np.random.default_rng(9)
    for n, k in ((6000, 5000), (16384, 16384), (20000, 4097), (300, 10), (40000, 16385), (65536, 65536)):
        vals = rng.standard_normal((2, n)).astype(np.float32)
        vals[0, ::7] = vals[0, 3]  # ties on the value: broken by the column index
        keys = np.zeros((2, k), np.int64)
        dist = np.zeros((2, k), np.float32)
        rc = emu.emu_row_select(_p(vals, C.c_float), C.c_int64(2), C.c_int64(n), C.c_int(k), C.c_int(1 if is_l2 else 0),
                                _p(keys, C.c_int64), _p(dist, C.c_float))
        assert rc == 0
        for r in range(2):
            idx = np.arange(n)
            order = np.lexsort((idx, vals[r])) if is_l2 else np.lexsort((-idx, -vals[r]))
            kk = min(k, n)
            assert np.array_equal(keys[r, :kk], order[:kk]), (n, k, r)
            assert np.array_equal(dist[r, :kk].view(np.uint32), vals[r][order[:kk]].view(np.uint32))


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("M", [8, 16, 64])
def test_emulated_pq_adc_dump_any_code_width(emu, port, M):
    """range.hip::pq_adc_dump_kernel (range search on IVF-PQ indexes whose code width has no dump mode in the fast ADC
    kernels): every distance it writes equals, bit for bit, what the oracle's range search reports for that row -- L2
    with the precomputed table, L2 with residual tables, inner product"""
    emu.emu_pq_adc_dump.restype = C.c_int
    nb, d, nlist, nq = 900, 128, 5, 4
    xb, xq = gen_data(nb, d, 72), gen_data(nq, d, 74)
    for metric, residual in ((ob.L2, False), (ob.L2, True), (ob.IP, False)):
        ix = ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=nlist, M=M)
        mode = 1 if metric == ob.IP else (2 if residual else 0)  # PqLutMode
        pre = None
        if metric == ob.L2 and not residual:
            pre = np.ascontiguousarray(ix.precomputed_table, np.float32)
        if residual:
            ix.use_precomputed_table = 0
            ix.precomputed_table = None
        cdis, keys = port.coarse_search(ix, xq, nlist)
        lens = np.array([len(c) for c in ix.list_codes], np.int64)
        row_off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        codes = np.ascontiguousarray(np.concatenate([c.reshape(-1, M) for c in ix.list_codes]).astype(np.uint8))
        ids = np.concatenate(ix.list_ids)
        cb = np.ascontiguousarray(ix.pq_centroids, np.float32)
        cen = np.ascontiguousarray(ix.centroids, np.float32)
        keys = np.ascontiguousarray(keys, np.int64)
        cdis = np.ascontiguousarray(cdis, np.float32)
        dist = np.full((nq, nb), np.nan, np.float32)
        rc = emu.emu_pq_adc_dump(C.c_int64(nlist), _p(lens, C.c_int64), _p(row_off, C.c_int64), _p(codes, C.c_uint8),
                                 C.c_int(M), C.c_int(d), C.c_int(mode), _p(pre, C.c_float), _p(cb, C.c_float),
                                 _p(cen, C.c_float), _p(np.ascontiguousarray(xq), C.c_float), C.c_int64(nq), C.c_int(nlist),
                                 _p(keys, C.c_int64), _p(cdis, C.c_float), C.c_int64(nb), _p(dist, C.c_float))
        assert rc == 0
        assert not np.isnan(dist).any()
        # the oracle's range search with an all-embracing radius reports every row: lists in coarse order, storage
        # order inside a list
        radius = np.float32(3.0e38) if metric == ob.L2 else np.float32(-3.0e38)
        lims, rid, rdis = port.range_search(ix, xq, radius, 0)
        for q in range(nq):
            exp_i, exp_d = rid[lims[q]:lims[q + 1]], rdis[lims[q]:lims[q + 1]]
            got_i, got_d = [], []
            for l in keys[q]:
                got_i.append(ids[row_off[l]:row_off[l] + lens[l]])
                got_d.append(dist[q, row_off[l]:row_off[l] + lens[l]])
            got_i, got_d = np.concatenate(got_i), np.concatenate(got_d)
            assert np.array_equal(got_i, exp_i), (M, metric, residual, q)
            assert np.array_equal(got_d.view(np.uint32), exp_d.view(np.uint32)), (M, metric, residual, q)
