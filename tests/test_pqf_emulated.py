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
    poor = np.zeros(1, np.int32)
    nbits = 0 if bitset is None else int(lens.sum())
    rc = lib.emu_pqf_search(C.c_int64(ix.nlist), _p(lens, C.c_int64), _p(row_off, C.c_int64), _p(codes, C.c_uint8),
                            _p(ids, C.c_int64), _p(pre, C.c_float), _p(cb, C.c_float), _p(cen, C.c_float),
                            _p(xq, C.c_float), C.c_int64(nq), C.c_int(nprobe), _p(keys, C.c_int64), _p(cdis, C.c_float),
                            C.c_int(k), C.c_int(1 if is_l2 else 0), C.c_int(cap), _p(bitset, C.c_uint8), C.c_int64(nbits),
                            C.c_int(use_hist), C.c_int(retry), _p(D, C.c_float), _p(I, C.c_int64), _p(cnt, C.c_int32), _p(ovf, C.c_int32),
                            _p(tau, C.c_float), _p(nunits, C.c_int64), _p(poor, C.c_int32))
    assert rc == 0, f"emulated pipeline failed at stage {rc}"
    run_emulated.poor = int(poor[0])
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
    # the selectivity guard: with room for 4096 candidates no query is predicted to need more than half of it; with
    # room for 16 every one is (the prediction, sample share x rows probed, is within a factor 2 of what was gathered)
    assert run_emulated.poor == 0
    run_emulated(emu, port, ix, xq, k, nprobe, cap=16)
    assert run_emulated.poor == nq, run_emulated.poor


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
    # a small capacity: a query that gathers more is flagged 1 by the filter kernel; the finish kernel's first
    # pass then prepares its RETRY (flag 2, candidate list emptied, bound = exact k-th of the 8 gathered rows -- the
    # retry and exact rounds themselves belong to the product's orchestration, not to this harness); the queries that
    # stayed within the capacity still equal the oracle
    # (the capacity is taken one below the largest candidate count of the first run: the matrix-core filter passes little
    # more than k rows per query)
    small = int(cnt.max()) - 1
    assert small >= k
    D2, I2, cnt2, ovf2, _, _ = run_emulated(emu, port, ix, xq, k, nprobe, cap=small, bitset=bs, use_hist=0)
    assert (ovf2 == 2).sum() >= 1 and set(ovf2.tolist()) <= {0, 2}, ovf2
    assert (cnt2[ovf2 == 2] == 0).all()
    ok = ovf2 == 0
    assert np.array_equal(I2[ok], Io[ok])
    # ... and with the retry round (one-query units under the tightened bound, second finish pass): the small capacity
    # overflows in the first round; in the second the bound is the exact k-th of whichever 64 rows got in first (thread
    # timing), so a query may overflow again (flag 1: the product's exact round) -- every query that finished here ends
    # with the oracle's result, and most do
    D3, I3, cnt3, ovf3, _, _ = run_emulated(emu, port, ix, xq, k, nprobe, cap=small, bitset=bs, use_hist=0, retry=1)
    ok3 = ovf3 == 0
    assert set(ovf3.tolist()) <= {0, 1} and ok3.sum() >= nq - 2, (ovf3, cnt3)
    assert np.array_equal(I3[ok3], Io[ok3]) and np.array_equal(D3[ok3].view(np.uint32), Do[ok3].view(np.uint32))


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
    for k, nprobe in (((1, 2), (70, nlist)) if metric == ob.L2 else ((10, 4),)):
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
    shapes = ((6000, 5000), (300, 10), (40000, 16385)) if is_l2 else ((9000, 4097), (33000, 32769))
    for n, k in shapes:
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


def _run_api_case(case, **env):
    """one end-to-end search through the product's ctypes harness, C ABI and orchestration (knhip_api.hip) on the
    emulated library, in a subprocess (the binding reads KNHIP_LIB at import)"""
    import subprocess
    import emu_build
    e = dict(os.environ)
    e.update({"KNHIP_LIB": emu_build.build_api(), "KNHIP_COARSE": "exact"})  # (exact coarse stage: less to emulate)
    e.update(env)
    e = {k: v for k, v in e.items() if v is not None}
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(emu_build.__file__), "run_api.py"), case], env=e,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and f"OK {case}" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("form", ["half", "int8", "decode"])
@pytest.mark.parametrize("case", ["pqf_l2", pytest.param("pqf_ip", marks=pytest.mark.skipif(
    os.environ.get("KNHIP_TEST_EMU_FULL") != "1", reason="KNHIP_TEST_EMU_FULL=1"))])
def test_emulated_api_ivfpq_prefilter(case, form):
    """KNHIP_PQF=1 through knhip_index_* / knhip_search: layouts built on first use, exact coarse stage, work table with
    the sample split, sample pass, row selection, tau, filter, finish passes, (empty) retry and exact rounds, merge --
    every query finished by the prefilter path, results equal to the oracle's bit for bit, with and without a bitset"""
    _run_api_case(case, KNHIP_PQF="1", KNHIP_PQF_FORM=form)


@pytest.mark.timeout(1800)
@pytest.mark.skipif(os.environ.get("KNHIP_TEST_EMU_FULL") != "1", reason="10 minutes of emulation (KNHIP_TEST_EMU_FULL=1); "
                    "the kernels it exercises have their own quick tests above and below")
def test_emulated_api_nprobe_above_4096():
    """IVF-Flat with nprobe = 4200 of 4500 lists: the larger row selection AND the partial-list merge over more than
    4096 slots (merge_partials_big_kernel: the emulation found the 64 x 64 slot limit of the original kernel's
    exhausted-slot mask)"""
    _run_api_case("limits")


@pytest.mark.timeout(1800)
def test_emulated_api_any_number_of_sub_quantizers():
    """IVF-PQ with m outside {8, 16, 32, 64} (pq_scan_any.hip, the reference takes any m that divides dim): m = 12, 6, 3, 1,
    precomputed / residual / inner-product tables, bitset, boundary ties, range search -- the oracle's results"""
    _run_api_case("pq_any")


@pytest.mark.timeout(1800)
def test_emulated_api_quantised_refine_stores():
    """knhip_rows (fp16 / bf16 / sq8 refine stores): device training, encoding and append give the oracle's bytes, and
    knhip_search_refine_rows the oracle's IndexRefine-over-IndexScalarQuantizer results, both metrics"""
    _run_api_case("refine_rows")


@pytest.mark.timeout(1800)
def test_emulated_api_range_search_pq16():
    _run_api_case("range_pq16")


@pytest.mark.timeout(1800)
def test_emulated_api_range_search_rank_waves():
    """range search with nlist > 128 and an early stop: wave gather, the list-based dump kernels, per-wave counts and the
    early-stop state kernel, all emulated; lims / ids / order / bits equal to the oracle's walk over every rank"""
    _run_api_case("range_waves")


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("is_l2", [True, False], ids=["l2", "ip"])
def test_emulated_merge_of_more_than_4096_partial_lists(emu, is_l2):
    """topk.hip::merge_partials_big_kernel: per query the k best of nslot sentinel-terminated sorted partial lists with
    nslot above 64 x 64 (the slot count the original kernel's per-lane exhausted mask holds), and the original kernel
    just below that limit"""
    emu.emu_merge_partials.restype = C.c_int
    rng = np.random.default_rng(11)
    for nslot, k in ((4200, 4), (4096, 3)):
        nq = 1
        pd = np.zeros((nq, nslot, k), np.float32)
        pi = np.full((nq, nslot, k), -1, np.int64)
        allv = []
        for q in range(nq):
            cand = []
            for s in range(nslot):
                n = int(rng.integers(0, k + 1)) if s % 3 else 0   # a third of the slots are empty
                v = np.sort(rng.standard_normal(n).astype(np.float32))
                if s >= 4096:  # the best entries sit in the slots behind the old limit
                    v = (v - np.float32(10.0)).astype(np.float32)
                if not is_l2:
                    v = (-v).astype(np.float32)
                ids = np.sort(rng.choice(1000, n, replace=False)).astype(np.int64) + s * 1000
                pd[q, s, :n], pi[q, s, :n] = v, ids
                cand += list(zip(v.tolist(), ids.tolist()))
            cand.sort(key=(lambda t: (t[0], t[1])) if is_l2 else (lambda t: (-t[0], -t[1])))
            allv.append(cand[:k])
        out_d = np.zeros((nq, k), np.float32)
        out_i = np.zeros((nq, k), np.int64)
        rc = emu.emu_merge_partials(_p(pd, C.c_float), _p(pi, C.c_int64), C.c_int64(nq), C.c_int(nslot), C.c_int(k),
                                    C.c_int(1 if is_l2 else 0), _p(out_d, C.c_float), _p(out_i, C.c_int64))
        assert rc == 0
        for q in range(nq):
            assert out_i[q].tolist() == [c[1] for c in allv[q]], (nslot, k, q)
            assert np.array_equal(out_d[q], np.array([c[0] for c in allv[q]], np.float32))


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("case", ["ms_flat_l2",
                                  pytest.param("ms_sq8_ip", marks=pytest.mark.skipif(
                                      os.environ.get("KNHIP_TEST_EMU_FULL") != "1", reason="KNHIP_TEST_EMU_FULL=1")),
                                  pytest.param("ms_sq8_l2", marks=pytest.mark.skipif(
                                      os.environ.get("KNHIP_TEST_EMU_FULL") != "1", reason="KNHIP_TEST_EMU_FULL=1")),
                                  pytest.param("ms_flat_ip", marks=pytest.mark.skipif(
                                      os.environ.get("KNHIP_TEST_EMU_FULL") != "1", reason="KNHIP_TEST_EMU_FULL=1"))])
def test_emulated_api_mfma_paths(case):
    """the matrix-core paths with v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_f16 emulated (operand and accumulator
    layouts of hip/hip_runtime.h): coarse GEMM prefilter + exact re-rank + certificate, list prefilter + exact finish.
    These kernels are validated on hardware; the emulated run is the safety net for changes made without a GPU."""
    _run_api_case(case, KNHIP_MSCAN="1", KNHIP_COARSE=None)
