"""tests/hipemu/run_api.py -- one emulated end-to-end search through the product's own ctypes harness and C ABI.

Run as a subprocess by tests/test_pqf_emulated.py with KNHIP_LIB = the emulated library (the binding reads it at import),
KNHIP_COARSE=exact (the MFMA coarse prefilter is not emulated) and, for the IVF-PQ prefilter, KNHIP_PQF=1.
usage: python run_api.py <case>     prints "OK <case> ..." or raises"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from conftest import gen_data  # noqa: E402
from oracle import binding as ob  # noqa: E402
from helpers import finish_ivfpq  # noqa: E402


def same(Do, Io, D, I, what):
    assert np.array_equal(I, Io), f"{what}: ids differ\n{I}\n{Io}"
    assert np.array_equal(D.view(np.uint32), Do.view(np.uint32)), f"{what}: distances differ"


def main():
    case = sys.argv[1]
    assert os.environ.get("KNHIP_LIB", "").endswith("libknhip_emu.so")
    from knowhere_amd import GpuIndex
    port = ob.Port()
    if case in ("pqf_l2", "pqf_ip"):
        metric = ob.L2 if case == "pqf_l2" else ob.IP
        assert os.environ.get("KNHIP_PQF") == "1"
        nb, d, nlist, nq = 1800, 128, 5, 10
        xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
        ix = ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=nlist, M=32)
        g = GpuIndex.from_data(ix, device=0)
        g.profile_enable(True)
        for k, nprobe in ((10, 3), (4, nlist)):
            g.profile_reset()
            Do, Io = port.search(ix, xq, k, nprobe)
            D, I = g.search(xq, k, nprobe)
            p = g.profile_get()
            same(Do, Io, D, I, f"{case} k={k} nprobe={nprobe}")
            assert p["mscan_queries"] == nq and p["mscan_overflow_queries"] == 0, p
            want = {"h": 1, "i": 2, "d": 3}.get(os.environ.get("KNHIP_PQF_FORM", " ")[0])
            assert want is None or p["pq_filter_form"] == want, (p["pq_filter_form"], want)
            # (a wrong operand layout or scale would flood the candidate lists -- or pass nothing and fail above)
            assert p["mscan_candidates"] < 60 * nq, p
            print(f"  form {p['pq_filter_form']} k={k} nprobe={nprobe}: {p['mscan_candidates'] / nq:.1f} candidates per query")
        bs = np.packbits(np.random.default_rng(3).random(nb) < 0.4, bitorder="little")
        Do, Io = port.search(ix, xq, 10, 3, bs, nb)
        D, I = g.search(xq, 10, 3, bs, nb)
        same(Do, Io, D, I, f"{case} bitset")
        g.close()
    elif case == "pqd_wide":
        # the decode form's wide units: 130 queries on 3 lists = one unit of 128 pairs (four query tiles) + one of 2 per
        # list, tiles per wave odd and even (lists of ~600 rows: 19 tiles, the last one ragged); 70 queries: three tiles
        assert os.environ.get("KNHIP_PQF") == "1" and os.environ.get("KNHIP_PQF_FORM", "")[:1] == "d"
        nb, d, nlist = 1800, 128, 3
        xb = gen_data(nb, d, 42)
        for metric, nq in ((ob.L2, 130), (ob.IP, 70)):
            xq = gen_data(nq, d, 44)
            ix = ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=nlist, M=32)
            g = GpuIndex.from_data(ix, device=0)
            g.profile_enable(True)
            g.profile_reset()
            bs = np.packbits(np.random.default_rng(3).random(nb) < 0.3, bitorder="little")
            Do, Io = port.search(ix, xq, 10, nlist, bs, nb)
            D, I = g.search(xq, 10, nlist, bs, nb)
            p = g.profile_get()
            same(Do, Io, D, I, f"{case} metric={metric} nq={nq}")
            assert p["pq_filter_form"] == 3 and p["mscan_queries"] == nq and p["mscan_overflow_queries"] == 0, p
            assert p["mscan_candidates"] < 80 * nq, p
            print(f"  metric {metric} nq={nq}: {p['mscan_candidates'] / nq:.1f} candidates per query")
            g.close()
    elif case in ("ms_flat_l2", "ms_sq8_ip", "ms_sq8_l2", "ms_flat_ip"):
        # the MFMA paths (hardware-validated) under the matrix-core emulation: coarse GEMM prefilter + re-rank +
        # certificate, fp32 / f16 list prefilter + exact finish -- a regression net for changes made without a GPU
        assert os.environ.get("KNHIP_MSCAN") == "1" and os.environ.get("KNHIP_COARSE") is None
        kind = ob.IVF_FLAT if "flat" in case else ob.IVF_SQ8
        metric = ob.L2 if case.endswith("l2") else ob.IP
        nb, d, nlist, nq = 2400, 48, 36, 40
        xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
        ix = ob.make_index(port, kind, metric, xb, nlist=nlist)
        g = GpuIndex.from_data(ix, device=0)
        g.profile_enable(True)
        Do, Io = port.search(ix, xq, 10, 8)
        D, I = g.search(xq, 10, 8)
        p = g.profile_get()
        same(Do, Io, D, I, case)
        assert p["mscan_queries"] == nq and p["coarse_fallback_queries"] == 0, p
        assert p["mscan_candidates"] < 40 * nq, p  # (a wrong operand layout would flood the candidate lists)
        g.close()
    elif case == "pq_any":
        # numbers of sub-quantizers without a fast kernel (pq_scan_any.hip): m = 12 (word loads, precomputed tables, bitset,
        # range search), m = 3 (byte loads, residual tables), m = 6 (inner product, boundary ties).  Small on purpose: an
        # emulated search takes ~15 s; the GPU suite sweeps the widths (tests/test_gpu_pq_any.py)
        nq, nb = 4, 600
        for M, d, nlist, metric, resid in ((12, 48, 4, ob.L2, False), (3, 24, 3, ob.L2, True)):
            xb, xq = gen_data(nb, d, 42 + M), gen_data(nq, d, 44)
            ix = ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=nlist, M=M)
            if resid:
                ix.use_precomputed_table = 0
                ix.precomputed_table = None
            else:
                finish_ivfpq(port, ix)
            g = GpuIndex.from_data(ix, device=0, precomputed_table_max_bytes=1024 if resid else 0)
            Do, Io = port.search(ix, xq, 10, 3)
            D, I = g.search(xq, 10, 3)
            same(Do, Io, D, I, f"pq_any M={M} resid={resid}")
            if M == 12:
                bs = np.packbits(np.random.default_rng(3).random(nb) < 0.4, bitorder="little")
                Do, Io = port.search(ix, xq, 70, nlist, bs, nb)
                D, I = g.search(xq, 70, nlist, bs, nb)
                same(Do, Io, D, I, f"pq_any M={M} k=70 bitset")
                D40, _ = port.search(ix, xq, 40, nlist)
                radius = float(np.median(D40[:, 20]))
                exp = port.range_search(ix, xq, radius, 2)
                got = g.range_search(xq, np.float32(radius), 2)
                assert np.array_equal(exp[0], got[0]) and np.array_equal(exp[1], got[1])
                assert np.array_equal(exp[2].view(np.uint32), got[2].view(np.uint32))
            g.close()
        # duplicated rows: more candidates at the k-th distance than places (the reference's first-come admission)
        rng = np.random.default_rng(7)
        proto = (rng.integers(-3, 4, (20, 24)) * 7.0).astype(np.float32)
        xb = np.ascontiguousarray(proto[rng.integers(0, 20, nb)])
        xq = np.ascontiguousarray(proto[rng.integers(0, 20, nq)] + rng.integers(0, 2, (nq, 24)).astype(np.float32))
        ix = finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, ob.IP, xb, nlist=4, M=6))
        g = GpuIndex.from_data(ix, device=0)
        g.profile_enable(True)
        g.profile_reset()
        Do, Io = port.search(ix, xq, 7, 3)
        D, I = g.search(xq, 7, 3)
        same(Do, Io, D, I, "pq_any ties")
        assert g.profile_get()["tie_queries"] > 0
        Do, Io = port.search(ix, xq, 70, 3)  # k > 64: the block selection, its boundary tied (the second bisection)
        D, I = g.search(xq, 70, 3)
        same(Do, Io, D, I, "pq_any ties, block selection")
        g.close()
    elif case == "refine_rows":
        # quantised refine stores: train / encode / append on the device == the oracle's restatement (pinned against
        # IndexScalarQuantizer), and knhip_search_refine_rows == IndexRefine over it
        from knowhere_amd import RowStore
        nb, nlist, nq = 320, 4, 3
        # d = 24: sq8 rows are not a multiple of 16 bytes (element loads); d = 32: every row type takes the 16-byte loads
        # d = 21: a ragged last group of 6-bit codes (sq6: 16 bytes per row) and a last byte with one 4-bit code (sq4u);
        # not a multiple of four: the fp32 refine takes its lane = row path there
        for metric, d, types in ((ob.L2, 24, (1, 4, 5, 6)), (ob.IP, 32, (2, 3, 4)), (ob.IP, 21, (4, 6))):
            xb, xq = gen_data(nb, d, 42, -40.0, 60.0), gen_data(nq, d, 44, -40.0, 60.0)
            xb[5, :4] = [1 + 2.0 ** -11, 2.0 ** -25, 65520.0, -(1 + 3 * 2.0 ** -11)]  # fp16 ties / subnormal / overflow
            ix = ob.make_index(port, ob.IVF_SQ8, metric, xb, nlist=nlist)
            g = GpuIndex.from_data(ix, device=0)
            # the fp32 store (IndexRefineFlat): rows gathered by the wave together through the LDS staging area (refine.hip;
            # d = 24 / 32: one ragged chunk of pieces, d = 22: not a multiple of four -> the lane = row path)
            from knowhere_amd import index as kidx
            raw = kidx.GpuIndex(kidx.BRUTE_FORCE, metric, d, device=0)
            raw.add_vectors(xb)
            for k, kb, nprobe in ((10, 70, 4),):  # (70 candidates: a second, partly filled round of 64)
                _, Ib = port.search(ix, xq, kb, nprobe)
                Do, Io = port.refine(metric, xb, xq, Ib, k)
                D, I = g.search_refine(raw, xq, k, kb, nprobe)
                same(Do, Io, D, I, f"fp32 refine metric={metric} d={d} k={k} k_base={kb}")
            raw.close()
            for rt in types:
                rows = RowStore(rt, d, device=0)
                tr = port.rows_train_uniform(metric, xb) if rt == 6 else (port.rows_train(xb) if rt in (3, 4) else None)
                codes = port.rows_encode(rt, xb, tr)
                if rt == 3:  # (build.hip -- column ranges, sq8 encoder -- is not part of the emulated library: GPU tests)
                    rows.set_trained(tr)
                    assert rows.trained().tobytes() == tr.tobytes()
                    rows.add_codes(codes[:200])
                    rows.add_codes(codes[200:])
                else:
                    if rt == 4:  # (ranges from the oracle: the column min / max kernel lives in build.hip; the encoder runs here)
                        rows.set_trained(tr)
                    elif rt == 6:  # (the quantile range by the device's radix select for L2; min / max needs build.hip)
                        if metric == ob.L2:
                            rows.train_uniform(xb, 2, 0.01)
                            assert rows.trained().tobytes() == tr.tobytes(), 'sq4u quantile range'
                        else:
                            rows.set_trained(tr)
                    else:
                        rows.train(xb)
                    rows.add(xb[:200])
                    rows.add(xb[200:])
                assert rows.count() == nb and rows.codes().tobytes() == codes.tobytes(), f"row type {rt}: code bytes"
                for k, kb, nprobe in ((5, 20, 3),) + (((10, 10, 4),) if rt == 4 else ()):  # (k_base == k once: the sq6 store)
                    _, Ib = port.search(ix, xq, kb, nprobe)
                    Do, Io = port.refine_rows(metric, rt, d, codes, tr, xq, Ib, k)
                    D, I = g.search_refine_rows(rows, xq, k, kb, nprobe)
                    same(Do, Io, D, I, f"refine_rows metric={metric} type={rt} k={k}")
                # a store filled from code bytes (Deserialize) behaves the same
                r2 = RowStore(rt, d, device=0)
                if rt in (3, 4, 6):
                    r2.set_trained(tr)
                r2.add_codes(codes)
                if rt in (2, 6) and d != 21:  # (a 16-bit and a ranged store, once each; the sq8 store above was itself filled from code bytes)
                    D1, I1 = g.search_refine_rows(rows, xq, 5, 20, 3)
                    D2, I2 = g.search_refine_rows(r2, xq, 5, 20, 3)
                    same(D1, I1, D2, I2, "store from codes")
                r2.close()
                rows.close()
            g.close()
    elif case == "limits":
        # nprobe above what the LDS sorts (the global-scratch row selection) through the whole search path
        nb, d, nlist, nq = 9000, 8, 4500, 3
        xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
        ix = ob.make_index(port, ob.IVF_FLAT, ob.L2, xb, nlist=nlist)
        g = GpuIndex.from_data(ix, device=0)
        Do, Io = port.search(ix, xq, 5, 4200)
        D, I = g.search(xq, 5, 4200)
        same(Do, Io, D, I, "nprobe=4200")
        g.close()
    elif case == "range_pq16":
        nb, d, nlist, nq = 1500, 128, 6, 5
        xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
        for metric in (ob.L2, ob.IP):
            ix = ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=nlist, M=16)
            g = GpuIndex.from_data(ix, device=0)
            D, _ = port.search(ix, xq, 40, nlist)
            radius = float(np.median(D[:, 20]))
            for max_empty in (0, 2):
                exp = port.range_search(ix, xq, radius, max_empty)
                got = g.range_search(xq, np.float32(radius), max_empty)
                assert np.array_equal(exp[0], got[0]) and np.array_equal(exp[1], got[1])
                assert np.array_equal(exp[2].view(np.uint32), got[2].view(np.uint32))
            g.close()
    elif case == "range_waves":
        # nlist > 128 with an early stop: the probes go in waves of coarse ranks (range.hip); IVF-Flat and IVF-SQ8
        nb, d, nlist, nq = 2500, 8, 200, 4
        xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
        full = os.environ.get("KNHIP_TEST_EMU_FULL") == "1"  # (the default run: one kind, two stop settings -- ~50 s)
        for kind in ((ob.IVF_FLAT, ob.IVF_SQ8) if full else (ob.IVF_FLAT,)):
            ix = ob.make_index(port, kind, ob.L2, xb, nlist=nlist)
            g = GpuIndex.from_data(ix, device=0)
            D, _ = port.search(ix, xq, 40, nlist)
            radius = float(np.median(D[:, 20]))
            for max_empty in ((1, 3, 0) if full else (3, 0)):
                exp = port.range_search(ix, xq, radius, max_empty)
                got = g.range_search(xq, np.float32(radius), max_empty)
                assert np.array_equal(exp[0], got[0]) and np.array_equal(exp[1], got[1]), (kind, max_empty)
                assert np.array_equal(exp[2].view(np.uint32), got[2].view(np.uint32))
                ranks = g.last_range_ranks()
                assert ranks == nlist if max_empty == 0 else ranks <= nlist, (kind, max_empty, ranks)
                assert exp[0][-1] > 0
            g.close()
    elif case in ("ties_flat", "ties_ivfflat", "ties_ivfsq8", "ties_ivfpq"):
        # duplicates of a few rows: more candidates at the k-th distance than places -> the reference's admission rule
        # (knhip_api.hip, search_batch_ties: k + 1 results, detect, dump pass in scan order, closed form of the heap)
        rng = np.random.default_rng(17)
        d = 32
        proto = (rng.integers(-3, 4, (30, d)) * 7.0).astype(np.float32)
        xb = np.ascontiguousarray(proto[rng.integers(0, 30, 1500)])
        xq = np.ascontiguousarray((proto[rng.integers(0, 30, 12)] + rng.integers(0, 2, (12, d))).astype(np.float32))
        kind = {"ties_flat": ob.FLAT, "ties_ivfflat": ob.IVF_FLAT, "ties_ivfsq8": ob.IVF_SQ8, "ties_ivfpq": ob.IVF_PQ}[case]
        for metric in (ob.L2, ob.IP):
            ix = ob.make_index(port, kind, metric, xb, nlist=8, M=8)
            g = GpuIndex.from_data(ix, device=0)
            g.profile_enable(True)
            g.profile_reset()
            bs = np.packbits(np.random.default_rng(3).random(len(xb)) < 0.3, bitorder="little")
            for k in (1, 6, 25):
                for b, nb_ in ((None, 0), (bs, len(xb))):
                    Do, Io = port.search(ix, xq, k, 4, b, nb_)
                    D, I = g.search(xq, k, 4, b, nb_)
                    same(Do, Io, D, I, f"{case} metric={metric} k={k} bitset={b is not None}")
            assert g.profile_get()["tie_queries"] > 0
            g.close()
    else:
        raise SystemExit(f"unknown case {case}")
    print("OK", case)


if __name__ == "__main__":
    main()
