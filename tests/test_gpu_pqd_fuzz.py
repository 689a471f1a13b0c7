"""GPU (-m gpu): randomized shapes through the decode form of the IVF-PQ prefilter (knowhere_amd/csrc/pq_decode.hip: loads in a
hand-managed ring of named registers, one wave per unit, parked records, per-pair reservations) against the exact ADC kernels
on the SAME index -- ids and distance bits equal.  The exact kernels are pinned against the oracle / the reference build in
tests/test_gpu_pqf.py, test_gpu_parity.py and test_gpu_scale_parity.py; this file only asks that no shape -- list lengths from
empty to tens of tiles, 1 .. 4 query tiles per unit, k up to 1000, heavy filters, units that overflow their record regions --
makes the two disagree.  The index is trained and filled on the device (fast), then attached twice."""
import os
import types

import numpy as np
import pytest

from conftest import gen_data

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch


def _clustered(n, d, ncenter, sigma, seed):
    r = np.random.default_rng(seed)
    c = r.random((ncenter, d), dtype=np.float32) * 10.0
    return (c[r.integers(0, ncenter, n)] + sigma * r.standard_normal((n, d), dtype=np.float32)).astype(np.float32)


def _index_pair(monkeypatch, metric, xb, nlist, spill=None, forced=True):
    """one device-built IVF-PQ m = 32 index, attached as (exact kernels, decode-form prefilter -- forced, guard off; or,
    forced=False, whatever the library's own decision logic picks: guard, form, exact kernels)"""
    from knowhere_amd import GpuIndex
    d = xb.shape[1]
    b = GpuIndex(2, metric, d, nlist, 32, 8, device=0)
    b.train(xb)
    b.add(xb)
    sizes, codes, ids = b.get_lists()
    ix = types.SimpleNamespace(kind=2, metric=metric, d=d, nlist=nlist, M=32, nbits=8, centroids=b.get_coarse(),
                               pq_centroids=b.get_pq(), list_codes=[], list_ids=[])
    pos = 0
    for l in range(nlist):
        n = int(sizes[l])
        ix.list_codes.append(codes[pos:pos + n])
        ix.list_ids.append(ids[pos:pos + n])
        pos += n
    b.close()
    monkeypatch.setenv("KNHIP_PQF", "0")
    g0 = GpuIndex.from_data(ix, device=0)
    monkeypatch.delenv("KNHIP_PQF")
    if forced:
        monkeypatch.setenv("KNHIP_PQF", "1")
        monkeypatch.setenv("KNHIP_PQF_GUARD", "0")
        monkeypatch.setenv("KNHIP_PQF_FORM", "decode")
    if spill is not None:
        monkeypatch.setenv("KNHIP_PQD_SPILL", str(spill))
    g1 = GpuIndex.from_data(ix, device=0)
    for v in ("KNHIP_PQF", "KNHIP_PQF_GUARD", "KNHIP_PQF_FORM", "KNHIP_PQD_SPILL"):
        monkeypatch.delenv(v, raising=False)
    return g0, g1


@pytest.mark.parametrize("seed", range(int(os.environ.get("KNHIP_FUZZ_SEEDS", "24"))))  # (KNHIP_FUZZ_SEEDS=400: a longer hunt)
def test_decode_form_equals_the_exact_kernels_on_random_shapes(torch_cuda, monkeypatch, seed):
    r = np.random.default_rng(1000 + seed)
    metric = int(r.integers(0, 2))
    d = 128
    nb = int(r.choice([3000, 20000, 90000, 250000]))
    nlist = int(r.choice([4, 16, 64, 256])) if nb >= 20000 else int(r.choice([4, 16, 40]))
    large = os.environ.get("KNHIP_FUZZ_LARGE") == "1"  # (a one-off hunt at sizes where units span hundreds of tiles)
    if large:
        nb, nlist = int(r.choice([600_000, 2_000_000])), int(r.choice([128, 512, 2048]))
    clustered = bool(r.integers(0, 2))
    xb = _clustered(nb, d, 2000 if large else 200, 0.4, seed) if clustered else gen_data(nb, d, seed, -5.0, 5.0)
    if seed % 3 == 0:
        xb[100:180] = xb[7]  # identical rows: identical codes, ties in the lists
    forced = seed % 4 != 3  # (every fourth shape: the library's own choice of path)
    g0, g1 = _index_pair(monkeypatch, metric, xb, nlist, spill=16 if seed % 6 == 5 else None, forced=forced)
    g1.profile_enable(True)
    ran = 0
    for case in range(5):
        nq = int(r.choice([1, 3, 40, 130, 600]))
        if large:
            nq = int(r.choice([300, 2000, 5000]))
        k = int(r.choice([1, 10, 100, 128, 500, 1000]))
        nprobe = int(min(nlist, r.choice([1, 2, 8, 32, 256])))
        xq = (xb[r.integers(0, nb, nq)] + 0.05 * r.standard_normal((nq, d), dtype=np.float32)).astype(np.float32) \
            if clustered else gen_data(nq, d, 77 + case, -5.0, 5.0)
        frac = float(r.choice([0.0, 0.0, 0.3, 0.9, 0.995]))
        bs = np.packbits(r.random(nb) < frac, bitorder="little") if frac > 0 else None
        g1.profile_reset()
        D1, I1 = g1.search(xq, k, nprobe, bs, nb if bs is not None else 0)
        p = g1.profile_get()
        D0, I0 = g0.search(xq, k, nprobe, bs, nb if bs is not None else 0)
        what = f"seed={seed} case={case} metric={metric} nb={nb} nlist={nlist} nq={nq} k={k} nprobe={nprobe} filter={frac}"
        assert np.array_equal(I0, I1), what + f": {int((I0 != I1).any(1).sum())} queries differ in ids"
        assert np.array_equal(D0.view(np.uint32), D1.view(np.uint32)), what + ": distance bits"
        ran += p["pq_filter_form"] == 3
    assert ran > 0 or not forced, "the decode form never ran"
    g0.close()
    g1.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("KNHIP_FUZZ_SEEDS", "16"))))
def test_row_kind_prefilters_equal_the_exact_kernels_on_random_shapes(torch_cuda, monkeypatch, seed):
    """the same for IVF-Flat (split-bf16 filter, mfma_scan_bf16.hip) and IVF-SQ8 (f16 filter on the code bytes, mfma_scan.hip)
    with their pruning finish: KNHIP_MSCAN=1 (the prefilter whenever the shape allows) against KNHIP_MSCAN=0 (exact row
    kernels) on one device-built index; dimensions off the multiples of 16, k up to 1000, heavy filters"""
    from knowhere_amd import GpuIndex
    r = np.random.default_rng(2000 + seed)
    kind = 1 if seed % 2 == 0 else 3
    metric = int(r.integers(0, 2))
    d = int(r.choice([24, 64, 100, 128, 200]))
    nb = int(r.choice([3000, 30000, 120000]))
    nlist = int(r.choice([4, 16, 64, 200])) if nb >= 30000 else int(r.choice([4, 16, 40]))
    xb = _clustered(nb, d, 150, 0.5, seed) if seed % 3 else gen_data(nb, d, seed, -3.0, 3.0)
    if seed % 4 == 0:
        xb[200:260] = xb[11]
    b = GpuIndex(kind, metric, d, nlist, 0, 8, device=0)
    b.train(xb)
    b.add(xb)
    sizes, codes, ids = b.get_lists()
    ix = types.SimpleNamespace(kind=kind, metric=metric, d=d, nlist=nlist, M=0, nbits=8, centroids=b.get_coarse(),
                               pq_centroids=None, sq_trained=b.get_sq() if kind == 3 else None, list_codes=[], list_ids=[])
    pos = 0
    for l in range(nlist):
        n = int(sizes[l])
        ix.list_codes.append(codes[pos:pos + n])
        ix.list_ids.append(ids[pos:pos + n])
        pos += n
    b.close()
    monkeypatch.setenv("KNHIP_MSCAN", "0")
    g0 = GpuIndex.from_data(ix, device=0)
    monkeypatch.setenv("KNHIP_MSCAN", "1")
    g1 = GpuIndex.from_data(ix, device=0)
    monkeypatch.delenv("KNHIP_MSCAN")
    g1.profile_enable(True)
    ran = 0
    for case in range(5):
        nq = int(r.choice([1, 5, 70, 129, 500]))
        k = int(r.choice([1, 10, 100, 128, 500, 1000]))
        nprobe = int(min(nlist, r.choice([1, 2, 8, 32, 200])))
        xq = (xb[r.integers(0, nb, nq)] + 0.05 * r.standard_normal((nq, d), dtype=np.float32)).astype(np.float32)
        frac = float(r.choice([0.0, 0.0, 0.3, 0.9, 0.995]))
        bs = np.packbits(r.random(nb) < frac, bitorder="little") if frac > 0 else None
        g1.profile_reset()
        D1, I1 = g1.search(xq, k, nprobe, bs, nb if bs is not None else 0)
        p = g1.profile_get()
        D0, I0 = g0.search(xq, k, nprobe, bs, nb if bs is not None else 0)
        what = f"seed={seed} case={case} kind={kind} metric={metric} d={d} nb={nb} nlist={nlist} nq={nq} k={k} nprobe={nprobe} filter={frac}"
        assert np.array_equal(I0, I1), what + f": {int((I0 != I1).any(1).sum())} queries differ in ids"
        assert np.array_equal(D0.view(np.uint32), D1.view(np.uint32)), what + ": distance bits"
        ran += (p["mscan_queries"] + p["mscan_overflow_queries"]) > 0
    assert ran > 0, "the prefilter never ran"
    g0.close()
    g1.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("KNHIP_FUZZ_SEEDS", "10"))))
def test_brute_force_matrix_core_path_equals_the_row_scan_on_random_shapes(torch_cuda, monkeypatch, seed):
    """BRUTE_FORCE: the bf16 prefilter + exact re-rank over chunks of the base (one pass per chunk behind the first, the running
    k-th best as the bound) against the exact row scan (KNHIP_BF=exact): random dimension, row count (1 .. 5 chunks, ragged
    last chunk), batch, k up to 400, both metrics, duplicated rows"""
    from knowhere_amd import GpuIndex
    r = np.random.default_rng(3000 + seed)
    metric = int(r.integers(0, 2))
    d = int(r.choice([16, 48, 96, 128, 256]))
    nb = int(r.choice([40_000, 131_072, 131_073, 300_000, 600_000]))
    nq = int(r.choice([16, 100, 700]))
    if nq * nb < 16_000_000:
        nq = int(16_000_000 // nb + 1)
    xb = _clustered(nb, d, 300, 0.5, seed) if seed % 2 else gen_data(nb, d, seed, -2.0, 2.0)
    xb[nb // 2:nb // 2 + 30] = xb[3]
    xb[nb - 5:] = xb[3]
    xq = np.concatenate([xb[3:4] + 0.001, xb[r.integers(0, nb, nq - 1)] + 0.1 * r.standard_normal((nq - 1, d), dtype=np.float32)]).astype(np.float32)
    g1 = GpuIndex(0, metric, d)
    g1.add_vectors(xb)
    monkeypatch.setenv("KNHIP_BF", "exact")
    g0 = GpuIndex(0, metric, d)
    g0.add_vectors(xb)
    monkeypatch.delenv("KNHIP_BF")
    g1.profile_enable(True)
    for k in (int(r.choice([1, 7, 50])), int(r.choice([99, 100, 400]))):
        g1.profile_reset()
        D1, I1 = g1.search(xq, k)
        assert g1.profile_get()["pq_filter_form"] == 10, "the matrix-core path did not run"
        D0, I0 = g0.search(xq, k)
        what = f"seed={seed} metric={metric} d={d} nb={nb} nq={nq} k={k}"
        assert np.array_equal(I0, I1), what + f": {int((I0 != I1).any(1).sum())} queries differ in ids"
        assert np.array_equal(D0.view(np.uint32), D1.view(np.uint32)), what + ": distance bits"
    g0.close()
    g1.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("KNHIP_FUZZ_SEEDS", "12"))))
def test_ivfpq_any_width_equals_the_oracle_on_random_shapes(torch_cuda, port, seed):
    """IVF-PQ with a random number of sub-quantizers and a random code width (1 .. 8 bits) against the ORACLE (oracle.c: generic
    decoder, pinned against the reference build in tests/test_oracle.py): small indexes built by the oracle's helpers, so the
    host boundary (the reference's bit strings) is part of the path"""
    from conftest import assert_parity
    from helpers import finish_ivfpq
    from oracle import binding as ob
    from knowhere_amd import GpuIndex
    r = np.random.default_rng(4000 + seed)
    metric = int(r.integers(0, 2))
    d, M = [(128, 32), (128, 16), (64, 8), (96, 12), (128, 64), (48, 4)][int(r.integers(0, 6))]
    nbits = int(r.integers(1, 9))
    nb, nlist = int(r.choice([2000, 6000])), int(r.choice([4, 12, 30]))
    xb = gen_data(nb, d, seed, -4.0, 4.0)
    ix = finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=nlist, M=M, nbits=nbits, seed=seed))
    g = GpuIndex.from_data(ix, device=0)
    for case in range(3):
        nq = int(r.choice([1, 33, 200]))
        k = int(r.choice([1, 10, 100, 300]))
        nprobe = int(min(nlist, r.choice([1, 3, 30])))
        xq = gen_data(nq, d, 90 + case, -4.0, 4.0)
        frac = float(r.choice([0.0, 0.4, 0.97]))
        bs = np.packbits(r.random(nb) < frac, bitorder="little") if frac > 0 else None
        Do, Io = port.search(ix, xq, k, nprobe, bs, nb if bs is not None else 0)
        D, I = g.search(xq, k, nprobe, bs, nb if bs is not None else 0)
        assert_parity(Do, Io, D, I, metric, f"seed={seed} d={d} m={M} nbits={nbits} nb={nb} nlist={nlist} nq={nq} k={k} "
                                            f"nprobe={nprobe} filter={frac}")
    g.close()
