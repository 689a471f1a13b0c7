"""GPU tests (-m gpu): which of several candidates TIED at the k-th distance is returned.

The reference admits first-come (HeapResultHandler::add_result, thirdparty/faiss/faiss/impl/ResultHandler.h:258-279) and
evicts by id (heap_replace_top / cmp2, utils/Heap.h:113-151) -- tests/test_tie_rule.py states the rule and its closed
form; the library applies it (knhip_api.hip::search_batch_ties, refine.hip).  Here: bases made of a few distinct rows
repeated many times, so that nearly every query has more candidates at its k-th distance than places, for every index
kind, both metrics, with and without a bitset, through the host and the device boundary and with a given coarse
assignment -- ids must EQUAL the oracle's (assert_parity without its licence), `tie_queries` must show that the rule
ran, and KNHIP_TIES=canonical must bring the old (licensed) canonical answer back without the read-back."""
import numpy as np
import pytest

from conftest import assert_parity
from helpers import finish_ivfpq
from oracle import binding as ob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch


def _dup_data(nb, d, nproto, seed, scale=7.0):
    """nproto distinct rows (small integers: sums are exact in fp32), each repeated ~nb / nproto times in random order"""
    rng = np.random.default_rng(seed)
    proto = (rng.integers(-3, 4, (nproto, d)) * scale).astype(np.float32)
    xb = proto[rng.integers(0, nproto, nb)]
    xq = (proto[rng.integers(0, nproto, 96)] + rng.integers(0, 2, (96, d)).astype(np.float32)).astype(np.float32)
    return np.ascontiguousarray(xb), np.ascontiguousarray(xq)


def _gpu(ix):
    from knowhere_amd import GpuIndex
    return GpuIndex.from_data(ix, device=0)


KINDS = [(ob.FLAT, "flat", {}), (ob.IVF_FLAT, "ivfflat", dict(nlist=24)), (ob.IVF_SQ8, "ivfsq8", dict(nlist=24)),
         (ob.IVF_PQ, "ivfpq8", dict(nlist=24, M=8)), (ob.IVF_PQ, "ivfpq32", dict(nlist=24, M=32))]


@pytest.mark.parametrize("kind,name,kw", KINDS, ids=[k[1] for k in KINDS])
@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_ties_at_the_kth_boundary_follow_the_reference(torch_cuda, port, kind, name, kw, metric):
    torch = torch_cuda
    d = 128 if name == "ivfpq32" else 32
    xb, xq = _dup_data(6000, d, 40, 17 + kind)
    ix = ob.make_index(port, kind, metric, xb, **kw)
    if kind == ob.IVF_PQ:
        finish_ivfpq(port, ix)
    g = _gpu(ix)
    g.profile_enable(True)
    g.profile_reset()
    nprobe = 9
    bs = np.packbits(np.random.default_rng(5).random(len(xb)) < 0.3, bitorder="little")
    for k in (1, 7, 40):
        for bitset, nbits in ((None, 0), (bs, len(xb))):
            Do, Io = port.search(ix, xq, k, nprobe, bitset, nbits)
            D, I = g.search(xq, k, nprobe, bitset, nbits)
            assert_parity(Do, Io, D, I, metric, f"{name} metric={metric} k={k} bitset={bitset is not None} (host boundary)")
    k = 7
    Do, Io = port.search(ix, xq, k, nprobe)
    Dt, It = g.search_device(torch.from_numpy(xq).cuda(), k, nprobe)
    torch.cuda.synchronize()
    assert_parity(Do, Io, Dt.cpu().numpy(), It.cpu().numpy(), metric, f"{name} metric={metric} (device boundary)")
    if kind != ob.FLAT:
        qt = torch.from_numpy(xq).cuda()
        cd, keys = g.coarse_search_device(qt, nprobe)
        Dp, Ip = g.search_preassigned_device(qt, k, keys, cd)
        torch.cuda.synchronize()
        assert_parity(Do, Io, Dp.cpu().numpy(), Ip.cpu().numpy(), metric, f"{name} metric={metric} (given assignment)")
    p = g.profile_get()
    assert p["tie_queries"] > 0, "the data did not produce a single ambiguous boundary: the test tests nothing"
    g.close()


@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_canonical_mode_is_the_old_licensed_answer(torch_cuda, port, monkeypatch, metric):
    xb, xq = _dup_data(6000, 32, 40, 3)
    ix = ob.make_index(port, ob.IVF_FLAT, metric, xb, nlist=24)
    g = _gpu(ix)
    g.profile_enable(True)
    g.profile_reset()
    Do, Io = port.search(ix, xq, 7, 9)
    monkeypatch.setenv("KNHIP_TIES", "canonical")
    D, I = g.search(xq, 7, 9)
    monkeypatch.delenv("KNHIP_TIES")
    assert g.profile_get()["tie_queries"] == 0
    assert_parity(Do, Io, D, I, metric, "canonical ties", licensed_ties=True)
    if metric == ob.IP:  # (L2: canonical = ascending ids = storage order: the first arrivals ARE the canonical ties here)
        assert (I != Io).any(), "ambiguous boundaries: the canonical answer differs from the reference's somewhere"
    # the canonical answer is the (distance, id) order: L2 smallest ids first, IP largest
    for q in range(len(xq)):
        tied = D[q] == D[q, -1]
        ids = I[q][tied]
        assert (np.diff(ids) > 0).all() if metric == ob.L2 else (np.diff(ids) < 0).all()
    g.close()


@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_refine_ties_follow_reorder_2_heaps(torch_cuda, port, metric):
    """IndexRefine::search pushes the re-scored candidates through a heap IN CANDIDATE ORDER (reorder_2_heaps): duplicates of
    one raw row tie exactly after the re-rank, and which of them are returned depends on where they stood in the first
    stage's result"""
    from knowhere_amd import index as kidx
    torch = torch_cuda
    xb, xq = _dup_data(6000, 32, 40, 11)
    ix = ob.make_index(port, ob.IVF_SQ8, metric, xb, nlist=24)
    g = _gpu(ix)
    kbase, k, nprobe = 60, 6, 9
    Db, Ib = port.search(ix, xq, kbase, nprobe)
    Dr, Ir = port.refine(metric, xb, xq, Ib, k)
    base_t, qt = torch.from_numpy(xb).cuda(), torch.from_numpy(xq).cuda()
    _, It = g.search_device(qt, kbase, nprobe)
    Dg, Ig = kidx.refine_device(metric, base_t, qt, It, k)
    torch.cuda.synchronize()
    assert np.array_equal(It.cpu().numpy(), Ib), "first stage (with its own boundary ties) == oracle"
    assert_parity(Dr, Ir, Dg.cpu().numpy(), Ig.cpu().numpy(), metric, "refine over duplicates")
    # host boundary: knhip_search_refine == the same two stages
    raw = kidx.GpuIndex(kidx.BRUTE_FORCE, metric, xb.shape[1], device=0)
    raw.add_vectors(xb)
    Dh, Ih = g.search_refine(raw, xq, k, kbase, nprobe)
    assert_parity(Dr, Ir, Dh, Ih, metric, "knhip_search_refine over duplicates")
    raw.close()
    g.close()
