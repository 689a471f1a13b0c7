"""(e) multi-GPU, C++ host (-m gpu): knowhere_amd/host/shard_group.cc behind include/knhip_shards.h -- one worker thread per
rank, every rank scans the lists it owns, ONE all-gather of the packed (nq, k) partials, knhip_merge_topk_device.
On a single-GPU box the protocol runs with several ranks on device 0 through the STAGED transport (device copies instead of
ncclAllGather, everything else identical); the RCCL transport itself is exercised at world 1 (ncclCommInitAll +
ncclAllGather on hardware) and, when the box has more GPUs, at world = all of them.  Bar: bit-identical to the
single-GPU search of the whole index."""
import copy
import ctypes as C
import os

import numpy as np
import pytest

from conftest import assert_parity, gen_data
from helpers import finish_ivfpq
from oracle import binding as ob

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "knowhere_amd", "libknhip_shards.so")


@pytest.fixture(scope="module")
def shards():
    assert os.path.exists(SO), "build with __graft_entry__.build()"
    import torch
    assert torch.cuda.is_available()  # (torch brings its own HIP runtime up before the shard library's RCCL is loaded)
    torch.zeros(1).cuda()
    from knowhere_amd import _lib
    _lib.load()  # libknhip.so first (the shard library links it)
    L = C.CDLL(SO)
    L.knhip_shard_group_last_error.restype = C.c_char_p
    return L


def _split(ix, world):
    """rank r's index: the lists dealt to it by the size-balanced deal, every other list empty"""
    from knowhere_amd.sharded import partition_lists
    masks = partition_lists(np.array([len(i) for i in ix.list_ids]), world)
    parts = []
    for r in range(world):
        p = copy.copy(ix)
        p.list_codes = [c if masks[r][l] else c[:0] for l, c in enumerate(ix.list_codes)]
        p.list_ids = [i if masks[r][l] else i[:0] for l, i in enumerate(ix.list_ids)]
        parts.append(p)
    return parts


def _group_search(L, gpus, devices, transport, xq, k, nprobe, bitset=None, nbits=0):
    W = len(gpus)
    g = C.c_void_p()
    dev = (C.c_int32 * W)(*devices)
    rc = L.knhip_shard_group_create(C.c_int32(W), dev, C.c_int32(transport), C.byref(g))
    assert rc == 0, L.knhip_shard_group_last_error().decode()
    try:
        for r, gi in enumerate(gpus):
            assert L.knhip_shard_group_set_index(g, C.c_int32(r), gi.h) == 0
        nq = xq.shape[0]
        I = np.empty((nq, k), np.int64)
        D = np.empty((nq, k), np.float32)
        ms = np.zeros((W, 4), np.float32)
        bs = None if bitset is None else np.ascontiguousarray(bitset, np.uint8)
        rc = L.knhip_shard_group_search(g, xq.ctypes.data_as(C.c_void_p), C.c_int64(nq), C.c_int32(k), C.c_int32(nprobe),
                                        None if bs is None else bs.ctypes.data_as(C.c_void_p), C.c_int64(nbits),
                                        I.ctypes.data_as(C.c_void_p), D.ctypes.data_as(C.c_void_p),
                                        ms.ctypes.data_as(C.c_void_p))
        assert rc == 0, L.knhip_shard_group_last_error().decode()
        return D, I, ms
    finally:
        L.knhip_shard_group_destroy(g)


@pytest.mark.parametrize("kind,metric", [(ob.IVF_PQ, ob.L2), (ob.IVF_FLAT, ob.IP), (ob.IVF_SQ8, ob.L2)],
                         ids=["ivfpq_l2", "ivfflat_ip", "ivfsq8_l2"])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_search_equals_the_single_index(shards, port, kind, metric, world):
    from knowhere_amd import GpuIndex
    nb, d, nlist, nq = 30000, 128, 40, 50
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    ix = finish_ivfpq(port, ob.make_index(port, kind, metric, xb, nlist=nlist, M=32))
    whole = GpuIndex.from_data(ix, device=0)
    parts = [GpuIndex.from_data(p, device=0) for p in _split(ix, world)]
    try:
        for k, nprobe in ((10, 8), (100, nlist), (1, 3)):
            Dw, Iw = whole.search(xq, k, nprobe)
            Do, Io = port.search(ix, xq, k, nprobe)
            D, I, ms = _group_search(shards, parts, [0] * world, 1, xq, k, nprobe)
            assert np.array_equal(I, Iw) and np.array_equal(D.view(np.uint32), Dw.view(np.uint32)), (kind, k, nprobe)
            assert_parity(Do, Io, D, I, metric, f"sharded world={world} kind={kind} k={k}")
            assert (ms[:, 3] > 0).all()
        bs = np.packbits(np.random.default_rng(1).random(nb) < 0.4, bitorder="little")
        Dw, Iw = whole.search(xq, 10, 8, bs, nb)
        D, I, _ = _group_search(shards, parts, [0] * world, 1, xq, 10, 8, bs, nb)
        assert np.array_equal(I, Iw) and np.array_equal(D.view(np.uint32), Dw.view(np.uint32))
    finally:
        whole.close()
        for p in parts:
            p.close()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_sharded_search_with_refine(shards, port, world, metric):
    """refine stage of the C++ host: k_base PQ candidates per rank -> all-gather + merge -> every rank re-ranks the
    candidates whose raw rows it holds (raw rows split by id range, NOT along the list split) -> all-gather + merge.
    Bit-identical to knhip_search_refine of the whole index on one GPU, and to the oracle's IndexRefine."""
    import torch
    from knowhere_amd import GpuIndex, index as kidx
    nb, d, nlist, nq, k, kb = 30000, 128, 40, 50, 10, 100
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    ix = finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=nlist, M=32))
    whole = GpuIndex.from_data(ix, device=0)
    parts = [GpuIndex.from_data(p, device=0) for p in _split(ix, world)]
    xb_t = torch.from_numpy(xb).cuda()
    cuts = [nb * r // world for r in range(world + 1)]
    L = shards
    g = C.c_void_p()
    dev = (C.c_int32 * world)(*([0] * world))
    assert L.knhip_shard_group_create(C.c_int32(world), dev, C.c_int32(1), C.byref(g)) == 0
    try:
        for r, gi in enumerate(parts):
            assert L.knhip_shard_group_set_index(g, C.c_int32(r), gi.h) == 0
            lo, hi = cuts[r], cuts[r + 1]
            assert L.knhip_shard_group_set_raw(g, C.c_int32(r), C.c_void_p(xb_t[lo:hi].data_ptr()), C.c_int64(hi - lo),
                                               C.c_int64(lo)) == 0
        I = np.empty((nq, k), np.int64)
        D = np.empty((nq, k), np.float32)
        ms = np.zeros((world, 7), np.float32)
        rc = L.knhip_shard_group_search_refine(g, xq.ctypes.data_as(C.c_void_p), C.c_int64(nq), C.c_int32(k), C.c_int32(kb),
                                               C.c_int32(8), None, C.c_int64(0), I.ctypes.data_as(C.c_void_p),
                                               D.ctypes.data_as(C.c_void_p), ms.ctypes.data_as(C.c_void_p))
        assert rc == 0, L.knhip_shard_group_last_error().decode()
        # one GPU: candidates of the whole index, exact re-rank against all raw rows
        xq_t = torch.from_numpy(xq).cuda()
        Dp, Ip = whole.search_device(xq_t, kb, 8)
        Dr, Ir = kidx.refine_device(metric, xb_t, xq_t, Ip, k)
        torch.cuda.synchronize()
        assert np.array_equal(I, Ir.cpu().numpy()) and np.array_equal(D.view(np.uint32), Dr.cpu().numpy().view(np.uint32))
        assert (ms[:, 6] > 0).all() and (ms[:, 3] > 0).all()
    finally:
        L.knhip_shard_group_destroy(g)
        whole.close()
        for p in parts:
            p.close()


def _int_data(n, d, seed, hi=4):
    """small integer coordinates: every distance is an exactly representable integer, so exact ties are everywhere -- also
    between rows of different lists, i.e. of different shards"""
    return np.random.default_rng(seed).integers(0, hi, (n, d)).astype(np.float32)


def _untie_coarse(port, ix):
    """integer centroids tie with each other at a query's nprobe-th place, and the coarse quantizer's own boundary is not
    part of the rule under test (include/knhip.h): tiny distinct offsets make every coarse distance unique.  The lists
    keep their contents -- an index is valid whatever its centroids are."""
    ix.centroids = (ix.centroids + np.random.default_rng(9).random(ix.centroids.shape).astype(np.float32) * 1e-3).astype(np.float32)
    ix.precomputed_table = None
    return finish_ivfpq(port, ix)


@pytest.mark.parametrize("kind,metric", [(ob.IVF_FLAT, ob.L2), (ob.IVF_FLAT, ob.IP), (ob.IVF_PQ, ob.L2), (ob.FLAT, ob.L2),
                                         (ob.FLAT, ob.IP)],
                         ids=["ivfflat_l2", "ivfflat_ip", "ivfpq_l2", "flat_l2", "flat_ip"])
@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_ties_follow_the_reference(shards, port, kind, metric, world):
    """candidates tied at the k-th distance ACROSS shards: the group applies the reference's first-come admission rule once,
    after the merge, over all shards' candidates (knhip_tie_flag / _arrivals / _resolve) -- the answer of the single index
    and of the reference, with no licence (the reference's own sharding contract is ids-equal,
    tests/ut/test_bruteforce.cc:128-181)"""
    from knowhere_amd import GpuIndex
    from knowhere_amd.index import BRUTE_FORCE, L2 as KL2, IP as KIP
    nb, d, nlist, nq = 12000, 16, 24, 60
    xb, xq = _int_data(nb, d, 42), _int_data(nq, d, 44)
    if kind == ob.IVF_PQ:
        # (PQ distances carry the list's centroid: ties come from equal codes in one list -- copies of a row -- while the
        # better candidates of the query sit on other shards)
        xb[6000:6300] = xb[:300]
        xb[9000:9300] = xb[:300]
        xq = np.ascontiguousarray(xb[:nq])
    if kind == ob.FLAT:
        ix = ob.IndexData(ob.FLAT, metric, d)
        ix.base = xb
        whole = GpuIndex.from_data(ix, device=0)
        parts = []
        for r in range(world):
            lo, hi = nb * r // world, nb * (r + 1) // world
            g = GpuIndex(BRUTE_FORCE, KL2 if metric == ob.L2 else KIP, d, device=0)
            g.add_vectors(np.ascontiguousarray(xb[lo:hi]), id_offset=lo)
            parts.append(g)
        cases = ((10, 1), (37, 1), (1, 1), (99, 1))
    else:
        ix = _untie_coarse(port, ob.make_index(port, kind, metric, xb, nlist=nlist, M=4))
        whole = GpuIndex.from_data(ix, device=0)
        parts = [GpuIndex.from_data(p, device=0) for p in _split(ix, world)]
        cases = ((10, 8), (100, nlist), (1, 3), (64, 5), (2, 8))
    try:
        ntie = 0
        for k, nprobe in cases:
            Dw, Iw = whole.search(xq, k, nprobe)
            Do, Io = port.search(ix, xq, k, nprobe)
            D, I, _ = _group_search(shards, parts, [0] * world, 1, xq, k, nprobe)
            assert_parity(Do, Io, D, I, metric, f"sharded ties world={world} kind={kind} k={k}")
            assert np.array_equal(I, Iw) and np.array_equal(D.view(np.uint32), Dw.view(np.uint32)), (kind, k, nprobe)
            # (how many rows really sit on a tie: the canonical choice -- smallest ids -- would differ on some of them)
            ntie += int((Do[:, -1:] == Do).sum(axis=1).max() > 1)
        assert ntie > 0, "the fixture produced no tie at any k-th boundary"
        bs = np.packbits(np.random.default_rng(1).random(nb) < 0.4, bitorder="little")
        Do, Io = port.search(ix, xq, 10, cases[0][1], bs, nb)
        D, I, _ = _group_search(shards, parts, [0] * world, 1, xq, 10, cases[0][1], bs, nb)
        assert_parity(Do, Io, D, I, metric, f"sharded ties + bitset world={world} kind={kind}")
    finally:
        whole.close()
        for p in parts:
            p.close()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("metric", [ob.L2, ob.IP], ids=["l2", "ip"])
def test_sharded_refine_ties_follow_the_reference(shards, port, world, metric):
    """refine over shards with exact ties among the re-scored candidates (integer rows): IndexRefine pushes them through its
    heap in CANDIDATE order whoever holds their rows; the group exchanges the distances and runs ONE selection --
    bit-identical to knhip_search_refine on one GPU and to the oracle's IndexRefine"""
    import torch
    from knowhere_amd import GpuIndex
    nb, d, nlist, nq, k, kb = 12000, 16, 24, 60, 5, 60
    xb, xq = _int_data(nb, d, 42), _int_data(nq, d, 44)
    ix = _untie_coarse(port, ob.make_index(port, ob.IVF_PQ, metric, xb, nlist=nlist, M=4))
    whole = GpuIndex.from_data(ix, device=0)
    parts = [GpuIndex.from_data(p, device=0) for p in _split(ix, world)]
    xb_t = torch.from_numpy(xb).cuda()
    cuts = [nb * r // world for r in range(world + 1)]
    L = shards
    g = C.c_void_p()
    dev = (C.c_int32 * world)(*([0] * world))
    assert L.knhip_shard_group_create(C.c_int32(world), dev, C.c_int32(1), C.byref(g)) == 0
    try:
        for r, gi in enumerate(parts):
            assert L.knhip_shard_group_set_index(g, C.c_int32(r), gi.h) == 0
            lo, hi = cuts[r], cuts[r + 1]
            assert L.knhip_shard_group_set_raw(g, C.c_int32(r), C.c_void_p(xb_t[lo:hi].data_ptr()), C.c_int64(hi - lo),
                                               C.c_int64(lo)) == 0
        I = np.empty((nq, k), np.int64)
        D = np.empty((nq, k), np.float32)
        rc = L.knhip_shard_group_search_refine(g, xq.ctypes.data_as(C.c_void_p), C.c_int64(nq), C.c_int32(k), C.c_int32(kb),
                                               C.c_int32(8), None, C.c_int64(0), I.ctypes.data_as(C.c_void_p),
                                               D.ctypes.data_as(C.c_void_p), None)
        assert rc == 0, L.knhip_shard_group_last_error().decode()
        from knowhere_amd.index import BRUTE_FORCE
        raw = GpuIndex(BRUTE_FORCE, metric, d, device=0)  # the store of the raw rows knhip_search_refine reads
        raw.add_vectors(xb)
        Dg, Ig = whole.search_refine(raw, xq, k, kb, 8)
        raw.close()
        assert np.array_equal(I, Ig) and np.array_equal(D.view(np.uint32), Dg.view(np.uint32))
        Dc, Ic = port.search(ix, xq, kb, 8)
        Dr, Ir = port.refine(metric, xb, xq, Ic, k)
        assert_parity(Dr, Ir, D, I, metric, f"sharded refine ties world={world}")
        assert int(((Dr[:, -1:] == Dr).sum(axis=1) > 1).sum()) > 0, "no tie at any k-th boundary of the refine stage"
    finally:
        L.knhip_shard_group_destroy(g)
        whole.close()
        for p in parts:
            p.close()


def test_rccl_transport_on_the_devices_present(shards, port):
    """ncclCommInitAll + ncclAllGather on hardware: world = every GPU of the box (1 on the test box: the collective is
    then a copy, but communicator creation, the rank-count check and the call path are the real ones)"""
    import torch
    from knowhere_amd import GpuIndex
    world = torch.cuda.device_count()
    nb, d, nlist, nq = 20000, 128, 32, 40
    xb, xq = gen_data(nb, d, 42), gen_data(nq, d, 44)
    ix = finish_ivfpq(port, ob.make_index(port, ob.IVF_PQ, ob.L2, xb, nlist=nlist, M=32))
    whole = GpuIndex.from_data(ix, device=0)
    parts = [GpuIndex.from_data(p, device=r) for r, p in enumerate(_split(ix, world))]
    try:
        Dw, Iw = whole.search(xq, 10, 8)
        D, I, ms = _group_search(shards, parts, list(range(world)), 0, xq, 10, 8)
        assert np.array_equal(I, Iw) and np.array_equal(D.view(np.uint32), Dw.view(np.uint32))
    finally:
        whole.close()
        for p in parts:
            p.close()


def test_rccl_transport_world2_on_two_devices(shards, port):
    """RCCL with more than one rank: runs whenever the box has two GPUs (skipped, loudly, on a one-GPU box -- no round's test
    box has had two so far, so the first multi-GPU box is where ncclAllGather at world 2 first executes).  The communicator
    must span exactly the two devices (a rank-count mismatch fails the group's creation: shard_group.cc checks
    ncclCommCount); results -- ties at the k-th boundary and the refine stage included -- equal the single index's."""
    import torch
    from knowhere_amd import GpuIndex
    from knowhere_amd.index import BRUTE_FORCE
    if torch.cuda.device_count() < 2:
        pytest.skip(f"RCCL at world 2 needs two GPUs; this box has {torch.cuda.device_count()}")
    world, nb, d, nlist, nq, k, kb = 2, 12000, 16, 24, 60, 5, 60
    xb, xq = _int_data(nb, d, 42), _int_data(nq, d, 44)
    for kind, metric in ((ob.IVF_FLAT, ob.L2), (ob.IVF_PQ, ob.IP)):
        ix = _untie_coarse(port, ob.make_index(port, kind, metric, xb, nlist=nlist, M=4))
        parts = [GpuIndex.from_data(p, device=r) for r, p in enumerate(_split(ix, world))]
        try:
            for kk_, nprobe in ((10, 8), (1, 3), (64, 5)):
                Do, Io = port.search(ix, xq, kk_, nprobe)
                D, I, _ = _group_search(shards, parts, [0, 1], 0, xq, kk_, nprobe)
                assert_parity(Do, Io, D, I, metric, f"RCCL world 2 kind={kind} k={kk_}")
            if kind == ob.IVF_PQ:
                g = C.c_void_p()
                dev = (C.c_int32 * world)(0, 1)
                assert shards.knhip_shard_group_create(C.c_int32(world), dev, C.c_int32(0), C.byref(g)) == 0, \
                    shards.knhip_shard_group_last_error().decode()
                assert shards.knhip_shard_group_size(g) == world
                raws = []
                try:
                    for r, gi in enumerate(parts):
                        assert shards.knhip_shard_group_set_index(g, C.c_int32(r), gi.h) == 0
                        lo, hi = nb * r // world, nb * (r + 1) // world
                        t = torch.from_numpy(xb[lo:hi]).to(f"cuda:{r}")
                        raws.append(t)
                        assert shards.knhip_shard_group_set_raw(g, C.c_int32(r), C.c_void_p(t.data_ptr()), C.c_int64(hi - lo),
                                                                C.c_int64(lo)) == 0
                    I = np.empty((nq, k), np.int64)
                    D = np.empty((nq, k), np.float32)
                    rc = shards.knhip_shard_group_search_refine(g, xq.ctypes.data_as(C.c_void_p), C.c_int64(nq), C.c_int32(k),
                                                                C.c_int32(kb), C.c_int32(8), None, C.c_int64(0),
                                                                I.ctypes.data_as(C.c_void_p), D.ctypes.data_as(C.c_void_p), None)
                    assert rc == 0, shards.knhip_shard_group_last_error().decode()
                    Dc, Ic = port.search(ix, xq, kb, 8)
                    Dr, Ir = port.refine(metric, xb, xq, Ic, k)
                    assert_parity(Dr, Ir, D, I, metric, "RCCL world 2 refine")
                finally:
                    shards.knhip_shard_group_destroy(g)
        finally:
            for p in parts:
                p.close()


def test_rccl_transport_refuses_one_device_twice(shards):
    g = C.c_void_p()
    dev = (C.c_int32 * 2)(0, 0)
    assert shards.knhip_shard_group_create(C.c_int32(2), dev, C.c_int32(0), C.byref(g)) != 0
    assert b"distinct devices" in shards.knhip_shard_group_last_error()


@pytest.mark.parametrize("seed", range(int(os.environ.get("KNHIP_FUZZ_SEEDS", "8"))))
def test_sharded_search_equals_the_single_index_on_random_shapes(shards, seed):
    """random kind / metric / world 2 .. 4 / shape / k / nprobe / filter: the list-sharded group (staged transport, every rank on
    device 0) returns the single index's ids and distance bits -- boundary ties included (duplicated rows spread over the
    shards).  Device-built indexes; the single index is the bar (pinned against the oracle elsewhere)."""
    import types
    from knowhere_amd import GpuIndex
    r = np.random.default_rng(5000 + seed)
    kind = int(r.choice([1, 2, 3]))
    metric = int(r.integers(0, 2))
    world = int(r.integers(2, 5))
    d = 128 if kind == 2 else int(r.choice([32, 96, 128]))
    nb, nlist = int(r.choice([4000, 40000])), int(r.choice([8, 32, 100]))
    xb = gen_data(nb, d, seed, -3.0, 3.0)
    xb[300:340] = xb[5]
    b = GpuIndex(kind, metric, d, nlist, 32 if kind == 2 else 0, 8, device=0)
    b.train(xb)
    b.add(xb)
    sizes, codes, ids = b.get_lists()
    ix = types.SimpleNamespace(kind=kind, metric=metric, d=d, nlist=nlist, M=32 if kind == 2 else 0, nbits=8,
                               centroids=b.get_coarse(), pq_centroids=b.get_pq() if kind == 2 else None,
                               sq_trained=b.get_sq() if kind == 3 else None, list_codes=[], list_ids=[])
    pos = 0
    for l in range(nlist):
        n = int(sizes[l])
        ix.list_codes.append(codes[pos:pos + n])
        ix.list_ids.append(ids[pos:pos + n])
        pos += n
    b.close()
    whole = GpuIndex.from_data(ix, device=0)
    parts = [GpuIndex.from_data(p, device=0) for p in _split(ix, world)]
    try:
        for case in range(4):
            nq = int(r.choice([1, 17, 300]))
            k = int(r.choice([1, 10, 100, 400]))
            nprobe = int(min(nlist, r.choice([1, 4, 100])))
            xq = np.concatenate([xb[5:6] + 0.001, gen_data(nq, d, 70 + case, -3.0, 3.0)])[:nq].astype(np.float32)
            frac = float(r.choice([0.0, 0.5, 0.98]))
            bs = np.packbits(r.random(nb) < frac, bitorder="little") if frac > 0 else None
            Dw, Iw = whole.search(xq, k, nprobe, bs, nb if bs is not None else 0)
            D, I, _ = _group_search(shards, parts, [0] * world, 1, xq, k, nprobe, bs, nb if bs is not None else 0)
            what = f"seed={seed} kind={kind} metric={metric} world={world} d={d} nb={nb} nlist={nlist} nq={nq} k={k} nprobe={nprobe} filter={frac}"
            assert np.array_equal(I, Iw), what + f": {int((I != Iw).any(1).sum())} queries differ in ids"
            assert np.array_equal(D.view(np.uint32), Dw.view(np.uint32)), what + ": distance bits"
    finally:
        whole.close()
        for p in parts:
            p.close()
