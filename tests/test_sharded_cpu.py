"""CPU tests of the multi-GPU plumbing (world_size 2, gloo): list partition, all-gather + merge of
per-shard partial top-k == monolithic search (reference property: tests/ut/test_bruteforce.cc:128-181).
The per-shard searches are played by the oracle here (no GPU); what is under test is the
partitioning, the collective and the product's host merge (knhip_merge_topk_host)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, assert_parity, gen_data


def _worker(rank, world, port_no, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port_no)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from knowhere_amd import sharded
    from oracle import binding as ob
    port = ob.Port()
    xb, xq = gen_data(6000, 32, 42), gen_data(40, 32, 44)
    ix = ob.make_index(port, ob.IVF_PQ, ob.L2, xb, nlist=24, M=8)
    sizes = np.array([len(i) for i in ix.list_ids])
    masks = sharded.partition_lists(sizes, world)
    # every list owned exactly once, loads balanced
    assert (np.sum(masks, axis=0) == 1).all()
    loads = [int(sizes[m].sum()) for m in masks]
    assert max(loads) - min(loads) <= sizes.max()
    sub = ob.IndexData(ix.kind, ix.metric, ix.d, ix.nlist, ix.M, ix.nbits)
    sub.centroids, sub.pq_centroids = ix.centroids, ix.pq_centroids
    sub.precomputed_table, sub.use_precomputed_table = ix.precomputed_table, 1
    sub.list_codes = [c if masks[rank][l] else c[:0] for l, c in enumerate(ix.list_codes)]
    sub.list_ids = [i if masks[rank][l] else i[:0] for l, i in enumerate(ix.list_ids)]
    Dl, Il = port.search(sub, xq, 10, 8)  # this rank's partial top-k (its own lists only)
    comm = sharded.Comm()
    D, I = comm.allgather_merge(ob.L2, torch.from_numpy(Dl), torch.from_numpy(Il))
    D0, I0 = port.search(ix, xq, 10, 8)
    assert_parity(D0, I0, D.numpy(), I.numpy(), ob.L2, f"rank {rank}: sharded == monolithic")
    # the coarse quantizer sharded by queries + search_preassigned over the owned lists (bench.py's N > 1 step):
    # 41 queries over 2 ranks exercises the padded last slice
    xq2 = gen_data(41, 32, 45)

    def coarse(lo, hi):
        cd, ck = port.coarse_search(ix, xq2[lo:hi], 8)
        return torch.from_numpy(cd), torch.from_numpy(ck)

    keys, cdis = sharded.sharded_coarse(comm, coarse, 41, 8)
    cd0, ck0 = port.coarse_search(ix, xq2, 8)
    assert np.array_equal(keys.numpy(), ck0) and np.array_equal(cdis.numpy().view(np.uint32), cd0.view(np.uint32))
    Dl2, Il2 = port.ivf_search_preassigned(sub, xq2, 10, keys.numpy(), cdis.numpy())
    D2, I2 = comm.allgather_merge(ob.L2, torch.from_numpy(Dl2), torch.from_numpy(Il2))
    D20, I20 = port.search(ix, xq2, 10, 8)
    assert_parity(D20, I20, D2.numpy(), I2.numpy(), ob.L2, f"rank {rank}: query-sharded coarse + list-sharded scan")
    t = comm.max_float(float(rank))
    assert t == world - 1
    b = torch.full((3,), float(rank))
    comm.broadcast(b, 0)
    assert (b == 0).all()
    # refine ownership (bench.py's N > 1 step): the global top-kbase PQ candidates (all-gather + merge) are re-ranked
    # by the rank that holds their raw vectors (kept only for owned lists); a second packed all-gather + merge ==
    # the monolithic IndexRefine result, bit for bit
    kbase, k = 40, 10
    own_ids = np.sort(np.concatenate([i for l, i in enumerate(ix.list_ids) if masks[rank][l]] or
                                     [np.empty(0, np.int64)]))
    own_rows = xb[own_ids]
    Dc, Ic = port.ivf_search_preassigned(sub, xq2, kbase, keys.numpy(), cdis.numpy())
    Dcu, Icu = comm.allgather_merge(ob.L2, torch.from_numpy(Dc), torch.from_numpy(Ic))
    Dc0, Ic0 = port.search(ix, xq2, kbase, 8)
    assert_parity(Dc0, Ic0, Dcu.numpy(), Icu.numpy(), ob.L2, "first stage union == monolithic")
    rows = sharded.ids_to_rows(Icu, torch.from_numpy(own_ids))
    Dr, Rr = port.refine(ob.L2, own_rows, xq2, rows.numpy(), k)
    Ir = sharded.rows_to_ids(torch.from_numpy(Rr), torch.from_numpy(own_ids))
    D3, I3 = comm.allgather_merge(ob.L2, torch.from_numpy(Dr), Ir)
    D30, I30 = port.refine(ob.L2, xb, xq2, Icu.numpy(), k)
    assert_parity(D30, I30, D3.numpy(), I3.numpy(), ob.L2, f"rank {rank}: owner-side refine == monolithic refine")
    # ids this rank does not hold -> -2 (skipped by the re-rank), -1 still ends a row
    m = sharded.ids_to_rows(torch.tensor([[1, 2, 3, -1]]), torch.tensor([1, 3, 7]))
    assert m.tolist() == [[0, -2, 1, -1]]
    assert sharded.rows_to_ids(m, torch.tensor([1, 3, 7])).tolist() == [[1, -2, 3, -1]]
    # the packed buffer round-trips (float32, int64) exactly
    Dx, Ix = comm.unpack(comm.pack(torch.from_numpy(Dl), torch.from_numpy(Il)))
    assert torch.equal(Dx.view(torch.int32), torch.from_numpy(Dl).view(torch.int32)) and torch.equal(Ix, torch.from_numpy(Il))
    s = comm.allreduce_sum(torch.tensor([rank + 1]))
    assert int(s.item()) == world * (world + 1) // 2
    dist.barrier()
    dist.destroy_process_group()
    ret[rank] = 1


def _worker_ties(rank, world, port_no, ret):
    """The tie rule ACROSS shards (sharded.search_sharded / refine_sharded, host forms of flag / resolve / select): integer
    coordinates make every distance an exact integer, so ties at the k-th distance between rows of different lists -- of
    different ranks -- are everywhere.  Every rank plays its shard in numpy (exact on integers): canonical partials, arrivals
    in scan order with their (probe rank, position) keys; the protocol, the collectives and the product's host rule are what
    is under test.  Bar: the reference's answer (oracle heap over the whole index), NO licence."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port_no)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from knowhere_amd import sharded
    from oracle import binding as ob
    port = ob.Port()
    rng = np.random.default_rng(42)
    nb, d, nlist, nq, nprobe = 5000, 12, 20, 48, 7
    xb = rng.integers(0, 4, (nb, d)).astype(np.float32)
    xq = np.random.default_rng(44).integers(0, 4, (nq, d)).astype(np.float32)
    comm = sharded.Comm()
    fmax = np.float32(np.finfo(np.float32).max)
    for metric in (ob.L2, ob.IP):
        l2 = metric == ob.L2
        ix = ob.make_index(port, ob.IVF_FLAT, metric, xb, nlist=nlist)
        sizes = np.array([len(i) for i in ix.list_ids])
        mask = sharded.partition_lists(sizes, world)[rank]
        cd0, ck0 = port.coarse_search(ix, xq, nprobe)
        rows = [np.ascontiguousarray(c).view(np.float32).reshape(-1, d) for c in ix.list_codes]
        bs = np.packbits(np.random.default_rng(3).random(nb) < 0.3, bitorder="little")

        def arrivals_of(q, bitset):
            """this shard's candidates of query q in scan order: (dist, id, key)"""
            out = []
            for r_, l in enumerate(ck0[q]):
                if l < 0 or not mask[l]:
                    continue
                x = rows[l]
                dis = ((xq[q] - x) ** 2).sum(1, dtype=np.float32) if l2 else (xq[q] * x).sum(1, dtype=np.float32)
                for pos, (dd, i) in enumerate(zip(dis, ix.list_ids[l])):
                    if bitset is not None and (bitset[i >> 3] >> (i & 7)) & 1:
                        continue
                    out.append((np.float32(dd), int(i), (r_ << 40) | pos))
            return out

        for k in (1, 5, 16, 40):
            for bitset in (None, bs):
                cands = [arrivals_of(q, bitset) for q in range(nq)]

                def partial_fn(kk):
                    D = np.full((nq, kk), fmax if l2 else -fmax, np.float32)
                    I = np.full((nq, kk), -1, np.int64)
                    for q in range(nq):
                        c = sorted(cands[q], key=(lambda t: (t[0], t[1])) if l2 else (lambda t: (-t[0], -t[1])))[:kk]
                        for j, t in enumerate(c):
                            D[q, j], I[q, j] = t[0], t[1]
                    return torch.from_numpy(D), torch.from_numpy(I)

                def arrivals_fn(flagged, can_d):
                    fl = flagged.numpy()
                    ad = np.zeros((len(fl), k), np.float32)
                    ai = np.full((len(fl), k), -1, np.int64)
                    ak = np.zeros((len(fl), k), np.int64)
                    an = np.zeros((len(fl),), np.int64)
                    for f, q in enumerate(fl):
                        v = can_d[q, k - 1].item()
                        arr = [t for t in cands[q] if (t[0] <= v if l2 else t[0] >= v)]
                        an[f] = len(arr)
                        for j, t in enumerate(arr[:k]):
                            ad[f, j], ai[f, j], ak[f, j] = t
                    return tuple(torch.from_numpy(a) for a in (ad, ai, ak, an))

                D, I = sharded.search_sharded(comm, metric, k, partial_fn, arrivals_fn)
                D0, I0 = port.search(ix, xq, k, nprobe, bitset, nb if bitset is not None else 0)
                assert_parity(D0, I0, D.numpy(), I.numpy(), metric,
                              f"rank {rank} world {world}: sharded ties k={k} metric={metric} bitset={bitset is not None}")
                if rank == 0 and k == 5 and bitset is None:
                    # (the fixture really ties: the canonical merge of the partials would not be the reference's answer)
                    Dc, Ic = partial_fn(k)
                    ret[f"tied{metric}"] = int(((D0[:, -1:] == D0).sum(1) > 1).sum())
        # ---- refine over shards: distances where the rows are, one all-gather, ONE selection (host form)
        k, kb = 4, 30
        Dc0, Ic0 = port.search(ix, xq, kb, nprobe)  # the merged first stage (same on every rank)
        lo, hi = nb * rank // world, nb * (rank + 1) // world  # raw rows cut by id range, not along the lists
        dist_mine = np.full((nq, kb), -1, np.int32).view(np.float32).copy()
        for q in range(nq):
            for c in range(kb):
                i = Ic0[q, c]
                if lo <= i < hi:
                    dist_mine[q, c] = ((xq[q] - xb[i]) ** 2).sum(dtype=np.float32) if l2 else (xq[q] * xb[i]).sum(dtype=np.float32)
        D3, I3 = sharded.refine_sharded(comm, metric, k, torch.from_numpy(Ic0), lambda: torch.from_numpy(dist_mine))
        D30, I30 = port.refine(metric, xb, xq, Ic0, k)
        assert_parity(D30, I30, D3.numpy(), I3.numpy(), metric, f"rank {rank}: sharded refine ties metric={metric}")
        if rank == 0:
            ret[f"rtied{metric}"] = int(((D30[:, -1:] == D30).sum(1) > 1).sum())
    dist.barrier()
    dist.destroy_process_group()
    ret[rank] = 1


@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_ties_follow_the_reference_over_gloo(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    port_no = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_worker_ties, args=(world, port_no, ret), nprocs=world, join=True)
    assert all(ret.get(r) == 1 for r in range(world))
    assert ret["tied0"] > 0 and ret["tied1"] > 0 and ret["rtied0"] > 0 and ret["rtied1"] > 0, dict(ret)


@pytest.mark.parametrize("world", [2, 4])
def test_list_sharding_allgather_merge(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    port_no = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port_no, ret), nprocs=world, join=True)
    assert len(ret) == world


def test_partition_is_deterministic_and_balanced():
    from knowhere_amd import sharded
    r = np.random.default_rng(0)
    sizes = r.gamma(2.0, 3000, 16384).astype(np.int64)
    for world in (2, 4, 8):
        m1 = sharded.partition_lists(sizes, world)
        m2 = sharded.partition_lists(sizes.copy(), world)
        assert all((a == b).all() for a, b in zip(m1, m2))
        loads = np.array([sizes[m].sum() for m in m1], np.float64)
        assert loads.max() / loads.mean() < 1.001
