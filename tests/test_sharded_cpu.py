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
    assert_parity(D0, I0, D.numpy(), I.numpy(), ob.L2, f"rank {rank}: sharded == monolithic", licensed_ties=True)
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
    assert_parity(D20, I20, D2.numpy(), I2.numpy(), ob.L2, f"rank {rank}: query-sharded coarse + list-sharded scan", licensed_ties=True)
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
    assert_parity(Dc0, Ic0, Dcu.numpy(), Icu.numpy(), ob.L2, "first stage union == monolithic", licensed_ties=True)
    rows = sharded.ids_to_rows(Icu, torch.from_numpy(own_ids))
    Dr, Rr = port.refine(ob.L2, own_rows, xq2, rows.numpy(), k)
    Ir = sharded.rows_to_ids(torch.from_numpy(Rr), torch.from_numpy(own_ids))
    D3, I3 = comm.allgather_merge(ob.L2, torch.from_numpy(Dr), Ir)
    D30, I30 = port.refine(ob.L2, xb, xq2, Icu.numpy(), k)
    assert_parity(D30, I30, D3.numpy(), I3.numpy(), ob.L2, f"rank {rank}: owner-side refine == monolithic refine", licensed_ties=True)
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
