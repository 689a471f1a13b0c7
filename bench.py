#!/usr/bin/env python
"""bench.py -- contract benchmark.

Metric (BASELINE.json): QPS at recall@10 >= 0.95, IVF-PQ m=32 x 8 bit, 100M x d=128 fp32,
nlist=16384, nprobe=128, batch = 10k queries, k=10, on 1/2/4/8 MI355X.

One "step" = one Search() of the whole 10k-query batch: coarse quantizer -> PQ query tables ->
per-list ADC scan -> per-query merge of the top-`refine_k` PQ candidates -> exact fp32 re-rank
(Knowhere's `refine`, IndexRefine) -> top-10.  Queries, index and raw vectors are resident in HBM
when the timed region starts; nothing is cached between steps (every step recomputes everything).

N > 1 (launched by torch.distributed.run, one rank per GPU): STRONG scaling on the same 100M
index.  Inverted lists are partitioned across ranks (size-balanced), the coarse quantizer and PQ
codebooks are replicated, every rank sees the full query batch and scans only the probes it owns;
the per-rank partial top-`refine_k` are exchanged with ONE all-gather (RCCL over xGMI) and merged
on device; each rank re-ranks the candidates whose raw vectors it owns and a second small
all-gather + merge yields the final top-10 (bit-identical to the single-GPU result).

Extra JSON objects: "roofline" (dominant kernel = the ADC scan, algorithmic bytes of SURVEY.md 8d
/ HIP-event time measured here) and "cpu_baseline" (the reference's own FAISS, or the oracle port,
timed on this box's host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from knowhere_amd import build as kb  # noqa: E402
from knowhere_amd import index as kidx  # noqa: E402
from knowhere_amd import sharded  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nb", type=int, default=100_000_000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--nlist", type=int, default=16384)
    ap.add_argument("--nprobe", type=int, default=128)
    ap.add_argument("--nq", type=int, default=10000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--m", type=int, default=32)
    ap.add_argument("--refine-k", type=int, default=100, help="PQ candidates re-ranked per query (0 = no refine)")
    ap.add_argument("--data", default="mixture", choices=["mixture", "uniform"])
    ap.add_argument("--sigma", type=float, default=0.35)
    ap.add_argument("--ncenter", type=int, default=0, help="mixture components (0 = nb/160 rounded to a power of two)")
    ap.add_argument("--latent", type=int, default=0, help="intrinsic dimension of a component (0 = isotropic)")
    ap.add_argument("--gt-queries", type=int, default=1000, help="queries used for the recall measurement")
    ap.add_argument("--cpu-queries", type=int, default=192, help="queries of the CPU baseline sample (0 = skip)")
    ap.add_argument("--backend", default=None, help="nccl (default for N>1) | gloo (single-GPU debugging)")
    ap.add_argument("--verbose", action="store_true")
    return ap.parse_args()


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    ndev = torch.cuda.device_count()
    assert ndev > 0, "bench.py needs a GPU"
    dev_id = local_rank % ndev
    torch.cuda.set_device(dev_id)
    dev = torch.device(f"cuda:{dev_id}")
    comm = None
    if world > 1:
        backend = a.backend or "nccl"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=dev if backend == "nccl" else None)
        comm = sharded.Comm(dev)

    # ---------------------------------------------------------------- build
    t_build = time.time()
    if a.ncenter <= 0:
        a.ncenter = 1 << max(4, int(round(np.log2(max(a.nb / 160.0, 16.0)))))
    spec = kb.DataSpec(a.nb, a.d, kind=a.data, seed=42, ncenter=a.ncenter, sigma=a.sigma, latent=a.latent)
    cen = cb = None
    if world > 1:
        # rank 0 trains; centroids and codebooks are broadcast so every shard quantises identically
        if rank == 0:
            tmp = kb.build_ivf(spec, kidx.IVF_PQ, kidx.L2, a.nlist, a.m, device=str(dev), train_only=True,
                               verbose=a.verbose)
            cen, cb = tmp.centroids, tmp.codebooks
        else:
            cen = torch.empty((a.nlist, a.d), device=dev)
            cb = torch.empty((a.m, 256, a.d // a.m), device=dev)
        cen = comm.broadcast(cen)
        cb = comm.broadcast(cb)
    built = kb.build_ivf(spec, kidx.IVF_PQ, kidx.L2, a.nlist, a.m, device=str(dev), centroids=cen, codebooks=cb,
                         verbose=a.verbose and rank == 0, keep_vectors=a.refine_k > 0)
    sizes = built.list_offsets[1:] - built.list_offsets[:-1]
    owned = sharded.partition_lists(sizes, world)[rank] if world > 1 else None
    g = built.to_gpu_index(device=dev_id, owned_lists=owned)
    vectors = getattr(built, "vectors", None)  # [nb, d] fp32, row = id
    own_row = None
    if world > 1 and vectors is not None:
        # a rank re-ranks only candidates whose raw vector it owns (owner = owner of the id's list)
        own_row = sharded.owned_id_mask(built, owned)
    xq = kb.queries(spec, a.nq, dev)
    torch.cuda.synchronize()
    build_s = time.time() - t_build
    log(rank, f"build {build_s:.1f}s {built.timings} index {g.device_bytes / 1e9:.2f} GB/rank, "
              f"precomputed_table={g.uses_precomputed_table}")

    kbase = a.refine_k if a.refine_k > 0 else a.k

    # N > 1: the coarse quantizer is sharded by QUERIES (each rank assigns nq / N of them, one all-gather of the
    # (nq, nprobe) assignment), the scan by LISTS (knhip_search_preassigned_device = IndexIVF::search_preassigned)
    def step():
        if world > 1:
            keys, cdis = sharded.sharded_coarse(comm, lambda lo, hi: g.coarse_search_device(xq[lo:hi], a.nprobe),
                                                a.nq, a.nprobe, device=dev)
            Dp, Ip = g.search_preassigned_device(xq, kbase, keys, cdis)
            Dp, Ip = comm.allgather_merge(kidx.L2, Dp, Ip)
        else:
            Dp, Ip = g.search_device(xq, kbase, a.nprobe)
        if a.refine_k > 0:
            cand = Ip if own_row is None else sharded.mask_unowned(Ip, own_row)
            D, I = kidx.refine_device(kidx.L2, vectors, xq, cand, a.k)
            if world > 1:
                D, I = comm.allgather_merge(kidx.L2, D, I)
            return D, I
        return Dp[:, :a.k].contiguous(), Ip[:, :a.k].contiguous()

    # ---------------------------------------------------------------- recall gate
    D, I = step()
    torch.cuda.synchronize()
    ngt = min(a.gt_queries, a.nq)
    _, gt = kb.ground_truth(spec, xq[:ngt], a.k, device=str(dev))
    hits = (I[:ngt].unsqueeze(2) == gt.unsqueeze(1)).any(dim=2).float().sum().item()
    rec = hits / (ngt * a.k)
    log(rank, f"recall@{a.k} = {rec:.4f} over {ngt} queries (refine_k={a.refine_k})")

    # ---------------------------------------------------------------- timed region
    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            comm.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    g.profile_enable(True)
    g.profile_reset()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(a.steps):
        step()
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = comm.max_float(dt)
    prof = g.profile_get()
    g.profile_enable(False)
    ms_per_step = dt / a.steps * 1e3
    qps = a.nq * a.steps / dt

    # dominant kernel: the ADC scan.  achieved = algorithmic bytes / mean launch time, both per launch
    # (the bulk launch over probes 1..nprobe-1; the rank-0 probes run in a separate dump + radix-select
    # phase whose time is reported as stage "scan_rank0" and whose bytes are excluded here)
    nlaunch = max(prof["launches"][kidx._lib.STAGE_SCAN], 1)
    scan_ms = prof["ms"][kidx._lib.STAGE_SCAN] / nlaunch
    scan_bytes = (prof["scan_bytes"] - prof["scan_bytes_rank0"]) / nlaunch
    achieved = scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": "knhip::pq_scan_v2_kernel<true, 2, false>", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": pmc_traffic(a, world),
                "algorithmic_bytes_per_launch": scan_bytes, "ms_per_launch": round(scan_ms, 3),
                "stage_ms_per_step": {n: round(prof["ms"][i] / a.steps, 3) for i, n in
                                      enumerate(["coarse", "group", "lut", "scan", "merge", "other", "scan_rank0"])}}

    # ---------------------------------------------------------------- CPU baseline (rank 0, N == 1)
    cpu = None
    if rank == 0 and world == 1 and a.cpu_queries > 0:
        cpu = cpu_baseline(a, built, vectors, xq, I, log)

    if rank == 0:
        out = {
            "metric": f"QPS at recall@{a.k}>=0.95, IVF-PQ {a.nb // 1_000_000}M x d={a.d} batch={a.nq // 1000}k",
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "recall_at_10": round(rec, 4), "recall_gate_met": bool(rec >= 0.95),
            "config": {"workload": f"IVF-PQ m={a.m} nbits=8, {a.nb} x d={a.d} fp32, nlist={a.nlist} "
                                   f"nprobe={a.nprobe}, batch={a.nq}, k={a.k}, refine_k={a.refine_k} (fp32 re-rank)",
                       "data_generator": f"{a.data} ncenter={a.ncenter} sigma={a.sigma} latent={a.latent} seed=42/44",
                       "parallelism": f"list-sharded x{world}" if world > 1 else "single GPU",
                       "build_s": round(build_s, 1)},
            "roofline": roofline,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def pmc_traffic(a, world):
    """HBM bytes per launch of the scan kernel from the PMC counters.  Counters cannot be collected
    inside a timed run: they come from separate rocprofv3 --pmc passes of this same command
    (tools/profile_bench.sh), stored under profiles/; returned only if the workload matches."""
    path = os.path.join(ROOT, "profiles", "bench_pmc_traffic.json")
    try:
        t = json.load(open(path))
    except Exception:
        return None
    key = f"nb={a.nb},nlist={a.nlist},nprobe={a.nprobe},nq={a.nq},m={a.m},refine_k={a.refine_k},gpus={world}"
    return t.get(key, {}).get("hbm_bytes_per_launch")


def cpu_baseline(a, built, vectors, xq, I_gpu, log):
    """Time the reference's FAISS (oracle/_ref) -- or the oracle port where _ref cannot load -- on
    the host cores over a bounded sample of the same batch, Knowhere-style (one query per task)."""
    from oracle import binding as ob  # checker / baseline only
    nth = os.cpu_count() or 1
    nqs = min(a.cpu_queries, a.nq)
    t0 = time.time()
    ix = built.export(ob.IndexData)
    q = xq[:nqs].cpu().numpy()
    kbase = a.refine_k if a.refine_k > 0 else a.k
    kind = "port"
    try:
        if ob.Ref.available():
            kind = "reference"
    except Exception:
        kind = "port"
    if kind == "reference":
        ref = ob.Ref()
        h = ref.from_data(ix)
        log(0, f"cpu baseline: reference index rebuilt on host in {time.time() - t0:.1f}s")
        t1 = time.time()
        Dc, Ic = ref.search(h, q, kbase, a.nprobe, nthreads=nth)
        dt = time.time() - t1
        cores = nth
    else:
        port = ob.Port()
        ix.use_precomputed_table = 1
        ix.precomputed_table = port.pq_precompute_table(ix.d, ix.M, 8, ix.centroids, ix.pq_centroids)
        t1 = time.time()
        Dc, Ic = port.search(ix, q, kbase, a.nprobe)
        dt = time.time() - t1
        cores = 1
    if a.refine_k > 0 and vectors is not None:
        port = ob.Port()
        t2 = time.time()
        # gather only the candidate rows (the 51 GB base never leaves HBM)
        uniq, inv = np.unique(Ic[Ic >= 0], return_inverse=True)
        rows = vectors[torch.from_numpy(uniq).to(vectors.device)].cpu().numpy()
        remap = np.full(Ic.shape, -1, np.int64)
        remap[Ic >= 0] = inv
        Dr, Ir = port.refine(ob.L2, rows, q, remap, a.k)
        Ir = np.where(Ir >= 0, uniq[np.clip(Ir, 0, None)], -1)
        dt += (time.time() - t2) / (cores if kind == "reference" else 1)
        Ic = Ir
    else:
        Ic = Ic[:, :a.k]
    agree = float((Ic == I_gpu[:nqs].cpu().numpy()).mean())
    log(0, f"cpu baseline ({kind}): {nqs} queries in {dt:.2f}s on {cores} thread(s); "
           f"id agreement with the GPU result {agree:.4f}")
    return {"value": round(nqs / dt, 2), "unit": "queries/s", "cores": cores, "kind": kind,
            "sample": f"first {nqs} of the {a.nq} queries, same index bytes, one query per task, "
                      f"scalar (SIMDLevel::NONE) FAISS build, omp=1 inside each task",
            "gpu_id_agreement_on_sample": round(agree, 4)}


if __name__ == "__main__":
    main()
