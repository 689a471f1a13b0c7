#!/usr/bin/env python
"""bench.py -- contract benchmark.

Default (= BASELINE.json's metric, config "C3"): QPS at recall@10 >= 0.95, IVF-PQ m=32 x 8 bit,
100M x d=128 fp32, nlist=16384, nprobe=128, batch = 10k queries, k=10, on 1/2/4/8 MI355X.
`--config C2` (IVF-Flat L2 10M x 128, nlist 4096, nprobe 64) and `--config C5` (IVF-SQ8 IP 100M x 768 int8-valued,
nlist 65536, nprobe 256) run BASELINE.json's other single-GPU configurations through the same harness.

One "step" = one Search() of the whole 10k-query batch: coarse quantizer -> (PQ tables) -> per-list scan ->
per-query merge [-> exact fp32 re-rank of the top-`refine_k` PQ candidates (Knowhere's `refine`, IndexRefine)]
-> top-10.  `value` is measured with queries, index and raw vectors resident in HBM when the timed region starts;
nothing is cached between steps.  The same step driven across the HOST boundary (pageable host queries in, host
results out: what IndexNode::Search does, H2D + D2H inside the timed region) is timed too and reported as
`host_boundary` -- SURVEY.md 8(d) quotes the reference's GPU path that way.

N > 1 (launched by torch.distributed.run, one rank per GPU): STRONG scaling on the same index.  Inverted lists
are partitioned across ranks (size-balanced); every rank builds, encodes and keeps ONLY the rows of the lists it
owns (codes and raw vectors), the coarse quantizer and codebooks are trained on rank 0 and broadcast.  Per step
the coarse quantizer is sharded by queries (one packed all-gather of the (nq, nprobe) assignment), every rank
scans the probes it owns; one packed all-gather of the per-rank (distance, id) partial top-`refine_k` over RCCL/xGMI
+ a device merge gives the global PQ candidates, every rank computes the exact distances of the candidates whose raw
vectors it holds, one all-gather of the distance arrays and ONE selection yield the final top-10
-- bit-identical to the single-GPU result, candidates tied at a k-th distance included (the reference's first-come
admission rule is applied once, after the merge: knowhere_amd/sharded.py::search_sharded / refine_sharded).

`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run (one rank per GPU on
127.0.0.1) and fails loudly when fewer than N GPUs are visible; `n_gpus` in the line is the size of the process group
that really ran.  The timed loop rotates through `--query-batches` (3) distinct query batches, so nothing a step learns
(the selectivity guard takes its decision from the previous batch's counters) is learnt from the batch it is applied to.
The default run (C3, N = 1) also runs C2 through the same harness and reports it under "extra_configs".

Extra JSON objects: "roofline" (dominant kernel; for the ADC scan the binding unit is the LDS gather, HBM
fractions are kept beside it) and "cpu_baseline" (the reference's own FAISS, AVX2 dynamic-dispatch build where it
loads, timed on this box's host cores on a bounded sample of the same workload, checked bitwise against the GPU).
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

# /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0                 # HBM3E spec peak
LDS_PEAK_GBPS = 256 * 256 * 2.4        # 256 B/clk/CU (conflict-free ds_read_b64/b128) x 256 CUs x 2.4 GHz
VALU_PEAK_TFLOPS = 157.3               # fp32 vector peak (FMA = 2 flop)
MFMA_F32_PEAK_TFLOPS = 157.3           # fp32 matrix peak (coarse quantizer, fp32 row prefilter)
MFMA_F16_PEAK_TFLOPS = 2500.0          # dense f16 / bf16 matrix peak (SQ8 prefilter)
MFMA_F16_SUSTAINED_TFLOPS = 1850.0  # measured: the 32x32x16 f16 / bf16 instruction back to back on every pipe for 1 - 6 ms (see roofline.algorithmic)
MFMA_I8_PEAK_TOPS = 1024 * 2048 * 2.4 / 1e3  # v_mfma_i32_16x16x64_i8: 32768 ops per 16 cycles and SIMD x 1024 SIMDs x 2.4 GHz
#                                        = 5033 TOP/s (the guide lists >= 3944 TOP/s measured for this instruction)

CONFIGS = {
    # BASELINE.json configs[0]: the reference's own CPU-runnable plumbing case, here on the GPU through the same ABI
    "C1": dict(kind="flat", metric="l2", nb=100_000, d=128, nlist=0, nprobe=1, nq=1000, k=10, m=0,
               refine_k=0, data="uniform", train_per_centroid=0, niter=0),
    # the same scan at a size where the matrix pipe matters: 1M rows, a 10k-query batch (north_star: BruteForce Search())
    "C1m": dict(kind="flat", metric="l2", nb=1_000_000, d=128, nlist=0, nprobe=1, nq=10000, k=10, m=0,
                refine_k=0, data="uniform", train_per_centroid=0, niter=0),
    # BASELINE.json configs[1]
    "C2": dict(kind="ivfflat", metric="l2", nb=10_000_000, d=128, nlist=4096, nprobe=64, nq=10000, k=10, m=0,
               refine_k=0, data="mixture", train_per_centroid=256, niter=10),
    # BASELINE.json configs[2]: the metric's configuration
    "C3": dict(kind="ivfpq", metric="l2", nb=100_000_000, d=128, nlist=16384, nprobe=128, nq=10000, k=10, m=32,
               refine_k=100, data="mixture", train_per_centroid=256, niter=10),
    # BASELINE.json configs[4]; 65536 x 768 centroids: fewer training points / iterations keep the build in minutes
    "C5": dict(kind="ivfsq8", metric="ip", nb=100_000_000, d=768, nlist=65536, nprobe=256, nq=10000, k=10, m=0,
               refine_k=0, data="int8", train_per_centroid=32, niter=10),
    # a SLICE of C5 for the default driver line (the full configuration builds for minutes and its 76.8 GB of codes do not
    # fit the host for the reference cross-check): a tenth of the rows and of the lists -- the same mean list length
    # (1526 rows), nprobe and row size, hence the same code bytes per query; the coarse stage is ten times smaller
    "C5s": dict(kind="ivfsq8", metric="ip", nb=10_000_000, d=768, nlist=6554, nprobe=256, nq=10000, k=10, m=0,
                refine_k=0, data="int8", train_per_centroid=32, niter=10),
    # the metric's configuration on data where the selectivity guard takes other decisions (driver-visible data dependence):
    # uniform [0, 100) (the reference's own test fixture; PQ32 cannot separate it: no recall gate) and mixture components
    # of intrinsic dimension 16
    "C3u": dict(kind="ivfpq", metric="l2", nb=100_000_000, d=128, nlist=16384, nprobe=128, nq=10000, k=10, m=32,
                refine_k=100, data="uniform", train_per_centroid=256, niter=10),
    "C3l": dict(kind="ivfpq", metric="l2", nb=100_000_000, d=128, nlist=16384, nprobe=128, nq=10000, k=10, m=32,
                refine_k=100, data="mixture", train_per_centroid=256, niter=10, latent=16),
}
CONFIG_NOTE = {"C5s": "slice of C5: 1/10 of the rows and lists, same list length / nprobe / row size",
               "C3u": "C3 on uniform [0, 100) data", "C3l": "C3 on mixture components of intrinsic dimension 16"}
KINDS = {"flat": kidx.BRUTE_FORCE, "ivfflat": kidx.IVF_FLAT, "ivfpq": kidx.IVF_PQ, "ivfsq8": kidx.IVF_SQ8}
KIND_LABEL = {"flat": "BruteForce (FLAT)", "ivfflat": "IVF-Flat", "ivfpq": "IVF-PQ", "ivfsq8": "IVF-SQ8"}
SHAPE_KEYS = ("nb", "d", "nlist", "nprobe", "nq", "k", "m", "refine_k", "train_per_centroid", "niter")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3", choices=sorted(CONFIGS), help="BASELINE.json configuration")
    for name, typ in (("nb", int), ("d", int), ("nlist", int), ("nprobe", int), ("nq", int), ("k", int), ("m", int),
                      ("refine_k", int), ("train_per_centroid", int), ("niter", int)):
        ap.add_argument("--" + name.replace("_", "-"), type=typ, default=None, help="override the configuration")
    ap.add_argument("--data", default=None, choices=["mixture", "uniform", "int8"])
    ap.add_argument("--sigma", type=float, default=0.35)
    ap.add_argument("--ncenter", type=int, default=0, help="mixture components (0 = nb/160 rounded to a power of two)")
    ap.add_argument("--latent", type=int, default=0, help="intrinsic dimension of a component (0 = isotropic)")
    ap.add_argument("--gt-queries", type=int, default=1000, help="queries used for the recall measurement")
    ap.add_argument("--cpu-queries", type=int, default=-1,
                    help="queries of the CPU baseline sample (-1 = 32 per host thread, 0 = skip)")
    ap.add_argument("--query-batches", type=int, default=3, help="distinct query batches the timed loop rotates through")
    ap.add_argument("--extra", default="auto",
                    help="further configurations run after the main one, one JSON line each: auto (= C1, C1m, C2, "
                         "C5s, C3u, C3l for the default C3 run on one GPU), none, or a comma-separated list")
    ap.add_argument("--dry-launch", action="store_true",
                    help="only bring the process group up and print its size (tests the --gpus N self-launch without a GPU)")
    ap.add_argument("--host-steps", type=int, default=-1, help="steps of the host-boundary timing (-1 = --steps, 0 = skip)")
    ap.add_argument("--backend", default=None, help="nccl (default for N>1) | gloo (single-GPU debugging)")
    ap.add_argument("--coarse-mode", default="auto", choices=["auto", "replicate", "shard"],
                    help="N > 1: coarse quantizer replicated on every rank (no collective) or sharded by queries (one "
                         "all-gather of the assignment); auto = shard")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    a.overridden = [key for key in SHAPE_KEYS if getattr(a, key, None) is not None] + (["data"] if a.data else [])
    apply_config(a, a.config, keep_overrides=True)
    return a


def apply_config(a, name, keep_overrides):
    cfg = dict(CONFIGS[name])
    user_latent = getattr(a, "latent", 0) or 0
    for key in list(cfg):
        if key == "latent":
            continue
        v = getattr(a, key, None)
        if keep_overrides and v is not None:
            cfg[key] = v
    cfg["latent"] = user_latent if (keep_overrides and user_latent) else cfg.get("latent", 0)
    for key, v in cfg.items():
        setattr(a, key, v)
    a.config = name
    return a


def self_launch(a):
    """`python bench.py --gpus N` with no launcher around it: become N ranks under torch.distributed.run (one per GPU,
    rendezvous on 127.0.0.1).  Fewer than N visible GPUs is an error, never a silent 1-GPU run."""
    import socket
    gloo = (a.backend or "nccl") == "gloo"
    if not a.dry_launch:
        ndev = torch.cuda.device_count()
        shared = os.environ.get("KNHIP_ALLOW_SHARED_GPU") == "1"
        if ndev < a.gpus and not (shared and gloo and ndev > 0):
            sys.exit(f"bench.py --gpus {a.gpus}: only {ndev} GPU(s) visible; refusing to report an N-GPU number from fewer "
                     f"devices (single-GPU debugging of the N-rank path: --backend gloo with KNHIP_ALLOW_SHARED_GPU=1)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["KNHIP_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] self-launch:", " ".join(cmd), file=sys.stderr, flush=True)
    os.execvpe(cmd[0], cmd, env)


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)  # (does not return)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if a.dry_launch:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group(backend=a.backend or "gloo", rank=rank, world_size=world)
            n = dist.get_world_size()
            t = torch.ones(1)
            dist.all_reduce(t)
            assert int(t.item()) == n == world
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_launch": True, "n_gpus": world, "self_launched": os.environ.get("KNHIP_BENCH_SELF_LAUNCHED") == "1"}),
                  flush=True)
        return
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
        assert comm.world == world == a.gpus == dist.get_world_size() and comm.backend == backend, (comm.world, world, comm.backend)
        if backend == "nccl":
            assert ndev >= world, f"{world} RCCL ranks need {world} GPUs (found {ndev})"
        else:
            assert ndev >= world or os.environ.get("KNHIP_ALLOW_SHARED_GPU") == "1", \
                f"{world} ranks on {ndev} GPU(s): set KNHIP_ALLOW_SHARED_GPU=1 for single-GPU debugging over gloo"
    out = run_config(a, rank, world, dev, dev_id, comm)
    if rank == 0:
        # (a copy for the log as soon as it exists: should a later configuration take the process down, the headline
        # is still on stderr; stdout gets it LAST so that a tail of stdout always ends with it)
        print("[bench-headline] " + json.dumps(slim_line(out)), file=sys.stderr, flush=True)
    extra = []
    if a.extra == "auto":
        extra = ["C1", "C1m", "C2", "C5s", "C3u", "C3l"] if (a.config == "C3" and world == 1 and not a.overridden) else []
    elif a.extra != "none":
        extra = [e for e in a.extra.split(",") if e]
    full = {a.config: out}
    for name in extra:
        assert name in CONFIGS, name
        import copy
        b = apply_config(copy.copy(a), name, keep_overrides=False)
        b.ncenter = 0
        if name in ("C3u", "C3l"):
            b.cpu_queries = 0  # (the same index kind and kernels as the main run: no second reference leg)
        torch.cuda.empty_cache()
        try:
            sub = run_config(b, rank, world, dev, dev_id, comm)
        except Exception as e:  # an extra configuration never costs the headline
            log(rank, f"extra configuration {name} failed: {e!r}")
            continue
        if rank == 0:
            full[name] = sub
            # one short line per extra configuration, BEFORE the headline
            print(json.dumps(slim_line(sub, extra=True)), flush=True)
    if rank == 0:
        write_full(full)
        print(json.dumps(slim_line(out)), flush=True)
    # C5 at its stated size (100M x 768: 80 GB of index, ~4 minutes with its build) runs AFTER the headline has been printed
    # and the headline is printed once more behind it: whatever happens to the long configuration, the last complete JSON
    # line of stdout is the headline
    late = ["C5"] if (a.extra == "auto" and a.config == "C3" and world == 1 and not a.overridden) else []
    for name in late:
        import copy
        b = apply_config(copy.copy(a), name, keep_overrides=False)
        b.ncenter = 0
        out_keep = slim_line(out) if rank == 0 else None
        torch.cuda.empty_cache()
        try:
            sub = run_config(b, rank, world, dev, dev_id, comm)
        except Exception as e:
            log(rank, f"extra configuration {name} failed: {e!r}")
            continue
        if rank == 0:
            full[name] = sub
            write_full(full)
            print(json.dumps(slim_line(sub, extra=True)), flush=True)
            print(json.dumps(out_keep), flush=True)
    if world > 1:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------
# What is printed.  The driver keeps the last ~10 kB of stdout and parses the LAST JSON line: the headline must be short
# (round 5's 21 kB line with every extra configuration nested inside it did not parse).  stdout = one line of < 1 kB per
# extra configuration, then the headline (< 6 kB) as the last line; everything else (prose notes, per-unit PMC fractions,
# thread sweeps) goes to gpurun_out/bench_full.json and, where it explains a number, to DESIGN.md section 5.
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "algorithmic", "executed", "filter_form",
                 "algorithmic_bytes_per_launch", "ms_per_launch", "traffic", "hbm_measured_frac", "hbm_algorithmic_frac",
                 "stage_ms_per_step", "mscan")
HOST_KEYS = ("value", "unit", "ms_per_step", "steps", "identical_to_device_path", "entry_point")
CPU_KEYS = ("value", "unit", "cores", "kind", "simd", "sample", "gpu_final_ids_equal", "gpu_final_distances_bit_equal")


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def slim_line(out, extra=False):
    """the printed form of a configuration's result: the contract's keys + roofline + host boundary + CPU baseline,
    without prose"""
    if out is None:
        return None
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype", "data", "value_form", "recall_at_10", "recall_gate_met", "result_crc32"))
    cfg = out.get("config", {})
    line["config"] = _pick(cfg, ("name", "workload", "data_generator") + (() if extra else ("parallelism",)))
    r = out.get("roofline")
    if r:
        rk = ("bound", "kernel", "achieved", "peak", "unit", "frac", "algorithmic", "ms_per_launch", "traffic",
              "hbm_measured_frac") if extra else ROOFLINE_KEYS
        line["roofline"] = _pick(r, rk)
        st = line["roofline"].get("stage_ms_per_step")
        if st:
            line["roofline"]["stage_ms_per_step"] = {k: v for k, v in st.items() if k != "note"}
        if r.get("unit_busy") and not extra:
            line["roofline"]["unit_busy"] = _pick(r["unit_busy"], ("mfma_pipe_busy", "lds_array_busy", "valu_issue_busy",
                                                                   "wave_cycles_waiting", "commit"))
        if r.get("coarse_stage") and not extra:
            line["roofline"]["coarse_stage"] = _pick(r["coarse_stage"], ("kernel", "algorithmic_flops", "flops", "ms",
                                                                         "achieved_TFLOPs_whole_stage", "peak"))
    if out.get("host_boundary"):
        line["host_boundary"] = _pick(out["host_boundary"], ("value", "ms_per_step") if extra else HOST_KEYS)
    if out.get("cpu_baseline"):
        c = _pick(out["cpu_baseline"], ("value", "cores", "kind", "gpu_final_ids_equal", "gpu_final_distances_bit_equal")
                  if extra else CPU_KEYS)
        if "simd" in c:
            c["simd"] = c["simd"].split(" (")[0]
        line["cpu_baseline"] = c
    if out.get("multi_gpu") and not extra:
        m = out["multi_gpu"]
        line["multi_gpu"] = {"backend": m.get("backend"), "world": m.get("world"), "coarse": m.get("coarse"),
                             "collectives_per_step": m.get("collectives_per_step"),
                             "ranks": [_pick(rk_, ("rank", "collective_ms_per_step", "step_ms", "filter_ms", "scan_bytes_per_step"))
                                       for rk_ in m.get("ranks", [])]}
    return line


def write_full(full):
    """every configuration's complete object (notes included) -> gpurun_out/bench_full.json (scratch; copied to profiles/
    by hand for the runs that are cited)"""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "bench_full.json"), "w") as f:
            json.dump(full, f, indent=1)
    except Exception as e:
        print(f"[bench] could not write gpurun_out/bench_full.json: {e!r}", file=sys.stderr)


def run_config(a, rank, world, dev, dev_id, comm):
    """build the configuration's index, gate recall, time the steps, collect roofline / host boundary / CPU baseline;
    returns the JSON object (rank 0) -- everything it allocated is released before it returns"""
    kind = KINDS[a.kind]
    metric = kidx.L2 if a.metric == "l2" else kidx.IP
    refine = a.refine_k > 0

    # ---------------------------------------------------------------- build
    t_build = time.time()
    if a.ncenter <= 0:
        a.ncenter = 1 << max(4, int(round(np.log2(max(a.nb / 160.0, 16.0)))))
    spec = kb.DataSpec(a.nb, a.d, kind=a.data, seed=42, ncenter=a.ncenter, sigma=a.sigma, latent=a.latent)
    cen = cb = sq = None
    if kind == kidx.BRUTE_FORCE:
        # C1: rows split contiguously over the ranks (id offset = first row), exact scan, same merge of the partials
        lo, hi = a.nb * rank // world, a.nb * (rank + 1) // world
        built = kb.BuiltIndex()
        built.kind, built.metric, built.d = kind, metric, a.d
        base = spec.rows(lo, hi, dev).contiguous()
        g = kidx.GpuIndex(kind, metric, a.d, device=dev_id)
        g.add_vectors_device(base, id_offset=lo)
        built.base = base
    elif world > 1:
        # rank 0 trains; centroids and codec parameters are broadcast so every shard quantises identically
        if rank == 0:
            # (keep_vectors as at N = 1: with refine the single-GPU build trains on the resident rows, else on a sample --
            # the same input here, so that every N searches the SAME index and `result_crc32` can be compared across N)
            tmp = kb.build_ivf(spec, kind, metric, a.nlist, a.m, device=str(dev), train_only=True, keep_vectors=refine,
                               train_per_centroid=a.train_per_centroid, niter=a.niter, verbose=a.verbose)
            cen, cb, sq = tmp.centroids, tmp.codebooks, tmp.sq_trained
            tmp.gpu.close()
            tmp = None
            torch.cuda.empty_cache()
        else:
            cen = torch.empty((a.nlist, a.d), device=dev)
            cb = torch.empty((a.m, 256, a.d // a.m), device=dev) if kind == kidx.IVF_PQ else None
            sq = torch.empty((2 * a.d,), device=dev) if kind == kidx.IVF_SQ8 else None
        cen = comm.broadcast(cen)
        cb = comm.broadcast(cb) if cb is not None else None
        sq = comm.broadcast(sq) if sq is not None else None
        # list ownership needs the list sizes: one cheap assignment pass per rank over its slice of the rows, summed
        ga = kidx.GpuIndex(kidx.IVF_FLAT, metric, a.d, nlist=a.nlist, device=dev_id)
        ga.set_coarse_device(cen)
        sizes = sharded.global_list_sizes(comm, spec, lambda x: ga.coarse_search_device(x.contiguous(), 1)[1][:, 0],
                                          a.nlist, rank, world, dev)
        ga.close()
        owned = sharded.partition_lists(sizes, world)[rank]
        built = kb.build_ivf(spec, kind, metric, a.nlist, a.m, device=str(dev), centroids=cen, codebooks=cb,
                             sq_trained=sq, owned_lists=owned, verbose=a.verbose and rank == 0, keep_vectors=refine)
        g = built.to_gpu_index(device=dev_id)
    else:
        built = kb.build_ivf(spec, kind, metric, a.nlist, a.m, device=str(dev), verbose=a.verbose,
                             keep_vectors=refine, train_per_centroid=a.train_per_centroid, niter=a.niter)
        g = built.to_gpu_index(device=dev_id)
    vectors = getattr(built, "vectors", None)        # raw fp32 rows this rank holds (row r <-> id vector_ids[r])
    vector_ids = getattr(built, "vector_ids", None)  # None: row r <-> id r
    # distinct query batches for the timed loop (batch 0 is the one recall, parity and the CPU leg are measured on)
    xqs = [kb.queries(spec, a.nq, dev, seed=44 + 2 * b) for b in range(max(1, a.query_batches))]
    xq = xqs[0]
    torch.cuda.synchronize()
    build_s = time.time() - t_build
    log(rank, f"build {build_s:.1f}s {built.timings} index {g.device_bytes / 1e9:.2f} GB/rank"
              f"{', raw vectors %.1f GB/rank' % (vectors.numel() * 4 / 1e9) if vectors is not None else ''}, "
              f"precomputed_table={g.uses_precomputed_table}")
    if getattr(built, "codes", None) is not None and built.codes.numel() > (16 << 30):
        # (C5: 76.8 GB of list-sorted codes: more than the host holds.  The reference leg runs on the SUB-INDEX the first
        # queries probe -- every list they visit, complete, the others empty: those queries get the whole index's answer)
        if rank == 0 and world == 1 and a.cpu_queries != 0:
            from oracle import binding as ob  # checker / baseline only
            nsub = 24
            _, kk = g.coarse_search_device(xq[:nsub].contiguous(), a.nprobe)
            uniq = torch.unique(kk[kk >= 0]).cpu().numpy()
            t_sub = time.time()
            built.sub_ix = built.export(ob.IndexData, lists=uniq)
            built.sub_nq = nsub
            log(rank, f"sub-index for the reference leg: {len(uniq)} lists of {a.nlist}, "
                      f"{sum(len(c) for c in built.sub_ix.list_codes) * built.codes.shape[1] / 1e9:.1f} GB on the host in "
                      f"{time.time() - t_sub:.1f}s")
        built.codes = None
        torch.cuda.empty_cache()
    kbase = a.refine_k if refine else a.k
    row_of_id = sharded.row_lookup(vector_ids, a.nb, dev) if (refine and vector_ids is not None) else None

    refine_ev = {"on": False, "ev": []}  # (event pairs around the re-rank: its stage time, itemised like the library's)

    def rerank(q, Ip):
        """exact fp32 re-rank of the PQ candidates whose raw vectors this rank holds"""
        if refine_ev["on"]:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = rerank_inner(q, Ip)
            e1.record()
            refine_ev["ev"].append((e0, e1))
            return out
        return rerank_inner(q, Ip)

    def rerank_inner(q, Ip):
        if row_of_id is None:
            return kidx.refine_device(metric, vectors, q, Ip, a.k)
        rows = sharded.ids_to_rows(Ip, row_of_id)          # -1 where the vector lives on another rank
        D, R = kidx.refine_device(metric, vectors, q, rows, a.k)
        return D, sharded.rows_to_ids(R, vector_ids)

    # N > 1: the coarse quantizer is sharded by QUERIES (each rank assigns nq / N of them, one all-gather of the
    # (nq, nprobe) assignment), the scan by LISTS (knhip_search_preassigned_device = IndexIVF::search_preassigned)
    # (auto = sharded: the replicated coarse stage -- 1.1 ms per rank at C3 -- is the first thing Amdahl charges a
    # list-sharded step for; the packed assignment is 15 MB per step at C3)
    coarse_replicated = kind == kidx.BRUTE_FORCE or a.coarse_mode == "replicate"

    def step(b=0):
        q = xqs[b % len(xqs)]
        if world > 1:
            # every rank contributes CANONICAL partials; the reference's admission rule at the k-th boundary is applied once,
            # after the merge, over all ranks' candidates (sharded.search_sharded): the single-GPU answer, ties included
            keys = cdis = None
            if kind != kidx.BRUTE_FORCE:
                if coarse_replicated:
                    cdis, keys = g.coarse_search_device(q, a.nprobe)
                else:
                    keys, cdis = sharded.sharded_coarse(comm, lambda lo, hi: g.coarse_search_device(q[lo:hi], a.nprobe),
                                                        a.nq, a.nprobe, device=dev)
            Dp, Ip = sharded.search_sharded(
                comm, metric, kbase, lambda kk: g.search_canonical_device(q, kk, a.nprobe, keys, cdis),
                lambda fl, can: g.tie_arrivals_device(q, fl, can, kbase, a.nprobe, keys, cdis,
                                                      key_base=(a.nb * rank // world) if kind == kidx.BRUTE_FORCE else 0),
                kind=kind)
            if not refine:
                return Dp, Ip
            # second stage: distances where the rows are (the others marked "not here"), one all-gather, ONE selection
            def held():
                rows = Ip if row_of_id is None else sharded.ids_to_rows(Ip, row_of_id)
                return kidx.refine_distances_device(metric, vectors, q, rows)
            return sharded.refine_sharded(comm, metric, a.k, Ip, held)
        Dp, Ip = g.search_device(q, kbase, a.nprobe)
        if refine:
            return rerank(q, Ip)
        return Dp, Ip

    # the same step across the host boundary (single GPU), through the entry points the node's Search() calls:
    # knhip_search / knhip_search_refine with HOST pointers (pageable queries in, host ids and distances out)
    xq_host = xq.cpu().numpy()
    raw_bf = None

    def host_step():
        if refine:
            return g.search_refine(raw_bf, xq_host, a.k, kbase, a.nprobe)
        return g.search(xq_host, a.k, a.nprobe)

    # ---------------------------------------------------------------- recall gate
    D, I = step()
    torch.cuda.synchronize()
    ngt = min(a.gt_queries, a.nq)
    _, gt = kb.ground_truth(spec, xq[:ngt], a.k, metric=metric, device=str(dev))
    hits = (I[:ngt].unsqueeze(2) == gt.unsqueeze(1)).any(dim=2).float().sum().item()
    rec = hits / (ngt * a.k)
    log(rank, f"recall@{a.k} = {rec:.4f} over {ngt} queries (refine_k={a.refine_k})")
    # a fingerprint of batch 0's (ids, distance bits): the same for every N when the same index is searched (the sharded
    # path returns the single GPU's answer bit for bit) -- lets the driver's N = 1, 2, 4, 8 lines be compared
    import zlib
    result_crc = "%08x" % (zlib.crc32(D.cpu().numpy().tobytes(), zlib.crc32(I.cpu().numpy().tobytes())) & 0xffffffff)

    # ---------------------------------------------------------------- timed region
    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            comm.barrier()
            torch.cuda.synchronize()

    for b in range(a.warmup):
        step(b)
    g.profile_enable(True)
    g.profile_reset()
    refine_ev["on"] = True
    barrier()
    t0 = time.perf_counter()
    for b in range(a.steps):
        step(a.warmup + b)
    barrier()
    dt = time.perf_counter() - t0
    refine_ev["on"] = False
    if world > 1:
        dt = comm.max_float(dt)
    prof = g.profile_get()
    prof["bench_refine_ms"] = sum(e0.elapsed_time(e1) for e0, e1 in refine_ev["ev"])
    refine_ev["ev"] = []
    g.profile_enable(False)
    multi = None
    if world > 1:
        # per-rank evidence for the scaling curve, taken in a separate pass of the same steps (the events around the
        # collectives stay out of the timed region): stage times of every rank, time inside the collectives
        comm.timed = True
        n0 = comm.ncollectives
        g.profile_enable(True)
        g.profile_reset()
        barrier()
        for b in range(a.steps):
            step(a.warmup + b)
        barrier()
        coll_ms = comm.collective_ms() / a.steps
        pr = g.profile_get()
        g.profile_enable(False)
        comm.timed = False
        mine = {"rank": rank, "device": dev_id, "collective_ms_per_step": round(coll_ms, 3),
                "collectives_per_step": (comm.ncollectives - n0) // a.steps,
                "filter_ms": round(pr["ms"][kidx._lib.STAGE_SCAN] / max(a.steps, 1), 3),
                "stage_ms_per_step": stage_table(a, kind, pr),
                "scan_bytes_per_step": pr["scan_bytes"] / a.steps}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        multi = {"backend": comm.backend, "world": comm.world, "coarse": "replicated" if coarse_replicated else "sharded by queries",
                 "collectives_per_step": mine["collectives_per_step"], "ranks": gathered}
    ms_per_step = dt / a.steps * 1e3
    qps = a.nq * a.steps / dt
    prof["bench_ms_per_step"] = ms_per_step

    host = None
    host_steps = a.steps if a.host_steps < 0 else a.host_steps
    if world == 1 and host_steps > 0:
        if refine:
            # the refine store knhip_search_refine reads: a BRUTE_FORCE index over the raw rows (a transient second copy here;
            # the node holds its rows in such an index from the start)
            raw_bf = kidx.GpuIndex(kidx.BRUTE_FORCE, metric, a.d, device=dev_id)
            raw_bf.add_vectors_device(vectors)
        Dh, Ih = host_step()
        same = bool((Ih == I.cpu().numpy()).all() and (Dh.view(np.uint32) == D.cpu().numpy().view(np.uint32)).all())
        t0 = time.perf_counter()
        for _ in range(host_steps):
            host_step()
        dth = (time.perf_counter() - t0) / host_steps
        host = {"value": round(a.nq / dth, 1), "unit": "queries/s", "ms_per_step": round(dth * 1e3, 3),
                "steps": host_steps, "h2d_bytes": int(xq_host.nbytes), "d2h_bytes": int(a.nq * a.k * 12),
                "identical_to_device_path": same,
                "entry_point": "knhip_search_refine" if refine else "knhip_search",
                "note": "SURVEY 8(d)'s form of the metric (one Search() of the batch across the HOST boundary): host pointers "
                        "through the C ABI entry point the IndexNode's Search() calls, pageable host queries in, host (ids, "
                        "distances) out; H2D, D2H, scratch and stream set-up inside the timed region; the same query batch "
                        "every step.  `value` of the line is the device-resident step, as the harness contract asks "
                        "(inputs resident in HBM when the timed region starts)"}
        if raw_bf is not None:
            raw_bf.close()
            raw_bf = None
            torch.cuda.empty_cache()

    roofline = make_roofline(a, kind, prof, world)

    # ---------------------------------------------------------------- CPU baseline (rank 0, N == 1)
    cpu = None
    if rank == 0 and world == 1 and a.cpu_queries != 0:
        try:
            cpu = cpu_baseline(a, kind, metric, built, vectors, xq, D, I, g, log)
        except Exception as e:  # the baseline is a reported side number: never lose the bench line over it
            log(0, f"cpu baseline failed: {e!r}")

    out = None
    if rank == 0:
        label = KIND_LABEL[a.kind]
        gate = a.config == "C3"
        out = {
            "metric": (f"QPS at recall@{a.k}>=0.95, {label} {a.nb // 1_000_000}M x d={a.d} batch={a.nq // 1000}k" if gate
                       else f"QPS, {label} {a.nb / 1e6:g}M x d={a.d} batch={a.nq // 1000}k"),
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "value_form": "device-resident step (queries, index, raw rows in HBM); SURVEY 8(d)'s host-pointer form = host_boundary",
            "recall_at_10": round(rec, 4), "recall_gate_met": bool(rec >= 0.95) if gate else None,
            "result_crc32": result_crc,
            "config": {"name": a.config,
                       "workload": f"{label}{' m=%d nbits=8' % a.m if kind == kidx.IVF_PQ else ''} {a.metric.upper()}, "
                                   f"{a.nb} x d={a.d} fp32{' (int8-valued)' if a.data == 'int8' else ''}, "
                                   + (f"nlist={a.nlist} nprobe={a.nprobe}, " if kind != kidx.BRUTE_FORCE else "")
                                   + f"batch={a.nq}, k={a.k}"
                                   + (f", refine: k_base={a.refine_k} first-stage candidates re-ranked in fp32 "
                                      f"(Knowhere refine_k = {a.refine_k / a.k:g})" if refine else ""),
                       "data_generator": f"{a.data} ncenter={a.ncenter} sigma={a.sigma} latent={a.latent} seed=42/44",
                       "query_batches": len(xqs),
                       "parallelism": f"list-sharded x{world}" if world > 1 else "single GPU",
                       "training": f"{a.niter} k-means iterations, {a.train_per_centroid} points per centroid",
                       "build_s": round(build_s, 1)},
            "roofline": roofline,
        }
        if world > 1:
            out["scaling_note"] = ("strong scaling of one index over N ranks; the hardware curve is whatever the driver's "
                                   "SCALE run records -- none had been measured when this code was written")
        if multi is not None:
            out["multi_gpu"] = multi
        if host is not None:
            out["host_boundary"] = host
        if cpu is not None:
            out["cpu_baseline"] = cpu
    # release everything this configuration held (the next one starts from an empty device)
    g.close()
    del built, vectors, vector_ids, xqs, xq, D, I
    torch.cuda.empty_cache()
    return out


def stage_table(a, kind, prof):
    """HIP-event time of every stage of a step (ms, mean over the timed steps), itemised so that the rows add up to the
    step: library stages (include/knhip.h, knhip_stage_times) + the re-rank this harness calls + what no event pair covers
    (launch gaps, the tie rule's 4-byte read-back, host time between the calls)."""
    S = kidx._lib
    n = max(a.steps, 1)
    ms = prof["ms"]
    prefilter = prof.get("mscan_queries", 0) > 0
    t = {"coarse": ms[S.STAGE_COARSE], "group": ms[S.STAGE_GROUP], "query_tables": ms[S.STAGE_LUT] + ms[S.STAGE_TABLES],
         ("sample" if prefilter else "scan_rank0"): ms[S.STAGE_SCAN_RANK0],
         ("filter" if prefilter else "scan"): ms[S.STAGE_SCAN],
         ("finish" if prefilter else "merge"): ms[S.STAGE_MERGE], "ties": ms[S.STAGE_TIES],
         "refine": ms[S.STAGE_REFINE] + prof.get("bench_refine_ms", 0.0), "other": ms[S.STAGE_OTHER]}
    out = {k: round(v / n, 3) for k, v in t.items()}
    if "bench_ms_per_step" in prof:
        out["not_in_any_stage"] = round(prof["bench_ms_per_step"] - sum(out.values()), 3)
        out["sum"] = round(prof["bench_ms_per_step"], 3)
    if prefilter and kind == kidx.IVF_PQ:
        out["note"] = ("query_tables = the filter's int8 / half tables + the selectivity guard; sample = tau_q from each "
                       "query's closest lists; ties = k + 1-th result, detection, read-back, rule; the work table (group) "
                       "is built on a side stream beside the sample pass: its time overlaps and is not part of the sum's "
                       "critical path")
    return out


def make_roofline(a, kind, prof, world):
    """dominant kernel = the list scan.  achieved = algorithmic work per launch / mean launch time (HIP events on the
    launch stream, inside the library; the rocprofv3 average of the same kernel is under profiles/)."""
    S = kidx._lib
    nlaunch = max(prof["launches"][S.STAGE_SCAN], 1)
    scan_ms = prof["ms"][S.STAGE_SCAN] / nlaunch
    # (IVF-PQ: the bulk launch over probes 1..nprobe-1; the rank-0 probes run in a separate dump + radix-select
    # phase whose time is stage "scan_rank0" and whose bytes are excluded here)
    scan_bytes = (prof["scan_bytes"] - prof["scan_bytes_rank0"]) / nlaunch
    sec = scan_ms * 1e-3
    stages = stage_table(a, kind, prof)
    hbm_algo = scan_bytes / sec / 1e9 if sec > 0 else 0.0
    common = {"tie_queries_per_step": round(prof.get("tie_queries", 0) / max(a.steps, 1), 2),
              "algorithmic_bytes_per_launch": scan_bytes, "ms_per_launch": round(scan_ms, 3), "traffic": None,
              "hbm_algorithmic_GBps": round(hbm_algo, 1), "hbm_algorithmic_frac": round(hbm_algo / HBM_PEAK_GBPS, 4),
              "hbm_measured_frac": None, "stage_ms_per_step": stages}

    def with_pmc(out):
        """attach the PMC evidence on file for the run's dominant kernel (or say that there is none)"""
        e = pmc_entry(a, world, out["kernel"])
        if e is None:
            out["traffic_note"] = "no PMC pass on file for this workload and kernel (profiles/bench_pmc_traffic.json)"
            return out
        out["traffic"] = e.get("hbm_bytes_per_launch")
        if out["traffic"] and sec > 0:
            out["hbm_measured_frac"] = round(out["traffic"] / sec / 1e9 / HBM_PEAK_GBPS, 4)
        out["traffic_source"] = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command at commit {e.get('commit')}"
        ub = unit_busy(e, sec)
        if ub:
            out["unit_busy"] = ub
        return out

    cf = prof["coarse_flops"] / max(prof["launches"][S.STAGE_COARSE], 1)
    if stages["coarse"] > 0 and cf > 0:
        # coarse_gemm.hip: where nlist >= 2048 and nlist / 32 (or / 16) >= 2 (nprobe + margin) the prefilter runs on the bf16 pipe --
        # two GEMM passes (group minima -> bound; candidates under the bound) of three bf16 products each; else one fp32 GEMM
        ncand = a.nprobe + max(32, a.nprobe // 4)
        g32, g16 = (a.nlist + 31) // 32, (a.nlist + 15) // 16  # (groups of 32 centroids, else of 16: coarse_bf16_group_rows)
        bf16 = (a.nlist >= 2048 and ((g32 >= 2 * ncand and g32 <= 4096) or (g16 >= 2 * ncand and g16 <= 4096))
                and os.environ.get("KNHIP_COARSE") is None)
        flops, peak = (6.0 * cf, MFMA_F16_PEAK_TFLOPS) if bf16 else (cf, MFMA_F32_PEAK_TFLOPS)
        common["coarse_stage"] = {"bound": "mfma", "kernel": "knhip::coarse_bf16_kernel (two passes x hi hi + hi lo + lo hi)"
                                  if bf16 else "knhip::coarse_gemm_kernel (fp32)",
                                  "algorithmic_flops": cf, "flops": flops, "ms": stages["coarse"],
                                  "achieved_TFLOPs_whole_stage": round(flops / (stages["coarse"] * 1e-3) / 1e12, 2),
                                  "peak": peak,
                                  "note": "whole stage: GEMM pass(es) + bound / select + exact re-rank + certificate"}
    if kind == kidx.IVF_PQ:
        # one 4-byte table lookup per code byte: the LDS gather is the unit that binds (round-1 PMC: HBM traffic
        # is 0.03-0.14 x the algorithmic bytes, LDS ~ busy); SURVEY 8(d)'s no-reuse HBM model is kept beside it
        if prof.get("mscan_queries", 0) > 0:
            # the matrix-core prefilter (pq_filter.hip) is the dominant kernel: one 2-byte table lookup per code byte and
            # query (16-byte entries hold 8 queries; one ds_read_b128 feeds one v_mfma_f32_16x16x32_f16 = 512 lookups:
            # the LDS gather at 4 cycles per wave-instruction and the matrix pipe at 16 cycles per instruction and SIMD
            # bind at the same 128 lookups/clk/CU), survivors recomputed by the exact finish
            steps = max(a.steps, 1)
            i8 = prof.get("pq_filter_form", 1) == 2
            l2s = "true" if a.metric == "l2" else "false"
            mscan = {"queries_per_step": prof["mscan_queries"] / steps,
                     "overflow_queries_per_step": prof["mscan_overflow_queries"] / steps,
                     "candidates_per_query": round(prof["mscan_candidates"] / max(prof["mscan_queries"], 1), 1),
                     "exact_recomputations_per_query": round(prof.get("mscan_recomputed", 0) / max(prof["mscan_queries"], 1), 1)}
            if prof.get("pq_filter_form", 1) == 3:
                # decode form (pq_decode.hip): rows decoded once per (list, <= 128 queries), dense half-precision contraction
                # on v_mfma_f32_32x32x16_f16: 128 MACs = 256 flop per (row, query) -- the ALGORITHMIC work of
                # dis = dis0 + psum - 2 <q, y> and what the kernel executes, up to the padding of its 32-query tiles
                pairs = scan_bytes / 32.0  # (row, query) pairs: 32 code bytes per row
                tf = pairs * 256.0 / sec / 1e12 if sec > 0 else 0.0
                return with_pmc(dict({"bound": "mfma", "kernel": f"knhip::pqd_kernel<{l2s}, false>", "achieved": round(tf, 1),
                                      "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_F16_PEAK_TFLOPS, 4),
                                      "algorithmic": {"flop_per_row_query": 256, "TFLOPs": round(tf, 1),
                                                      "frac": round(tf / MFMA_F16_PEAK_TFLOPS, 4),
                                                      "table_lookups_per_ns_per_cu": round(scan_bytes / sec / 1e9 / 256.0, 1)
                                                      if sec > 0 else None,
                                                      # what 1024 matrix pipes SUSTAIN for a launch of this length (the same
                                                      # instruction back to back, nothing else: tools/ubench/mfma_stream long,
                                                      # profiles/r06_ubench_mfma_sustained.log: 1.67 - 1.99 PFLOP/s at 1 - 6 ms)
                                                      "sustained_peak_measured_TFLOPs": MFMA_F16_SUSTAINED_TFLOPS,
                                                      "frac_of_sustained_peak": round(tf / MFMA_F16_SUSTAINED_TFLOPS, 4)},
                                      "filter_form": "decode: f16 contraction, <= 128 queries per unit", "mscan": mscan},
                                     **common))
            if i8:
                # integer form: 1 byte per (lookup, query): an entry holds 16 queries, one ds_read_b128 feeds one
                # v_mfma_i32_16x16x64_i8 = 1024 lookups; LDS gather and matrix pipe bind at 256 lookups/clk/CU
                lds = scan_bytes * 1.0 / sec / 1e9 if sec > 0 else 0.0
                kn = f"knhip::pqi_kernel<{l2s}>"
                note = ("achieved = 1 B x code bytes scanned / launch time (int8 table, 16 queries per ds_read_b128 = one "
                        "v_mfma_i32_16x16x64_i8); peak = 256 B/clk/CU x 256 CU x 2.4 GHz (the matrix pipe binds at the same "
                        "rate: 1024 lookups per 16 cycles and SIMD); the kernel is launched twice per step -- the scan and the "
                        "normally empty retry round (~5 us) -- so rocprofv3's per-kernel AVERAGE is half of ms_per_launch, its "
                        "MAX is the scan launch (profiles/r03_c3_pqi_stats.json)")
            else:
                lds = scan_bytes * 2.0 / sec / 1e9 if sec > 0 else 0.0
                kn = f"knhip::pqf_kernel<{l2s}, false>"
                note = ("achieved = 2 B x code bytes scanned / launch time (half-precision table, 8 queries per ds_read_b128 "
                        "= one v_mfma_f32_16x16x32_f16); peak = 256 B/clk/CU x 256 CU x 2.4 GHz (the matrix pipe binds at "
                        "the same rate: 512 lookups per 16 cycles and SIMD); the kernel is launched twice per step -- the scan "
                        "and the normally empty retry round -- so rocprofv3's per-kernel AVERAGE is half of ms_per_launch, its "
                        "MAX is the scan launch")
            # What binds the filter is the matrix pipe and -- at the very same rate -- the LDS gather that feeds it (one
            # ds_read_b128 per matrix instruction): `bound` names the matrix pipe, the LDS view of the same number is kept
            # beside it, the HBM side (measured traffic of the PMC passes on file) under hbm_*.  32 integer (half: floating
            # point) operations per (lookup, query): a 16x16x64 (16x16x32) instruction = 32768 (16384) for 1024 (512) pairs.
            tops = scan_bytes * 32.0 / sec / 1e12 if sec > 0 else 0.0
            peak = MFMA_I8_PEAK_TOPS if i8 else MFMA_F16_PEAK_TFLOPS
            return with_pmc(dict({"bound": "mfma", "kernel": kn, "achieved": round(tops, 1), "peak": round(peak, 1),
                         "unit": "TOP/s" if i8 else "TFLOP/s", "frac": round(tops / peak, 4),
                         "lds_gather": {"achieved_GBps": round(lds, 1), "peak_GBps": round(LDS_PEAK_GBPS, 1),
                                        "frac": round(lds / LDS_PEAK_GBPS, 4)},
                         "note": note, "filter_form": "int8 x 16 queries" if i8 else "half x 8 queries",
                         # `achieved` counts EXECUTED matrix operations (32 per lookup: the selector operand); the algorithm's
                         # own work is one addition per lookup
                         "executed": {"ops_per_lookup": 32, "T_ops": round(tops, 1)},
                         "algorithmic": {"ops_per_lookup": 1, "T_ops": round(tops / 32.0, 1), "frac": round(tops / 32.0 / peak, 4)},
                         "lookups_per_ns_per_cu": round(scan_bytes / sec / 1e9 / 256.0, 1) if sec > 0 else None,
                         "mscan": mscan}, **common))
        lds = scan_bytes * 4.0 / sec / 1e9 if sec > 0 else 0.0
        out = dict({"bound": "lds", "kernel": "knhip::pq_scan_q4_kernel<true, 2>", "achieved": round(lds, 1),
                    "peak": round(LDS_PEAK_GBPS, 1), "unit": "GB/s", "frac": round(lds / LDS_PEAK_GBPS, 4),
                    "note": "achieved = 4 B x code bytes scanned / launch time; peak = 256 B/clk/CU x 256 CU x 2.4 GHz"},
                   **common)
        # (the loop is co-limited by VALU issue, DESIGN 4.2: unit_busy carries the LDS-array and VALU-issue fractions of the
        # PMC passes on file for this kernel)
        return with_pmc(out)
    if prof.get("mscan_queries", 0) > 0:
        # MFMA prefilter + exact finish (mfma_scan.hip): the dominant kernel is a grouped (rows of a list) x (queries
        # that probe it) x d contraction; every unit streams its list once for up to 64 (fp32 rows) / 32 (SQ8) queries.
        code_size = a.d * 4 if kind == kidx.IVF_FLAT else a.d
        macs = scan_bytes / code_size * a.d
        stream = prof["mscan_stream_bytes"] / nlaunch
        stream_gbps = stream / sec / 1e9 if sec > 0 else 0.0
        if kind == kidx.IVF_FLAT and os.environ.get("KNHIP_MSCAN_FLAT") == "fp32":
            flop, peak, kname, unit_note = 2.0 * macs, MFMA_F32_PEAK_TFLOPS, "knhip::mscan_flat_kernel", \
                "2 flop per (row, query, dim) on v_mfma_f32_32x32x2_f32; peak = fp32 matrix peak"
        elif kind == kidx.IVF_FLAT:
            flop, peak, kname, unit_note = 6.0 * macs, MFMA_F16_PEAK_TFLOPS, "knhip::mscan_flatb_kernel", \
                "6 flop per (row, query, dim): both operands split into two bf16 terms, hi hi + hi lo + lo hi on " \
                "v_mfma_f32_32x32x16_bf16 (mfma_scan_bf16.hip); peak = dense bf16 matrix peak; the algorithmic 2 flop per " \
                "(row, query, dim) are a third of `achieved`"
        else:
            flop, peak, kname, unit_note = 4.0 * macs, MFMA_F16_PEAK_TFLOPS, "knhip::mscan_sq8_kernel", \
                "4 flop per (row, query, dim): the query operand is split into two halves (hi + lo) on " \
                "v_mfma_f32_32x32x16_f16; peak = dense f16 matrix peak"
        tf = flop / sec / 1e12 if sec > 0 else 0.0
        tf_algo = 2.0 * macs / sec / 1e12 if sec > 0 else 0.0  # (2 flop per (row, query, dim): the contraction itself)
        # HBM side: the measured traffic (PMC pass of this workload, profiles/bench_pmc_traffic.json) where there is one;
        # the bytes the units stream are an upper bound of it (units of one list share it through L2)
        e = pmc_entry(a, world, kname)
        traffic = e.get("hbm_bytes_per_launch") if e else None
        hbm_gbps = (traffic / sec / 1e9) if (traffic and sec > 0) else stream_gbps
        mfma_frac, hbm_frac = tf / peak, hbm_gbps / HBM_PEAK_GBPS
        steps = max(a.steps, 1)
        extra = {"mfma": {"achieved_TFLOPs": round(tf, 2), "peak_TFLOPs": peak, "frac": round(mfma_frac, 4)},
                 "executed": {"flop_per_mac": round(flop / max(macs, 1.0), 1), "TFLOPs": round(tf, 2)},
                 "algorithmic": {"flop_per_mac": 2, "TFLOPs": round(tf_algo, 2), "frac": round(tf_algo / peak, 4)},
                 "stream": {"bytes_per_launch": stream, "GBps": round(stream_gbps, 1),
                            "hbm_frac_upper_bound": round(stream_gbps / HBM_PEAK_GBPS, 4),
                            "note": "bytes the units read (L2 + HBM); `traffic` is the part that came from HBM "
                                    "(FETCH_SIZE pass), null when no PMC pass of this workload is on file"},
                 "mscan": {"queries_per_step": prof["mscan_queries"] / steps,
                           "overflow_queries_per_step": prof["mscan_overflow_queries"] / steps,
                           "candidates_per_query": round(prof["mscan_candidates"] / max(prof["mscan_queries"], 1), 1)}}
        # without a PMC pass of this workload the HBM share is unknown (the bytes the units stream include every L2 hit): the
        # matrix-core fraction is then the one number that is measured, and the stream rate is listed as an upper bound only
        if mfma_frac >= hbm_frac or not traffic:
            return with_pmc(dict({"bound": "mfma", "kernel": kname, "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s",
                                  "frac": round(mfma_frac, 4), "note": unit_note}, **common, **extra))
        return with_pmc(dict({"bound": "hbm", "kernel": kname, "achieved": round(hbm_gbps, 1), "peak": HBM_PEAK_GBPS,
                              "unit": "GB/s", "frac": round(hbm_frac, 4),
                              "note": "achieved = HBM bytes per launch (PMC traffic where on file, else the bytes the units "
                                      "stream) / launch time"}, **common, **extra))
    # exact row scans: lane = row, the queries of a work item share each row fetch -> VALU-bound by construction:
    # per (row, query, dim) L2 = sub, mul, add; IP = mul, add (+ SQ8: decode fma per (row, dim))
    code_size = a.d * 4 if kind in (kidx.IVF_FLAT, kidx.BRUTE_FORCE) else a.d
    if kind == kidx.BRUTE_FORCE:
        # every (query, row) pair of the rank's rows: SURVEY 8(d) C1 = 2 nq nb d flop, nq nb d 4 B query-major
        scan_bytes = float(a.nq) * (a.nb / world) * a.d * 4.0
        common["algorithmic_bytes_per_launch"] = scan_bytes
        hbm_algo = scan_bytes / sec / 1e9 if sec > 0 else 0.0
        common["hbm_algorithmic_GBps"], common["hbm_algorithmic_frac"] = round(hbm_algo, 1), round(hbm_algo / HBM_PEAK_GBPS, 4)
    if kind == kidx.BRUTE_FORCE and prof.get("pq_filter_form", 0) == 10:
        # the coarse quantizer's machinery over the base rows (knhip_api.hip::bf_mfma_batch): GEMM passes of three bf16 products
        # each + bound + exact re-rank; the base is cut into chunks of <= 131072 rows, the first chunk takes two passes (group
        # minima -> bound -> candidates), every other chunk ONE (the running k-th best is its bound).  ALGORITHMIC work
        # 2 nq nb d flop, executed 3 x (chunks + 1) / chunks x that
        algo = 2.0 * a.nq * (a.nb / world) * a.d
        tf_a = algo / sec / 1e12 if sec > 0 else 0.0
        nch = max(1, -(-int(a.nb / world) // 131072))
        ex = 3.0 * (nch + 1) / nch
        return dict({"bound": "mfma", "kernel": f"knhip::coarse_bf16_kernel (BRUTE_FORCE rows: {nch + 1} passes over {nch} chunk(s) x hi hi + hi lo + lo hi)",
                     "achieved": round(ex * tf_a, 2), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(ex * tf_a / MFMA_F16_PEAK_TFLOPS, 4),
                     "executed": {"flop_per_mac": round(2 * ex, 3), "TFLOPs": round(ex * tf_a, 2)},
                     "algorithmic": {"flop_per_mac": 2, "TFLOPs": round(tf_a, 2), "frac": round(tf_a / MFMA_F16_PEAK_TFLOPS, 4),
                                     "frac_of_fp32_matrix_peak": round(tf_a / MFMA_F32_PEAK_TFLOPS, 4)}}, **common)
    pairs_dims = scan_bytes / code_size * a.d
    flop = pairs_dims * (3.0 if a.metric == "l2" else 2.0)
    tf = flop / sec / 1e12 if sec > 0 else 0.0
    name = "knhip::flat_scan_kernel" if kind in (kidx.IVF_FLAT, kidx.BRUTE_FORCE) else "knhip::sq_scan_kernel"
    return dict({"bound": "valu", "kernel": name, "achieved": round(tf, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": round(tf / VALU_PEAK_TFLOPS, 4),
                 "note": "separately rounded sub/mul/add per (row, query, dim) counted as 1 flop each; "
                         "peak = fp32 vector peak (an FMA counts 2)"}, **common)


def pmc_key(a, world, kernel):
    """key of a workload in profiles/bench_pmc_traffic.json (a PMC pass belongs to one kernel path: the IVF-PQ prefilter is
    a different dominant kernel than the exact ADC scan)"""
    # (build=r3: indexes trained as the reference trains them -- 10 level-1 iterations, spherical k-means for the inner
    # product -- since round 3; the traffic of a kernel depends on the index it scans, entries of older builds are not used)
    key = (f"config={a.config},nb={a.nb},nlist={a.nlist},nprobe={a.nprobe},nq={a.nq},m={a.m},"
           f"refine_k={a.refine_k},gpus={world},data={a.data},latent={a.latent},build=r3")
    if "pqf_kernel" in kernel:
        key += ",pqf=1"
    if "pqi_kernel" in kernel:
        key += ",pqi=1"
    if "pqd_kernel" in kernel:
        key += ",pqd=1"
    return key


def pmc_entry(a, world, kernel):
    """The PMC evidence on file for this workload AND this dominant kernel, or None.  Counters cannot be collected inside a
    timed run: they come from separate rocprofv3 --pmc passes of this same command (tools/profile_bench.sh ->
    tools/pmc_traffic.py), stored under profiles/ with the commit they were taken at.  An entry whose kernel is not
    the kernel this run was dominated by is refused (a stale file must not dress up a different kernel)."""
    path = os.path.join(ROOT, "profiles", "bench_pmc_traffic.json")
    try:
        t = json.load(open(path))
    except Exception:
        return None
    e = t.get(pmc_key(a, world, kernel))
    if not e:
        return None
    short = kernel.split("::")[-1].split("<")[0]
    if short not in e.get("kernel", ""):
        return None
    return e


def unit_busy(e, sec):
    """busy fractions of the dominant kernel's units from the PMC passes on file: counter / (GRBM_GUI_ACTIVE / 8 XCDs =
    the kernel's cycles, from the same passes -- no assumed clock)"""
    sq = (e or {}).get("sq") or {}
    cyc = sq.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if cyc <= 0:
        return None
    # entries written since round 4 hold the LARGEST dispatch of every counter (the scan launch); older ones the mean over
    # the dispatches, half of them the empty retry launch: ratios are unaffected, absolutes only trusted for "max"
    stat = sq.get("_dispatch_stat", "mean")
    out = {"kernel_cycles": cyc, "dispatch_statistic": stat,
           "effective_clock_GHz_in_the_pmc_pass": round(cyc / sec / 1e9, 3) if (stat == "max" and sec > 0) else None,
           "source": sq.get("source"), "commit": e.get("commit")}
    if "SQ_LDS_IDX_ACTIVE" in sq:
        out["lds_array_busy"] = round(sq["SQ_LDS_IDX_ACTIVE"] / 256.0 / cyc, 4)
    if "SQ_LDS_BANK_CONFLICT" in sq and sq.get("SQ_LDS_IDX_ACTIVE"):
        out["lds_bank_conflict_frac"] = round(sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"], 5)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in sq:
        out["mfma_pipe_busy"] = round(sq["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc, 4)
    if "SQ_ACTIVE_INST_VALU" in sq:  # (quad-cycles a wave spent issuing VALU, summed over waves)
        out["valu_issue_busy"] = round(sq["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc, 4)
    if "SQ_WAIT_ANY" in sq and sq.get("SQ_WAVE_CYCLES"):
        out["wave_cycles_waiting"] = round(sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], 4)
    return out


def host_threads():
    """threads this process may really use: the affinity mask, capped by the cgroup CPU quota"""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(a, kind, metric, built, vectors, xq, D_gpu, I_gpu, g, log):
    """Time the reference's FAISS (oracle/_ref; the AVX2 dynamic-dispatch build when it loads, else the scalar
    build, else the oracle port) on the host cores over a bounded sample of the same batch, Knowhere-style (one
    query per task), on the same index bytes; compare its result with the GPU's bit for bit.
    Threads: the affinity mask capped by the cgroup quota; the timed build is swept over {n/4, n/2, n} threads with
    >= 32 queries per thread of the widest point on a warm pool, the best point is reported."""
    from oracle import binding as ob  # checker / baseline only
    nth = host_threads()
    if kind == kidx.BRUTE_FORCE:
        ix = ob.IndexData(ob.FLAT, metric, a.d)
        ix.base = built.base.cpu().numpy()
        code_bytes_per_query = float(a.nb) * a.d * 4
    else:
        ix = None
        if built.codes is None:
            ix = getattr(built, "sub_ix", None)
            if ix is None:
                log(0, "cpu baseline skipped: the list-sorted codes were released (index too large for the host leg)")
                return None
        code_bytes_per_query = None
    nqs = a.cpu_queries if a.cpu_queries > 0 else 32 * nth
    nqs = min(nqs, a.nq)
    sub = kind != kidx.BRUTE_FORCE and built.codes is None
    if sub:
        nqs = built.sub_nq  # (only these queries' lists are on the host)
    n_chk = min(nqs, 2048)  # the bitwise check against the scalar build runs on this prefix
    t0 = time.time()
    if ix is None:
        ix = built.export(ob.IndexData)
    q = xq[:nqs].cpu().numpy()
    kbase = a.refine_k if a.refine_k > 0 else a.k
    variants = []
    try:
        if ob.Ref.available("avx2"):
            variants.append(("reference", "avx2"))
        if ob.Ref.available():
            variants.append(("reference", "scalar"))
    except Exception:
        pass
    res = {}
    sweep = {}
    simd_used = None
    best_threads = nth
    for kindname, simd in variants:
        ref = ob.Ref(simd)
        h = ref.from_data(ix)
        ref.search(h, q[:min(nqs, 2 * nth)], kbase, a.nprobe, nthreads=nth)  # warm: page-in, thread pool
        if simd_used is None:
            # the timed build: sweep the thread count, keep the best rate
            simd_used = simd
            best = None
            for t in sorted({max(1, nth // 4), max(1, nth // 2), nth}, reverse=True):
                t1 = time.time()
                Dc, Ic = ref.search(h, q, kbase, a.nprobe, nthreads=t)
                dt_t = time.time() - t1
                sweep[t] = round(nqs / dt_t, 2)
                if best is None or dt_t < best[0]:
                    best = (dt_t, Dc, Ic)
                    best_threads = t
                if dt_t > 20.0:
                    break  # (bounded: the narrower points would take longer still)
            res[simd] = best
        else:
            t1 = time.time()
            Dc, Ic = ref.search(h, q[:n_chk], kbase, a.nprobe, nthreads=nth)
            res[simd] = (time.time() - t1, Dc, Ic)
        ref.free(h)
    if not res:
        port = ob.Port()
        if kind == kidx.IVF_PQ and metric == kidx.L2:
            ix.use_precomputed_table = 1
            ix.precomputed_table = port.pq_precompute_table(ix.d, ix.M, 8, ix.centroids, ix.pq_centroids)
        nqs = n_chk = min(nqs, 64)
        q = q[:nqs]
        t1 = time.time()
        Dc, Ic = port.search(ix, q, kbase, a.nprobe)
        res["port"] = (time.time() - t1, Dc, Ic)
        simd_used = "port"
        best_threads = 1
    log(0, f"cpu baseline: index on host in {time.time() - t0:.1f}s; threads available {nth}; sweep {sweep}; " +
        ", ".join(f"{s}: {len(r[1])} queries in {r[0]:.2f}s" for s, r in res.items()))
    dt = res[simd_used][0]
    # parity of the first stage (bit-exact bar) against the scalar reference where present, else the timed variant
    chk = "scalar" if "scalar" in res else simd_used
    Dg1, Ig1 = g.search_device(xq[:n_chk], kbase, a.nprobe)
    torch.cuda.synchronize()
    Dg1, Ig1 = Dg1.cpu().numpy(), Ig1.cpu().numpy()
    _, Dc, Ic = res[chk]  # every comparison below is against the scalar build where present (the parity bar)
    Dc, Ic = Dc[:n_chk], Ic[:n_chk]
    dist_equal = float((Dc.view(np.uint32) == Dg1.view(np.uint32)).mean())
    id_equal = float((Ic == Ig1).mean())
    cores = best_threads if simd_used != "port" else 1
    if a.refine_k > 0 and vectors is not None:
        port = ob.Port()
        t2 = time.time()
        uniq, inv = np.unique(Ic[Ic >= 0], return_inverse=True)  # gather only the candidate rows (the base stays in HBM)
        rows = vectors[torch.from_numpy(uniq).to(vectors.device)].cpu().numpy()
        remap = np.full(Ic.shape, -1, np.int64)
        remap[Ic >= 0] = inv
        Dr2, Ir2 = port.refine(metric, rows, q[:n_chk], remap, a.k)
        Ir2 = np.where(Ir2 >= 0, uniq[np.clip(Ir2, 0, None)], -1)
        # (the re-rank runs single-threaded on the check prefix; charged per query as if spread over the cores)
        dt += (time.time() - t2) / cores * (nqs / max(n_chk, 1))
        final_id = float((Ir2 == I_gpu[:n_chk].cpu().numpy()).mean())
        final_dist = float((Dr2.view(np.uint32) == D_gpu[:n_chk].cpu().numpy().view(np.uint32)).mean())
    else:
        final_id = float((Ic[:, :a.k] == I_gpu[:n_chk].cpu().numpy()).mean())
        final_dist = float((Dc[:, :a.k].view(np.uint32) == D_gpu[:n_chk].cpu().numpy().view(np.uint32)).mean())
    if code_bytes_per_query is None:
        # code bytes one query scans: its probed lists (the same count the GPU's roofline uses, per query)
        code_bytes_per_query = float(a.nprobe) * (a.nb / max(a.nlist, 1)) * ix.code_size
    return {"value": round(nqs / dt, 2), "unit": "queries/s", "cores": cores, "cores_available": nth,
            "thread_sweep_qps": {str(t): v for t, v in sorted(sweep.items())},
            "kind": "reference" if simd_used != "port" else "port",
            "simd": {"avx2": "AVX2 (dynamic dispatch build, cmake/libs/libfaiss.cmake:388-463 shape, -O3)",
                     "scalar": "none (SIMDLevel::NONE build, -O2)", "port": "none (oracle.c)"}[simd_used],
            "sample": f"first {nqs} of the {a.nq} queries ({nqs / max(cores, 1):.1f} per thread at the reported point), same "
                      f"index bytes{' (the sub-index of the lists these queries probe: the whole index does not fit the host)' if sub else ''}, "
                      f"one query per task, omp=1 inside each task, warm thread pool; bitwise check on the first {n_chk}",
            "code_GB_scanned_per_s": round(nqs / dt * code_bytes_per_query / 1e9, 2),
            "gpu_vs_scalar_reference_first_stage": {"checked_against": chk, "ids_equal": round(id_equal, 6),
                                                    "distances_bit_equal": round(dist_equal, 6)},
            "gpu_final_ids_equal": round(final_id, 6), "gpu_final_distances_bit_equal": round(final_dist, 6)}


if __name__ == "__main__":
    main()
