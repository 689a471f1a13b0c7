#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r2h_bench_c3.log 2>&1; tail -4 gpurun_out/r2h_bench_c3.log | cut -c1-3000
timeout 600 python bench.py --config C2 --steps 10 --warmup 3 > gpurun_out/r2h_bench_c2.log 2>&1; tail -4 gpurun_out/r2h_bench_c2.log | cut -c1-3000
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --nb 10000000 --nlist 4096 --nprobe 64 --steps 3 --warmup 1 > gpurun_out/r2h_bench_2rank.log 2>&1; tail -5 gpurun_out/r2h_bench_2rank.log | cut -c1-2500
timeout 600 python bench.py --nb 10000000 --nlist 4096 --nprobe 64 --steps 3 --warmup 1 --cpu-queries 0 > gpurun_out/r2h_bench_1rank_10m.log 2>&1; tail -2 gpurun_out/r2h_bench_1rank_10m.log | cut -c1-600
