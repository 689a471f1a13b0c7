#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2j_pytest.log 2>&1; tail -3 gpurun_out/r2j_pytest.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r2j_bench_c3.log 2>&1; tail -3 gpurun_out/r2j_bench_c3.log | cut -c1-1500
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --nb 10000000 --nlist 4096 --nprobe 64 --steps 3 --warmup 1 > gpurun_out/r2j_bench_2rank.log 2>&1; tail -3 gpurun_out/r2j_bench_2rank.log | cut -c1-700
