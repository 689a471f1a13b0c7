#!/bin/bash
# finish kernel with 8 row pieces in flight per candidate; SQ8 IP query operands prepared once per batch
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/r2x_pytest.log 2>&1; tail -4 gpurun_out/r2x_pytest.log | cut -c1-300
B="--cpu-queries 0 --host-steps 0"
timeout 600 python bench.py --config C2 --steps 10 --warmup 3 $B > gpurun_out/r2x_bench_c2.log 2>&1; tail -1 gpurun_out/r2x_bench_c2.log | cut -c1-1200
timeout 600 python bench.py --config C5 --nb 8000000 --nlist 8192 --nprobe 64 --steps 10 --warmup 3 --cpu-queries 256 --host-steps 0 > gpurun_out/r2x_bench_c5_8m.log 2>&1; tail -1 gpurun_out/r2x_bench_c5_8m.log | cut -c1-1800
timeout 1200 python bench.py --config C5 --steps 5 --warmup 2 > gpurun_out/r2x_bench_c5.log 2>&1; tail -1 gpurun_out/r2x_bench_c5.log | cut -c1-2200
