#!/bin/bash
# MFMA prefilter v6: rigorous but 20x tighter SQ8 error bound, batched prologue loads (flat)
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_mscan.py -x -q > gpurun_out/r2t_mscan_tests.log 2>&1; tail -5 gpurun_out/r2t_mscan_tests.log | cut -c1-400
timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_gpu_mscan.py > gpurun_out/r2t_pytest.log 2>&1; tail -5 gpurun_out/r2t_pytest.log | cut -c1-300
B="--steps 10 --warmup 3 --cpu-queries 0 --host-steps 0"
timeout 600 python bench.py --config C2 $B > gpurun_out/r2t_bench_c2.log 2>&1; tail -1 gpurun_out/r2t_bench_c2.log | cut -c1-2000
timeout 600 python bench.py --config C5 --nb 8000000 --nlist 8192 --nprobe 64 --steps 10 --warmup 3 --cpu-queries 256 > gpurun_out/r2t_bench_c5_8m.log 2>&1; tail -1 gpurun_out/r2t_bench_c5_8m.log | cut -c1-2600
timeout 1500 python bench.py --config C5 --steps 5 --warmup 2 --verbose > gpurun_out/r2t_bench_c5.log 2>&1; tail -3 gpurun_out/r2t_bench_c5.log | cut -c1-3000
