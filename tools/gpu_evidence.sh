#!/bin/bash
# evidence refresh (run on the GPU box from the repo root): full GPU suite, smoke, the contract bench (with
# the CPU baseline leg), then the rocprofv3 passes of the same command (stats + PMC, tools/profile_bench.sh)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r03_pytest_gpu.log 2>&1; tail -3 gpurun_out/r03_pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke.log 2>&1; tail -2 gpurun_out/r03_smoke.log | cut -c1-300
timeout 900 python bench.py > gpurun_out/r03_bench_c3_final.log 2>&1; tail -1 gpurun_out/r03_bench_c3_final.log | cut -c1-400
PROFILE_TAG=r03_c3_pqi bash tools/profile_bench.sh 2>&1 | tail -12 | cut -c1-200
