#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r2k_pytest.log 2>&1; tail -3 gpurun_out/r2k_pytest.log
(cd knowhere_amd/host && timeout 600 ./test_hip_index) > gpurun_out/r2k_cpp_node_test.log 2>&1; tail -4 gpurun_out/r2k_cpp_node_test.log
timeout 1500 python bench.py --config C5 --steps 5 --warmup 2 --verbose > gpurun_out/r2k_bench_c5.log 2>&1; tail -2 gpurun_out/r2k_bench_c5.log | cut -c1-1800
timeout 600 python bench.py --config C2 --steps 10 --warmup 3 > gpurun_out/r2k_bench_c2.log 2>&1; tail -1 gpurun_out/r2k_bench_c2.log | cut -c1-1800
