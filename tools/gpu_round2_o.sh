#!/bin/bash
# MFMA prefilter (mfma_scan.hip): parity tests, full GPU suite, C2 bench, a C5-shaped 8M bench
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_mscan.py -x -q > gpurun_out/r2o_mscan_tests.log 2>&1; tail -25 gpurun_out/r2o_mscan_tests.log | cut -c1-400
timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_gpu_mscan.py > gpurun_out/r2o_pytest.log 2>&1; tail -12 gpurun_out/r2o_pytest.log | cut -c1-300
timeout 600 python bench.py --config C2 --steps 10 --warmup 3 > gpurun_out/r2o_bench_c2.log 2>&1; tail -3 gpurun_out/r2o_bench_c2.log | cut -c1-3000
timeout 600 python bench.py --config C5 --nb 8000000 --nlist 8192 --nprobe 64 --steps 5 --warmup 2 --cpu-queries 256 > gpurun_out/r2o_bench_c5_8m.log 2>&1; tail -4 gpurun_out/r2o_bench_c5_8m.log | cut -c1-3000
KNHIP_MSCAN=0 timeout 600 python bench.py --config C5 --nb 8000000 --nlist 8192 --nprobe 64 --steps 3 --warmup 1 --cpu-queries 0 --host-steps 0 > gpurun_out/r2o_bench_c5_8m_exact.log 2>&1; tail -1 gpurun_out/r2o_bench_c5_8m_exact.log | cut -c1-1200
