#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r2m_pytest.log 2>&1; tail -8 gpurun_out/r2m_pytest.log
timeout 300 python tools/bench_prims.py > gpurun_out/r2m_prims.log 2>&1; tail -30 gpurun_out/r2m_prims.log | cut -c1-200
timeout 1500 python bench.py --config C5 --steps 5 --warmup 2 --verbose > gpurun_out/r2m_bench_c5.log 2>&1; tail -2 gpurun_out/r2m_bench_c5.log | cut -c1-1800
