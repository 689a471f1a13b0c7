#!/bin/bash
# last validation of HEAD: what the driver runs at round end (GPU suite, smoke, default bench)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2y_pytest.log 2>&1; tail -3 gpurun_out/r2y_pytest.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2y_smoke.log 2>&1; tail -3 gpurun_out/r2y_smoke.log | cut -c1-300
timeout 900 python bench.py > gpurun_out/r2y_bench_c3.log 2>&1; tail -1 gpurun_out/r2y_bench_c3.log | cut -c1-1400
