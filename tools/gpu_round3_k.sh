#!/bin/bash
# coarse stage with the two-pass row selection: parity tests, the stage alone at the C3 / C5 shapes, the C3 bench
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_limits.py tests/test_gpu_mscan.py -q -m gpu -k "coarse or limits or nprobe or mscan" > gpurun_out/r3k_tests.log 2>&1
tail -3 gpurun_out/r3k_tests.log | cut -c1-300
python tools/diag/coarse_only.py C3 2>&1 | tail -2
python tools/diag/coarse_only.py C5 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r3k_bench_c3.log 2>&1; tail -1 gpurun_out/r3k_bench_c3.log | cut -c1-700
