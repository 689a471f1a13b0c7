#!/bin/bash
# integer form of the PQ prefilter: parity tests (both forms), C3 bench (form chosen by the guard), phase timers
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pqf.py -q -m gpu > gpurun_out/r3g_pqf.log 2>&1
rc=$?; tail -4 gpurun_out/r3g_pqf.log | cut -c1-600
if [ $rc -eq 0 ]; then
  timeout 600 python bench.py > gpurun_out/r3g_bench_c3.log 2>&1; tail -1 gpurun_out/r3g_bench_c3.log | cut -c1-2600
  KNHIP_LIB=tools/prof/libknhip_prof.so timeout 600 python bench.py --steps 2 --warmup 1 --cpu-queries 0 --host-steps 0 > gpurun_out/r3g_bench_c3_prof.log 2>&1
  grep "pqf timers" gpurun_out/r3g_bench_c3_prof.log | awk '!seen[$0]++' | grep -A16 "int8" | head -17 | cut -c1-200
fi
