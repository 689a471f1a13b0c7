#!/bin/bash
# PQ prefilter parity tests + C3 bench (default path) + phase timers
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_gpu_pqf.py tests/test_gpu_limits.py -q -m gpu > gpurun_out/r3f_pqf.log 2>&1
rc=$?; tail -3 gpurun_out/r3f_pqf.log | cut -c1-400
if [ $rc -eq 0 ]; then
  timeout 600 python bench.py > gpurun_out/r3f_bench_c3.log 2>&1; tail -1 gpurun_out/r3f_bench_c3.log | cut -c1-2600
  if [ -n "$PROF" ]; then
    KNHIP_LIB=tools/prof/libknhip_prof.so timeout 600 python bench.py --steps 2 --warmup 1 --cpu-queries 0 --host-steps 0 > gpurun_out/r3f_bench_c3_prof.log 2>&1
    grep "pqf timers" gpurun_out/r3f_bench_c3_prof.log | awk '!seen[$0]++' | sed -n 19,36p | cut -c1-200
  fi
fi
