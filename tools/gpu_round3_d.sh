#!/bin/bash
# Matrix-core ADC prefilter (pq_filter.hip, round 3): gated parity tests, GPU build tests (train parity), then the C3
# bench with the prefilter on (guard on / off) and the phase timers.
mkdir -p gpurun_out
export TMPDIR=/tmp
KNHIP_TEST_PQF=1 timeout 420 python -m pytest tests/test_gpu_pqf.py -q -m gpu > gpurun_out/r3d_pqf.log 2>&1
rc=$?; tail -5 gpurun_out/r3d_pqf.log | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_build.py -q -m gpu > gpurun_out/r3d_build.log 2>&1
tail -5 gpurun_out/r3d_build.log | cut -c1-400
if [ $rc -eq 0 ]; then
  KNHIP_PQF=1 timeout 600 python bench.py > gpurun_out/r3d_bench_c3_pqf.log 2>&1; tail -1 gpurun_out/r3d_bench_c3_pqf.log | cut -c1-2400
  KNHIP_PQF=1 KNHIP_PQF_GUARD=0 KNHIP_LIB=tools/prof/libknhip_prof.so timeout 600 python bench.py --steps 2 --warmup 1 > gpurun_out/r3d_bench_c3_pqf_prof.log 2>&1
  grep "pqf timers" gpurun_out/r3d_bench_c3_pqf_prof.log | awk '!seen[$0]++' | head -36 | cut -c1-200
fi
