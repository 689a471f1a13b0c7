#!/bin/bash
# GPU call of round 3: the code written blind at the end of round 2.
#  1. the gated parity tests of the half-precision ADC prefilter (pq_filter.hip) and of the larger row selection,
#     each under its own timeout (a hang must not eat the call);
#  2. only if (1) is green: the C3 bench with the prefilter off / on, then on the phase-timer build.
# gpurun --timeout 1500 -- 'bash tools/gpu_round3_a.sh'
mkdir -p gpurun_out
export TMPDIR=/tmp
KNHIP_TEST_PQF=1 timeout 420 python -m pytest tests/test_gpu_pqf.py -q -m gpu > gpurun_out/r3a_pqf.log 2>&1
rc=$?; tail -5 gpurun_out/r3a_pqf.log | cut -c1-400
KNHIP_TEST_UNVALIDATED=1 timeout 420 python -m pytest tests/test_gpu_limits.py -x -q -m gpu > gpurun_out/r3a_limits.log 2>&1
tail -3 gpurun_out/r3a_limits.log | cut -c1-400
if [ $rc -eq 0 ]; then
  timeout 600 python bench.py > gpurun_out/r3a_bench_c3_q4.log 2>&1; tail -1 gpurun_out/r3a_bench_c3_q4.log | cut -c1-1400
  KNHIP_PQF=1 timeout 600 python bench.py > gpurun_out/r3a_bench_c3_pqf.log 2>&1; tail -1 gpurun_out/r3a_bench_c3_pqf.log | cut -c1-1400
  KNHIP_PQF=1 KNHIP_LIB=tools/prof/libknhip_prof.so timeout 600 python bench.py --steps 2 --warmup 1 > gpurun_out/r3a_bench_c3_pqf_prof.log 2>&1
  grep "pqf timers" gpurun_out/r3a_bench_c3_pqf_prof.log | tail -40 | cut -c1-200
fi
