#!/bin/bash
# C3 bench through the half-precision prefilter with the selectivity guard off: what the filter really passes on the
# bench data, and the phase timers of the filter kernel.
mkdir -p gpurun_out
export TMPDIR=/tmp
KNHIP_PQF=1 KNHIP_PQF_GUARD=0 timeout 600 python bench.py > gpurun_out/r3c_bench_c3_pqf_noguard.log 2>&1; tail -1 gpurun_out/r3c_bench_c3_pqf_noguard.log | cut -c1-2400
KNHIP_PQF=1 KNHIP_PQF_GUARD=0 KNHIP_LIB=tools/prof/libknhip_prof.so timeout 600 python bench.py --steps 2 --warmup 1 > gpurun_out/r3c_bench_c3_pqf_noguard_prof.log 2>&1
grep "pqf timers" gpurun_out/r3c_bench_c3_pqf_noguard_prof.log | tail -18 | cut -c1-200
