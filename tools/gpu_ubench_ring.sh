#!/bin/bash
# round 5, first GPU call: the micro-benchmarks behind the pqi_kernel redesign + the query-group union statistic
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 tools/ubench/pqi_ring > gpurun_out/r05_ubench_pqi_ring.log 2>&1; echo "ring rc=$?"; cat gpurun_out/r05_ubench_pqi_ring.log | cut -c1-260
timeout 300 python tools/diag/query_group_union.py > gpurun_out/r05_query_group_union.log 2>&1; echo "union rc=$?"; tail -6 gpurun_out/r05_query_group_union.log | cut -c1-330
