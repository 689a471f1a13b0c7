#!/bin/bash
# round-6 evidence (run on the GPU box from the repo root): the driver's own bench command, then the rocprofv3 passes of the
# same workload (kernel statistics + separate PMC passes) for the decode-form prefilter
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_driver_bench.sh r06_bench_driver
PROFILE_TAG=r06_c3_pqd BENCH_ARGS="--steps 5 --warmup 2 --cpu-queries 0 --host-steps 0 --extra none --gt-queries 10" bash tools/profile_bench.sh 2>&1 | cut -c1-200
