#!/bin/bash
# round-6 evidence at the final code commit (run on the GPU box from the repo root): GPU suite + smoke, the driver's own bench
# command, the rocprofv3 passes of the C3 step (statistics, PMC, one step as a timeline), then statistics + HBM traffic of C5
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_suite.sh r06
bash tools/gpu_driver_bench.sh r06_bench_driver
PROFILE_TAG=r06_c3_pqd BENCH_ARGS="--steps 5 --warmup 2 --cpu-queries 0 --host-steps 0 --extra none --gt-queries 10" bash tools/profile_bench.sh 2>&1 | grep "^wrote" | cut -c1-120
bash tools/gpu_step_timeline.sh r06_c3 | tail -3
R=$(pwd)
cd /tmp
ARGS="--config C5 --steps 3 --warmup 1 --cpu-queries 0 --host-steps 0 --extra none --gt-queries 10"
run() { name=$1; shift; rm -rf /tmp/pb_$name; (timeout 900 rocprofv3 "$@" --output-format csv -d /tmp/pb_$name -- python $R/bench.py $ARGS) > /tmp/pb_$name.log 2>&1; python $R/tools/pmc_summary.py /tmp/pb_$name $R/gpurun_out/r06_c5_$name.json; }
run stats --kernel-trace --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
