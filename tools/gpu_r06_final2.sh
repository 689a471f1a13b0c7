#!/bin/bash
# round-6 closing evidence at the last code commit: GPU suite + smoke, the driver's own bench command
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_suite.sh r06
bash tools/gpu_driver_bench.sh r06_bench_driver
