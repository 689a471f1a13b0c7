#!/bin/bash
# one C3 step as a timeline of dispatches (rocprofv3 --kernel-trace): gpurun_out/<tag>_timeline.txt
mkdir -p gpurun_out; R=$(pwd); TAG="${1:-r06_c3}"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl_$TAG
(timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$TAG -- python $R/bench.py --steps 6 --warmup 2 --cpu-queries 0 --host-steps 0 --extra none --gt-queries 10) > /tmp/tl_$TAG.log 2>&1
python $R/tools/kernel_gaps.py /tmp/tl_$TAG "pqd_kernel" $R/gpurun_out/${TAG}_timeline.txt; cat $R/gpurun_out/${TAG}_timeline.txt | cut -c1-150
