#!/bin/bash
# per-kernel times of the contract bench's search steps (rocprofv3 --kernel-trace --stats only), search kernels listed
set -u
R=$(pwd)
TAG="${PROFILE_TAG:-r03_c3_stats}"
ARGS="${BENCH_ARGS:---steps 5 --warmup 1 --cpu-queries 0 --host-steps 0}"
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb_stats
(timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_stats -- python $R/bench.py $ARGS) > /tmp/pb_stats.log 2>&1
python $R/tools/pmc_summary.py /tmp/pb_stats $R/gpurun_out/${TAG}.json
grep -v "output_stream.cpp\|simple_timer" /tmp/pb_stats.log | tail -12 | cut -c1-300
python - <<PY
import json
d=json.load(open("$R/gpurun_out/${TAG}.json"))["__kernel_stats__"]
steps=6
rows=[(r["Name"].split("(")[0][-60:], int(r["Calls"]), float(r["TotalDurationNs"])) for r in d]
tot=0
for n,c,t in sorted(rows,key=lambda r:-r[2]):
    if c % steps == 0 and c//steps <= 4:
        print(f"{t/steps/1e3:9.1f} us/step  x{c//steps}  {n}"); tot+=t/steps/1e3
print(f"{tot:9.1f} us/step total of the listed kernels")
PY
