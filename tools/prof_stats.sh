#!/bin/bash
# tools/prof_stats.sh -- rocprofv3 --kernel-trace --stats of one bench.py run -> gpurun_out/${TAG}_stats.json + a per-kernel table
# (run on the GPU box from the repo root).  BENCH_ARGS / PROFILE_TAG as tools/profile_bench.sh.
set -u
R=$(pwd)
TAG="${PROFILE_TAG:-bench}"
ARGS="${BENCH_ARGS:---steps 5 --warmup 2 --cpu-queries 0 --host-steps 0 --extra none --gt-queries 10}"
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps_$TAG
(timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$TAG -- python $R/bench.py $ARGS) > /tmp/ps_$TAG.log 2>&1
python $R/tools/pmc_summary.py /tmp/ps_$TAG $R/gpurun_out/${TAG}_stats.json
python - <<PY
import json
d=json.load(open("$R/gpurun_out/${TAG}_stats.json"))
rows=d.get("__kernel_stats__",[])
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:28]:
    print("%-70s calls %5s avg %10.1f us  tot %9.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
