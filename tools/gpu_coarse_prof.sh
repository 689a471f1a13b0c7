#!/bin/bash
# per-kernel times of the coarse stage alone (C3 / C5 shapes, random centroids): rocprofv3 --kernel-trace --stats
mkdir -p gpurun_out
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for shape in ${SHAPES:-C3 C5}; do
  rm -rf /tmp/cp_$shape
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_$shape -- python $R/tools/diag/coarse_only.py $shape > /tmp/cp_$shape.log 2>&1
  python $R/tools/pmc_summary.py /tmp/cp_$shape $R/gpurun_out/${TAG:-r05}_coarse_${shape}_stats.json > /dev/null 2>&1
  grep "coarse ms" /tmp/cp_$shape.log | tail -1
  python - <<PY
import json
d = json.load(open("$R/gpurun_out/${TAG:-r05}_coarse_${shape}_stats.json"))
for r in d.get("__kernel_stats__", [])[:12]:
    print(f"{r['Name'][:86]:86s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs']) / 1e3:9.1f} pct {r['Percentage']}")
PY
done
