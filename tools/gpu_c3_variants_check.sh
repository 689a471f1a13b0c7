#!/bin/bash
# the 100M-row IVF-PQ step on the other two data generators WITH the reference leg (the default run skips it for them): the
# first 256 queries against the scalar reference build, ids and distance bits
mkdir -p gpurun_out; export TMPDIR=/tmp
for c in C3u C3l; do
  timeout 1200 python bench.py --config $c --steps 5 --warmup 2 --cpu-queries 256 --host-steps 0 --extra none > gpurun_out/r06_bench_${c}_checked.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/r06_bench_${c}_checked.log"):
    if l.startswith("{"):
        d = json.loads(l); cb = d.get("cpu_baseline", {})
        print("$c", d["value"], d["ms_per_step"], d.get("recall_at_10"), d["roofline"]["kernel"], "ids", cb.get("gpu_final_ids_equal"), "bits", cb.get("gpu_final_distances_bit_equal"), "cpu qps", cb.get("value"))
PY
done
