#!/bin/bash
# C3 step against the size of the IVF-PQ sample pass (KNHIP_PQ_SAMPLE_ROWS), decode form
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
for n in 4096 2048 1024; do
  KNHIP_PQ_SAMPLE_ROWS=$n timeout 600 python bench.py --steps 8 --warmup 2 --cpu-queries 0 --host-steps 0 --extra none --gt-queries 10 > gpurun_out/r06_srows_$n.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/r06_srows_$n.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]; st = r["stage_ms_per_step"]
        print("rows=$n step %.3f sample %.3f filter %.3f finish %.3f cand/query %s recall %s" % (d["ms_per_step"], st["sample"], st["filter"], st["finish"], r.get("mscan", {}).get("candidates_per_query"), d.get("recall_at_10")))
PY
done; done
