#!/bin/bash
# filter time of the C3 step under each compile-time knock-out of pq_decode.hip (tools/build_pqd_variants.sh)
mkdir -p gpurun_out; export TMPDIR=/tmp
for x in "$@"; do
  L=tools/prof/libknhip_pqd_x$x.so
  [ "$x" = "0" ] && L=knowhere_amd/libknhip.so
  KNHIP_LIB=$L timeout 600 python bench.py --steps 5 --warmup 2 --cpu-queries 0 --host-steps 0 --extra none --gt-queries 10 > gpurun_out/r06_pqd_x$x.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/r06_pqd_x$x.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print("PD_EXP=$x step %.3f filter %.3f cand/query %s recall %s" % (d["ms_per_step"], r["stage_ms_per_step"]["filter"], r.get("mscan", {}).get("candidates_per_query"), d.get("recall_at_10")))
PY
done
