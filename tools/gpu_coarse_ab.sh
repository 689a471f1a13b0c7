#!/bin/bash
# coarse stage / BRUTE_FORCE timings (C3 with the extra points C1, C1m): stage tables of each line
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python bench.py --steps 5 --warmup 2 --cpu-queries 0 --host-steps 0 --extra C1,C1m,C2 --gt-queries 10 > gpurun_out/r06_coarse_ab.log 2>&1
python - <<PY
import json
for l in open("gpurun_out/r06_coarse_ab.log"):
    if l.startswith("{"):
        d = json.loads(l); st = d["roofline"].get("stage_ms_per_step", {})
        print(d["config"]["name"], d["ms_per_step"], {k: st.get(k) for k in ("coarse", "scan", "filter", "sample")})
PY
