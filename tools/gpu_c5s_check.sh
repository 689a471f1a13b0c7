#!/bin/bash
# IVF-SQ8 prefilter: parity tests, then the C5s step (10M x 768) with its stage table
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG="${1:-r05_c5s}"
timeout 900 python -m pytest tests/test_gpu_mscan.py tests/test_gpu_limits.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/${TAG}_tests.log 2>&1
rc=$?; tail -4 gpurun_out/${TAG}_tests.log | cut -c1-400
if [ $rc -eq 0 ]; then
  timeout 600 python bench.py --config C5s --steps 10 --warmup 3 --extra none --cpu-queries 64 --host-steps 0 > gpurun_out/${TAG}_bench.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/${TAG}_bench.log"):
    if l.startswith("{"):
        d = json.loads(l)
        r = d["roofline"]
        print(d["value"], d["ms_per_step"], d.get("recall_at_10"), json.dumps(d.get("stages_ms_per_step") or r.get("stage_ms_per_step")))
        print("   ", r.get("kernel"), r.get("frac"), json.dumps(r.get("mscan")), json.dumps(r.get("stream")), d["cpu_baseline"].get("gpu_final_ids_equal"), d["cpu_baseline"].get("gpu_final_distances_bit_equal"))
PY
fi
