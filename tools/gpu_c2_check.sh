#!/bin/bash
# IVF-Flat / IVF-SQ8 prefilter: parity tests, then the C2 step with its stage table (arguments: tag, extra bench flags)
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG="${1:-r05_c2}"; shift
timeout 900 python -m pytest tests/test_gpu_mscan.py tests/test_gpu_limits.py -q -m gpu -x > gpurun_out/${TAG}_tests.log 2>&1
rc=$?; tail -4 gpurun_out/${TAG}_tests.log | cut -c1-400
if [ $rc -eq 0 ]; then
  timeout 600 python bench.py --config C2 --steps 20 --warmup 5 --extra none --cpu-queries 128 --host-steps 0 "$@" > gpurun_out/${TAG}_bench.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/${TAG}_bench.log"):
    if l.startswith("{"):
        d = json.loads(l)
        r = d["roofline"]
        print(d["value"], d["ms_per_step"], d.get("recall_at_10"), json.dumps(d.get("stages_ms_per_step") or r.get("stage_ms_per_step")))
        print("   ", r.get("kernel"), r.get("frac"), r.get("ms_per_launch"), json.dumps(r.get("mscan")), d["cpu_baseline"].get("gpu_final_ids_equal"), d["cpu_baseline"].get("gpu_final_distances_bit_equal"))
PY
fi
