#!/bin/bash
# IVF-Flat / IVF-SQ8 prefilter paths: parity tests, then C2 and C5s with their stage tables
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG="${1:-r05_rowkinds}"
timeout 900 python -m pytest tests/test_gpu_mscan.py tests/test_gpu_limits.py tests/test_gpu_parity.py tests/test_gpu_cosine.py tests/test_gpu_shards.py -q -m gpu -x > gpurun_out/${TAG}_tests.log 2>&1
rc=$?; tail -4 gpurun_out/${TAG}_tests.log | cut -c1-400
show() {
python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        r = d["roofline"]
        print(d["config"]["name"], d["value"], d["ms_per_step"], d.get("recall_at_10"), json.dumps(d.get("stages_ms_per_step") or r.get("stage_ms_per_step")))
        print("   ", r.get("kernel"), r.get("bound"), r.get("frac"), d["cpu_baseline"].get("gpu_final_ids_equal"), d["cpu_baseline"].get("gpu_final_distances_bit_equal"))
PY
}
if [ $rc -eq 0 ]; then
  timeout 600 python bench.py --config C2 --steps 20 --warmup 5 --extra none --cpu-queries 128 --host-steps 0 > gpurun_out/${TAG}_c2.log 2>&1; show gpurun_out/${TAG}_c2.log
  timeout 600 python bench.py --config C5s --steps 10 --warmup 3 --extra none --cpu-queries 64 --host-steps 0 > gpurun_out/${TAG}_c5s.log 2>&1; show gpurun_out/${TAG}_c5s.log
fi
