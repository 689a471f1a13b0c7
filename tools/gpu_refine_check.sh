#!/bin/bash
# the refine stage: parity tests that go through refine.hip, then the C3 step with its stage table
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG="${1:-r05_refine}"
timeout 1200 python -m pytest tests/test_gpu_ties.py tests/test_gpu_parity.py tests/test_gpu_shards.py tests/test_gpu_refine_rows.py tests/test_gpu_node_devices.py tests/test_host_node.py -q -m gpu -x > gpurun_out/${TAG}_tests.log 2>&1
rc=$?; tail -4 gpurun_out/${TAG}_tests.log | cut -c1-400
if [ $rc -eq 0 ]; then
  timeout 600 python bench.py --steps 20 --warmup 5 --extra none --cpu-queries 128 > gpurun_out/${TAG}_bench.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/${TAG}_bench.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["value"], d["ms_per_step"], json.dumps(d.get("stages_ms_per_step", d["roofline"].get("stage_ms_per_step")))[:420])
        print(d["host_boundary"]["value"], d["cpu_baseline"].get("gpu_final_ids_equal"), d["cpu_baseline"].get("gpu_final_distances_bit_equal"))
PY
fi
