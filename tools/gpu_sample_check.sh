#!/bin/bash
# the IVF-PQ sample pass (pq_sample_kernel): parity tests that go through it, then the C3 step with its stage table
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG="${1:-r05_sample}"
timeout 900 python -m pytest tests/test_gpu_pqf.py tests/test_gpu_scale_parity.py tests/test_gpu_limits.py -q -m gpu -x > gpurun_out/${TAG}_tests.log 2>&1
rc=$?; tail -4 gpurun_out/${TAG}_tests.log | cut -c1-400
if [ $rc -eq 0 ]; then
  timeout 600 python bench.py --steps 20 --warmup 5 --extra none --cpu-queries 128 --host-steps 0 > gpurun_out/${TAG}_bench.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/${TAG}_bench.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["value"], d["ms_per_step"], json.dumps(d.get("stages_ms_per_step", d["roofline"].get("stage_ms_per_step"))))
        print(d["cpu_baseline"].get("gpu_final_ids_equal"), d["cpu_baseline"].get("gpu_final_distances_bit_equal"))
PY
fi
