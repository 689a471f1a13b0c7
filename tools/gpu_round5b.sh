#!/bin/bash
# round 5, second block: the 512-thread IVF-PQ sample pass and the bf16 IVF-Flat filter -- parity tests, then C3 and C2
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG="${1:-r05b2}"
timeout 1200 python -m pytest tests/test_gpu_mscan.py tests/test_gpu_pqf.py tests/test_gpu_scale_parity.py tests/test_gpu_limits.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/${TAG}_tests.log 2>&1
rc=$?; tail -6 gpurun_out/${TAG}_tests.log | cut -c1-400
if [ $rc -eq 0 ]; then
  timeout 900 python bench.py --steps 20 --warmup 5 --extra C2 --cpu-queries 128 --host-steps 0 > gpurun_out/${TAG}_bench.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/${TAG}_bench.log"):
    if l.startswith("{"):
        d = json.loads(l)
        def show(n, x):
            r = x["roofline"]
            print(n, x["value"], x["ms_per_step"], json.dumps(x.get("stages_ms_per_step") or r.get("stage_ms_per_step")))
            print("   ", r.get("kernel"), r.get("frac"), r.get("ms_per_launch"), json.dumps(r.get("mscan")), x["cpu_baseline"].get("gpu_final_ids_equal"), x["cpu_baseline"].get("gpu_final_distances_bit_equal"))
        show("C3", d)
        for n, x in d.get("extra_configs", {}).items():
            show(n, x)
PY
fi
