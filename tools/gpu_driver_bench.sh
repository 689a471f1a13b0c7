#!/bin/bash
# the driver's own command (N = 1), timed from outside; log + a one-line digest per configuration
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG="${1:-r05_bench_driver}"
t0=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}.log 2>&1
echo "rc=$? wall=$(( $(date +%s) - t0 ))s"
python - <<PY
import json
for l in open("gpurun_out/${TAG}.log"):
    if l.startswith("{"):
        d = json.loads(l)
        def show(n, x):
            r = x.get("roofline", {})
            cb = x.get("cpu_baseline", {})
            print(n, x.get("value"), x.get("ms_per_step"), x.get("recall_at_10"), r.get("kernel"), r.get("bound"), r.get("frac"),
                  cb.get("value"), cb.get("gpu_final_ids_equal"), cb.get("gpu_final_distances_bit_equal"))
            st = x.get("stages_ms_per_step") or r.get("stage_ms_per_step")
            print("    ", json.dumps(st)[:400])
        show("C3", d)
        print("   host", d.get("host_boundary", {}).get("value"))
        for n, x in d.get("extra_configs", {}).items():
            show(n, x)
PY
grep -i "error\|traceback" gpurun_out/${TAG}.log | head -5
