#!/bin/bash
# bench.py --gpus N over gloo on ONE GPU (KNHIP_ALLOW_SHARED_GPU=1): every N searches the same index, so the fingerprint of
# batch 0's results (result_crc32) must be the same at N = 1, 2, 3 -- the exchange protocol with this round's kernels
mkdir -p gpurun_out; export TMPDIR=/tmp
ARGS="--nb 8000000 --nlist 4096 --nprobe 64 --steps 3 --warmup 1 --cpu-queries 0 --host-steps 0 --extra none"
for n in 1 2 3; do
  KNHIP_ALLOW_SHARED_GPU=1 timeout 900 python bench.py --gpus $n --backend gloo $ARGS > gpurun_out/r06_result_crc_n$n.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/r06_result_crc_n$n.log"):
    if l.startswith("{"):
        d = json.loads(l); print("N=$n", d["n_gpus"], d["result_crc32"], d["value"], d["ms_per_step"], d.get("recall_at_10"), d["roofline"]["kernel"])
PY
done
