#!/bin/bash
# quantised refine stores (fp16 / bf16 / sq8 / sq6 / int8): device encoders, refine, node round trips, sharded node
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG="${1:-r05_rows}"
timeout 1200 python -m pytest tests/test_gpu_refine_rows.py tests/test_faiss_io.py tests/test_gpu_node_devices.py tests/test_gpu_shards.py -q -m gpu -x > gpurun_out/${TAG}_tests.log 2>&1
echo "rc=$?"; tail -6 gpurun_out/${TAG}_tests.log | cut -c1-400
