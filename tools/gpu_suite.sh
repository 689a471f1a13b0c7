#!/bin/bash
# the whole -m gpu suite + smoke (run on the GPU box from the repo root); log under gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG="${1:-r05}"
timeout 2400 python -m pytest tests -q -m gpu -x --durations=8 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/${TAG}_smoke.log | cut -c1-300
