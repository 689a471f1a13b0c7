#!/bin/bash
# round 5: the coarse prefilter on the bf16 matrix pipe: parity tests, then the stage alone at the C3 / C5 shapes (both forms)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ties.py tests/test_gpu_build.py tests/test_gpu_shards.py -q -m gpu > gpurun_out/r05_coarse_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r05_coarse_tests.log | cut -c1-400
for shape in C3 C5; do
  for mode in fp32 bf16; do
    echo "== $shape $mode"; KNHIP_COARSE=$mode timeout 300 python tools/diag/coarse_only.py $shape 2>&1 | tail -3
  done
done
