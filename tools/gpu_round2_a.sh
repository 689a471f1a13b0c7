#!/bin/bash
# first GPU check of the round: q4 parity tests, full GPU suite, random-code quick bench A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "q4 or headline or device_boundary" > gpurun_out/r2a_q4tests.log 2>&1
echo "q4 tests rc=$?" >> gpurun_out/r2a_q4tests.log
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/r2a_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2a_pytest.log
for q4 in 0 1; do
  for k in 10 100; do
    KNHIP_Q4=$q4 timeout 600 python tools/quick_bench.py --kind pq --nb 100000000 --nlist 16384 --nprobe 128 --nq 10000 --k $k --iters 3 >> gpurun_out/r2a_qb.log 2>&1
  done
done
tail -5 gpurun_out/r2a_q4tests.log; tail -3 gpurun_out/r2a_pytest.log; cat gpurun_out/r2a_qb.log
