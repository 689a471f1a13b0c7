#!/bin/bash
R=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2c_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c_pytest.log)
QB="python $R/tools/quick_bench.py --kind pq --nb 100000000 --nlist 16384 --nprobe 128 --nq 10000 --iters 3"
KNHIP_Q4=1 KNHIP_LIB=$R/knowhere_amd/libknhip_prof.so timeout 300 $QB --k 10 --iters 1 > $R/gpurun_out/r2c_timers.log 2>&1
for q4 in 0 1; do for k in 10 100; do KNHIP_Q4=$q4 timeout 300 $QB --k $k >> $R/gpurun_out/r2c_qb.log 2>&1; done; done
tail -3 $R/gpurun_out/r2c_pytest.log; grep -v amdgpu.ids $R/gpurun_out/r2c_timers.log | tail -20; grep -v amdgpu.ids $R/gpurun_out/r2c_qb.log
