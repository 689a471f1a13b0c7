#!/bin/bash
R=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "q4 or headline or device_boundary or golden" > gpurun_out/r2d_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2d_pytest.log)
QB="python $R/tools/quick_bench.py --kind pq --nb 100000000 --nlist 16384 --nprobe 128 --nq 10000 --iters 3"
KNHIP_Q4=1 KNHIP_LIB=$R/knowhere_amd/libknhip_prof.so timeout 300 $QB --k 10 --iters 1 > $R/gpurun_out/r2d_timers.log 2>&1
for k in 10 100; do KNHIP_Q4=1 timeout 300 $QB --k $k >> $R/gpurun_out/r2d_qb.log 2>&1; done
tail -3 $R/gpurun_out/r2d_pytest.log; grep -v amdgpu.ids $R/gpurun_out/r2d_timers.log | tail -19; grep -v amdgpu.ids $R/gpurun_out/r2d_qb.log
