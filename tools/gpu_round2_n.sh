#!/bin/bash
# re-entry baseline: full GPU suite (with durations), q4 phase timers on random codes, C2 + C5 contract benches
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu --durations=12 > gpurun_out/r2n_pytest.log 2>&1; tail -22 gpurun_out/r2n_pytest.log
QB="python $R/tools/quick_bench.py --kind pq --nb 100000000 --nlist 16384 --nprobe 128 --nq 10000"
KNHIP_LIB=$R/knowhere_amd/libknhip_prof.so timeout 300 $QB --k 100 --iters 1 > gpurun_out/r2n_timers.log 2>&1
grep -v amdgpu.ids gpurun_out/r2n_timers.log | tail -22 | cut -c1-200
timeout 600 python bench.py --config C2 --steps 10 --warmup 3 > gpurun_out/r2n_bench_c2.log 2>&1; tail -1 gpurun_out/r2n_bench_c2.log | cut -c1-2500
timeout 1500 python bench.py --config C5 --steps 5 --warmup 2 --verbose > gpurun_out/r2n_bench_c5.log 2>&1; tail -4 gpurun_out/r2n_bench_c5.log | cut -c1-2500
