#!/bin/bash
# retry round of overflowed queries: full GPU suite (incl. the forced-small-capacity tests), the C5 shape at 8M with a
# capacity of 64 (most queries retried; result checked bitwise against the CPU reference), C2 unchanged
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -x -q -m gpu > gpurun_out/r2z_pytest.log 2>&1; tail -3 gpurun_out/r2z_pytest.log | cut -c1-300
KNHIP_MSCAN_CAP=64 timeout 300 python bench.py --config C5 --nb 8000000 --nlist 8192 --nprobe 64 --steps 3 --warmup 1 --cpu-queries 256 --host-steps 0 > gpurun_out/r2z_bench_c5_8m_cap64.log 2>&1; tail -1 gpurun_out/r2z_bench_c5_8m_cap64.log | cut -c1-2600
timeout 300 python bench.py --config C2 --steps 10 --warmup 3 --cpu-queries 0 --host-steps 0 > gpurun_out/r2z_bench_c2.log 2>&1; tail -1 gpurun_out/r2z_bench_c2.log | cut -c1-900
