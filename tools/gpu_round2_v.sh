#!/bin/bash
# after restoring the simple flat prologue: parity tests + C2; then the FETCH_SIZE pass of the full C5 bench
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/r2v_pytest.log 2>&1; tail -3 gpurun_out/r2v_pytest.log | cut -c1-300
timeout 600 python bench.py --config C2 --steps 10 --warmup 3 --cpu-queries 0 --host-steps 0 > gpurun_out/r2v_bench_c2.log 2>&1; tail -1 gpurun_out/r2v_bench_c2.log | cut -c1-1500
cd /tmp
rm -rf /tmp/pb_c5_fetch; (timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pb_c5_fetch -- python $R/bench.py --config C5 --steps 2 --warmup 1 --cpu-queries 0 --host-steps 0 --gt-queries 100) > /tmp/pb_c5_fetch.log 2>&1; python $R/tools/pmc_summary.py /tmp/pb_c5_fetch $R/gpurun_out/r2v_c5_rocprof_fetch.json | tail -1; tail -1 /tmp/pb_c5_fetch.log | cut -c1-600
