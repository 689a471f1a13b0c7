#!/bin/bash
# MFMA prefilter v2 (sample -> tau, all probes filtered, compact exact fallback) + pipelined coarse GEMM:
# parity tests, full GPU suite, C2 / C3 / C5-shaped 8M benches, rocprof stats of C2, full C5
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_mscan.py -x -q > gpurun_out/r2p_mscan_tests.log 2>&1; tail -25 gpurun_out/r2p_mscan_tests.log | cut -c1-400
timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_gpu_mscan.py > gpurun_out/r2p_pytest.log 2>&1; tail -12 gpurun_out/r2p_pytest.log | cut -c1-300
timeout 600 python bench.py --config C2 --steps 10 --warmup 3 > gpurun_out/r2p_bench_c2.log 2>&1; tail -1 gpurun_out/r2p_bench_c2.log | cut -c1-3000
timeout 600 python bench.py --config C5 --nb 8000000 --nlist 8192 --nprobe 64 --steps 5 --warmup 2 --cpu-queries 256 > gpurun_out/r2p_bench_c5_8m.log 2>&1; tail -1 gpurun_out/r2p_bench_c5_8m.log | cut -c1-3000
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r2p_bench_c3.log 2>&1; tail -1 gpurun_out/r2p_bench_c3.log | cut -c1-2000
(cd /tmp && rm -rf /tmp/pb_c2 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_c2 -- python $R/bench.py --config C2 --steps 5 --warmup 2 --cpu-queries 0 --host-steps 0 > /tmp/pb_c2.log 2>&1; python $R/tools/pmc_summary.py /tmp/pb_c2 $R/gpurun_out/r2p_c2_rocprof_stats.json; tail -1 /tmp/pb_c2.log | cut -c1-300)
timeout 1500 python bench.py --config C5 --steps 5 --warmup 2 --verbose > gpurun_out/r2p_bench_c5.log 2>&1; tail -3 gpurun_out/r2p_bench_c5.log | cut -c1-3000
