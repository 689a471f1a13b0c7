#!/bin/bash
# C3 on data that is not 524k clean blobs (uniform, low intrinsic dimension), then C2 and C5 re-timed
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --data uniform > gpurun_out/r3j_bench_c3_uniform.log 2>&1; tail -1 gpurun_out/r3j_bench_c3_uniform.log | cut -c1-1500
timeout 900 python bench.py --latent 16 > gpurun_out/r3j_bench_c3_latent16.log 2>&1; tail -1 gpurun_out/r3j_bench_c3_latent16.log | cut -c1-1500
timeout 900 python bench.py --config C2 > gpurun_out/r3j_bench_c2.log 2>&1; tail -1 gpurun_out/r3j_bench_c2.log | cut -c1-1500
timeout 1500 python bench.py --config C5 > gpurun_out/r3j_bench_c5.log 2>&1; tail -1 gpurun_out/r3j_bench_c5.log | cut -c1-1800
