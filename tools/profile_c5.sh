#!/bin/bash
# PMC evidence for the SQ8 prefilter: all passes on C5s (10M x 768: the default line's configuration), and the kernel-stats /
# FETCH_SIZE / matrix-pipe passes on the full C5 (100M x 768: 90 s of build per pass)
R=$(pwd)
PROFILE_TAG=r05_c5s BENCH_ARGS="--config C5s --steps 3 --warmup 1 --cpu-queries 0 --host-steps 0 --extra none" bash tools/profile_bench.sh 2>&1 | grep "^wrote" | tail -7
cd /tmp && export TMPDIR=/tmp
ARGS="--config C5 --steps 3 --warmup 1 --cpu-queries 0 --host-steps 0 --extra none"
run() { name=$1; shift; rm -rf /tmp/pb_$name; (timeout 900 rocprofv3 "$@" --output-format csv -d /tmp/pb_$name -- python $R/bench.py $ARGS) > /tmp/pb_$name.log 2>&1; python $R/tools/pmc_summary.py /tmp/pb_$name $R/gpurun_out/r05_c5_$name.json; }
run stats --kernel-trace --stats
run fetch --pmc FETCH_SIZE
run mfma --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
