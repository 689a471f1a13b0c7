#!/bin/bash
# tools/prof_m24.sh -- two rocprofv3 --pmc passes (issue counters, LDS counters) of the generic-width IVF-PQ kernel
# (pq_scan_any.hip) on a 10M x 96, m = 24 index; run on the GPU box from the repo root.  Round 4 used it to find that the
# first version of the kernel spent its time in sequential top-k insertion (27 k vector instructions per wave at k = 101).
set -u
R=$(pwd)
ARGS="--config C3 --nb 10000000 --d 96 --m 24 --nlist 4096 --nprobe 64 --extra none --cpu-queries 0 --host-steps 0 --gt-queries 10 --steps 2 --warmup 1"
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rm -rf /tmp/pb_$name; (timeout 120 rocprofv3 "$@" --output-format csv -d /tmp/pb_$name -- python $R/bench.py $ARGS) > /tmp/pb_$name.log 2>&1; python $R/tools/pmc_summary.py /tmp/pb_$name $R/gpurun_out/m24_$name.json; }
run sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
