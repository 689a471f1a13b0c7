#!/bin/bash
# tools/profile_bench.sh -- rocprofv3 evidence for the contract bench (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of `python bench.py` (the same command the driver times)
#   2. separate --pmc passes (never combined with a trace domain): FETCH_SIZE / WRITE_SIZE (TCC has 4 slots, FETCH costs
#      3 and WRITE 2) for the HBM traffic of the scan kernel, corrected as MI355X_MICROARCH.md prescribes for gfx950
#      (FETCH_SIZE is in KB and reports 1/2 of a wide coalesced read -> bytes = FETCH_SIZE * 1024 * 2), the SQ issue
#      counters, and the LDS counters (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE: the ADC scan is bound by the LDS gather)
# Writes small summaries to gpurun_out/ (raw rocprof output stays in /tmp on the box); tools/pmc_traffic.py turns the
# fetch / write summaries into profiles/bench_pmc_traffic.json, stamped with the commit they were taken at.
set -u
R=$(pwd)
TAG="${PROFILE_TAG:-bench}"
ARGS="${BENCH_ARGS:---steps 3 --warmup 1 --cpu-queries 0 --host-steps 0 --extra none}"
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rm -rf /tmp/pb_$name; (timeout 900 rocprofv3 "$@" --output-format csv -d /tmp/pb_$name -- python $R/bench.py $ARGS) > /tmp/pb_$name.log 2>&1; python $R/tools/pmc_summary.py /tmp/pb_$name $R/gpurun_out/${TAG}_$name.json; tail -1 /tmp/pb_$name.log | cut -c1-300; }
run stats --kernel-trace --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA
run mfma --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
run wait --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_IFETCH SQ_WAVES
