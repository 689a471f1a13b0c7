set -u
R=$(pwd)
TAG=r06_c3_pqd4
ARGS="--steps 3 --warmup 1 --cpu-queries 0 --host-steps 0 --extra none --gt-queries 10"
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rm -rf /tmp/pb_$name; (timeout 900 rocprofv3 "$@" --output-format csv -d /tmp/pb_$name -- python $R/bench.py $ARGS) > /tmp/pb_$name.log 2>&1; python $R/tools/pmc_summary.py /tmp/pb_$name $R/gpurun_out/${TAG}_$name.json; }
run sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run wait --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_IFETCH SQ_WAVES
run mfma --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
run misc --pmc SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT
