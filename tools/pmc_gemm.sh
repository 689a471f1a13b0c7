# PMC passes of the split-bf16 GEMM at the C5 coarse shape (tools/diag/coarse_only.py): what bounds coarse_bf16_kernel
set -u
R=$(pwd)
TAG=r06_gemm_c5
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rm -rf /tmp/pg_$name; (timeout 600 rocprofv3 "$@" --output-format csv -d /tmp/pg_$name -- python $R/tools/diag/coarse_only.py C5) > /tmp/pg_$name.log 2>&1; python $R/tools/pmc_summary.py /tmp/pg_$name $R/gpurun_out/${TAG}_$name.json; }
run sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
run mfma --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
run fetch --pmc FETCH_SIZE
cd $R
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_gemm_c5_*.json")):
    d = json.load(open(f))
    for k, v in d.items():
        if "coarse_bf16_kernel" in k:
            print(f.split("_")[-1], k[:60], json.dumps(v)[:600])
PY
