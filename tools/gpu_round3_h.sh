#!/bin/bash
# LDS counters of the C3 bench (bank conflicts of the prefilter kernels)
R=$(pwd); mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb_lds
(timeout 900 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d /tmp/pb_lds -- python $R/bench.py --steps 3 --warmup 1 --cpu-queries 0 --host-steps 0) > /tmp/pb_lds.log 2>&1
python $R/tools/pmc_summary.py /tmp/pb_lds $R/gpurun_out/r03_c3_lds.json
tail -1 /tmp/pb_lds.log | cut -c1-200
