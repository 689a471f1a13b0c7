#!/bin/bash
R=$(pwd); mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
for S in C3 C5; do
rm -rf /tmp/pc_$S
(timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$S -- python $R/tools/diag/coarse_only.py $S) > /tmp/pc_$S.log 2>&1
grep "coarse ms" /tmp/pc_$S.log | tail -3
python $R/tools/pmc_summary.py /tmp/pc_$S $R/gpurun_out/r03_coarse_${S}_stats.json
done
