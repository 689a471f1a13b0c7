#!/bin/bash
# rocprofv3 kernel stats of the C3 bench (matrix-core prefilter default)
R=$(pwd); mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb_stats
(timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_stats -- python $R/bench.py --steps 5 --warmup 2 --cpu-queries 0 --host-steps 0) > /tmp/pb_stats.log 2>&1
python $R/tools/pmc_summary.py /tmp/pb_stats $R/gpurun_out/r03_c3_pqf_stats.json
tail -1 /tmp/pb_stats.log | cut -c1-600
