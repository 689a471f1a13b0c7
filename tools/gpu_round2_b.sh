#!/bin/bash
# phase timers + PMC counters of the q4 scan on the random-code 100M quick bench
R=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
QB="python $R/tools/quick_bench.py --kind pq --nb 100000000 --nlist 16384 --nprobe 128 --nq 10000 --k 10 --iters 2"
KNHIP_Q4=1 KNHIP_LIB=$R/knowhere_amd/libknhip_prof.so timeout 300 $QB > $R/gpurun_out/r2b_timers.log 2>&1
KNHIP_Q4=1 KNHIP_LIB=$R/knowhere_amd/libknhip_prof8.so timeout 300 $QB >> $R/gpurun_out/r2b_timers.log 2>&1
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*" | sort -u > $R/gpurun_out/r2b_lds_counters.txt
run() { name=$1; shift; rm -rf /tmp/pb_$name; (KNHIP_Q4=1 timeout 600 rocprofv3 "$@" --output-format csv -d /tmp/pb_$name -- $QB) > /tmp/pb_$name.log 2>&1; python $R/tools/pmc_summary.py /tmp/pb_$name $R/gpurun_out/r2b_$name.json; tail -1 /tmp/pb_$name.log | cut -c1-300; }
run stats --kernel-trace --stats
run sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA
run fetch --pmc FETCH_SIZE
cat $R/gpurun_out/r2b_timers.log | grep -v amdgpu.ids
