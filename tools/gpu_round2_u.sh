#!/bin/bash
# final evidence of the round: full GPU suite, C++ node flow, the three contract benches (device + host boundary +
# CPU reference), rocprofv3 stats and PMC passes (FETCH / WRITE / MFMA) of the C2 and C5-shaped benches
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/r2u_pytest.log 2>&1; tail -4 gpurun_out/r2u_pytest.log | cut -c1-300
(cd knowhere_amd/host && timeout 600 ./test_hip_index) > gpurun_out/r2u_cpp_node_test.log 2>&1; tail -3 gpurun_out/r2u_cpp_node_test.log | cut -c1-300
timeout 900 python bench.py > gpurun_out/r2u_bench_c3.log 2>&1; tail -1 gpurun_out/r2u_bench_c3.log | cut -c1-1500
timeout 600 python bench.py --config C2 --steps 10 --warmup 3 > gpurun_out/r2u_bench_c2.log 2>&1; tail -1 gpurun_out/r2u_bench_c2.log | cut -c1-1500
timeout 600 python bench.py --config C5 --nb 8000000 --nlist 8192 --nprobe 64 --steps 10 --warmup 3 > gpurun_out/r2u_bench_c5_8m.log 2>&1; tail -1 gpurun_out/r2u_bench_c5_8m.log | cut -c1-1500
timeout 1200 python bench.py --config C5 --steps 5 --warmup 2 > gpurun_out/r2u_bench_c5.log 2>&1; tail -1 gpurun_out/r2u_bench_c5.log | cut -c1-2500
cd /tmp
run() { tag=$1; name=$2; args=$3; shift 3; rm -rf /tmp/pb_$tag_$name; (timeout 600 rocprofv3 "$@" --output-format csv -d /tmp/pb_${tag}_$name -- python $R/bench.py $args --steps 3 --warmup 1 --cpu-queries 0 --host-steps 0) > /tmp/pb_${tag}_$name.log 2>&1; python $R/tools/pmc_summary.py /tmp/pb_${tag}_$name $R/gpurun_out/r2u_${tag}_rocprof_$name.json | tail -1; }
C2="--config C2"
C58="--config C5 --nb 8000000 --nlist 8192 --nprobe 64"
run c2 stats "$C2" --kernel-trace --stats
run c2 fetch "$C2" --pmc FETCH_SIZE
run c2 write "$C2" --pmc WRITE_SIZE
run c2 mfma "$C2" --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES
run c5_8m stats "$C58" --kernel-trace --stats
run c5_8m fetch "$C58" --pmc FETCH_SIZE
run c5_8m write "$C58" --pmc WRITE_SIZE
run c5_8m mfma "$C58" --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES
