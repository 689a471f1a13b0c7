#!/bin/bash
# Second GPU call of round 3 (after tools/gpu_round3_a.sh is green): rocprofv3 evidence for the C3 bench on the
# half-precision ADC prefilter path -- kernel stats + the separate PMC passes (HBM traffic, SQ issue, LDS), as
# tools/profile_bench.sh collects them -- and the traffic file entry for this kernel path.
# gpurun --timeout 2400 -- 'bash tools/gpu_round3_b.sh'
export KNHIP_PQF=1
export PROFILE_TAG=r03_c3_pqf
bash tools/profile_bench.sh
KEY="config=C3,nb=100000000,nlist=16384,nprobe=128,nq=10000,m=32,refine_k=100,gpus=1,pqf=1"
python tools/pmc_traffic.py gpurun_out/${PROFILE_TAG}_fetch.json gpurun_out/${PROFILE_TAG}_write.json "$KEY" pqf_kernel
cp profiles/bench_pmc_traffic.json gpurun_out/bench_pmc_traffic.json
