#!/bin/bash
# validation of HEAD: full GPU suite (new goldens, DeserializeFromFile in the C++ flow), smoke(), one mscan bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/r2w_pytest.log 2>&1; tail -4 gpurun_out/r2w_pytest.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2w_smoke.log 2>&1; tail -4 gpurun_out/r2w_smoke.log | cut -c1-300
timeout 600 python bench.py --config C5 --nb 8000000 --nlist 8192 --nprobe 64 --steps 10 --warmup 3 --cpu-queries 0 --host-steps 0 > gpurun_out/r2w_bench_c5_8m.log 2>&1; tail -1 gpurun_out/r2w_bench_c5_8m.log | cut -c1-1800
