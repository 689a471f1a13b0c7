#!/bin/bash
# phase timers of pqi8_kernel under the knock-out experiments (garbage results in some: timing only)
mkdir -p gpurun_out
export TMPDIR=/tmp
for x in "$@"; do
  KNHIP_LIB=tools/prof/libknhip_prof_x$x.so timeout 600 python bench.py --steps 2 --warmup 1 --cpu-queries 0 --host-steps 0 --extra none > gpurun_out/r05_pqi8_exp_x$x.log 2>&1
  echo "== experiment mask $x"; grep "pqf timers" gpurun_out/r05_pqi8_exp_x$x.log | awk '!seen[$0]++' | grep -A8 "int8" | sed -n '2,9p' | cut -c1-120
done
