#!/bin/bash
# same-box A/B of the split-bf16 GEMM tile shapes: coarse stage at the C3 / C5 shapes, BRUTE_FORCE at 1M x 128
mkdir -p gpurun_out; export TMPDIR=/tmp
for L in tools/prof/libknhip_oldgemm.so knowhere_amd/libknhip.so tools/prof/libknhip_oldgemm.so knowhere_amd/libknhip.so; do
  echo "== $L"
  KNHIP_LIB=$L timeout 300 python tools/diag/coarse_only.py C3 2>&1 | tail -2
  KNHIP_LIB=$L timeout 300 python tools/diag/coarse_only.py C5 2>&1 | tail -2
  KNHIP_LIB=$L timeout 300 python tools/diag/bf_time.py 2>&1 | tail -3
done
