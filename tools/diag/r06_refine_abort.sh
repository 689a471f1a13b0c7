#!/bin/bash
# the test that aborted the round-6 suite, alone, under each prefilter form
mkdir -p gpurun_out; export TMPDIR=/tmp
for form in "" half int8; do
  echo "== KNHIP_PQF_FORM='$form'"
  KNHIP_PQF_FORM=$form timeout 600 python -m pytest tests/test_gpu_refine_rows.py::test_refine_rows_at_scale_properties -x -q -s 2>&1 | grep -v "^  File" | tail -15
done
