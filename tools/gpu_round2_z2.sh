#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests -x -q -m gpu > gpurun_out/r2z2_pytest.log 2>&1; tail -3 gpurun_out/r2z2_pytest.log | cut -c1-300
