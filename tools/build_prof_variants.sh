#!/bin/bash
# phase-timer builds of pq_filter.hip with experiment knock-outs (KNHIP_P8_EXP bit mask): tools/prof/libknhip_prof_x<mask>.so
set -e
cd "$(dirname "$0")/../knowhere_amd/csrc"
make -s -j8
OBJS=$(ls build/*.o | grep -v "pq_filter\|pq_scan_q4\|_prof")
mkdir -p ../../tools/prof
for x in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -Wall -Wno-unused-function -DKNHIP_PHASE_TIMERS -DKNHIP_P8_EXP=$x -c pq_filter.hip -o build/pq_filter_prof_x$x.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/prof/libknhip_prof_x$x.so $OBJS build/pq_filter_prof_x$x.o build/pq_scan_q4.o
done
