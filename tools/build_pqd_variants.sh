#!/bin/bash
# production builds of pq_decode.hip with compile-time knock-outs (PD_EXP bit mask, see pqd_scan): tools/prof/libknhip_pqd_x<mask>.so
# usage: tools/build_pqd_variants.sh 1 2 4 8 16 ...   (then: KNHIP_LIB=tools/prof/libknhip_pqd_x<mask>.so python bench.py ...)
set -e
cd "$(dirname "$0")/../knowhere_amd/csrc"
make -s -j8
OBJS=$(ls build/*.o | grep -v "pq_decode\|_prof")
mkdir -p ../../tools/prof
for x in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -DPD_EXP=$x -c pq_decode.hip -o build/pq_decode_x$x.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/prof/libknhip_pqd_x$x.so $OBJS build/pq_decode_x$x.o
done
