#!/bin/bash
# lab: tools/r6/libhqq_hip_<name>.so with extra flags on every compilation of the decode kernel text (gemv.hip, gemv_w3s.hip, gemv_block.hip x 3)
#   tools/r6/build_gv_variant.sh <name> "<flags>"
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/../.." && pwd)
cd $root/hqq_amd/csrc
make -j8 >/dev/null 2>&1
d=build/var_$name; mkdir -p $d
C="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-kernarg-preload-count=16 $flags"
$C -c gemv.hip -o $d/gemv.o 2>/dev/null &
$C -c gemv_w3s.hip -o $d/gemv_w3s.o 2>/dev/null &
for b in 4 3 2; do $C -ffp-contract=off -DGB_NBITS=$b -c gemv_block.hip -o $d/gemv_block_$b.o 2>/dev/null & done
wait
OBJS=$(ls build/*.o | grep -v "build/gemv.o\|build/gemv_w3s.o\|build/gemv_block_")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/r6/libhqq_hip_$name.so $OBJS $d/*.o
echo built tools/r6/libhqq_hip_$name.so
