#!/bin/bash
# lab: build tools/libhqq_hip_<name>.so with extra flags for ONE source file:  tools/build_variant.sh <name> <file.hip> "<flags>"
set -e
cd "$(dirname "$0")/../hqq_amd/csrc"
make -j8 >/dev/null
base=$(basename $2 .hip)
extra=""; [ "$base" = "quantize" ] && extra="-ffp-contract=off"; [ "$base" = "gemv" ] && extra="-mllvm -amdgpu-mfma-vgpr-form -DGV_LAB_PRELOAD -mllvm -amdgpu-kernarg-preload-count=16"; [ "$base" = "gemv_chain" ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra $3 -c $2 -o build/${base}_var_$1.o
OBJS=$(ls build/*.o | grep -v "_var_\|_lab" | grep -v "build/${base}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libhqq_hip_$1.so $OBJS build/${base}_var_$1.o
echo built tools/libhqq_hip_$1.so
