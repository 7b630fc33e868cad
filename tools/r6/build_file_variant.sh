#!/bin/bash
# lab: tools/r6/libhqq_hip_<name>.so with extra flags on ONE source file:  tools/r6/build_file_variant.sh <name> <file.hip> "<flags>"
set -e
name=$1; src=$2; flags=$3
root=$(cd "$(dirname "$0")/../.." && pwd)
cd $root/hqq_amd/csrc
make -j8 >/dev/null 2>&1
base=$(basename $src .hip)
d=build/var_$name; mkdir -p $d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $flags -c $src -o $d/$base.o 2>/dev/null
OBJS=$(ls build/*.o | grep -v "build/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/r6/libhqq_hip_$name.so $OBJS $d/$base.o
echo built tools/r6/libhqq_hip_$name.so
