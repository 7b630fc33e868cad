#!/bin/bash
# lab builds for the bs = 32 floor probe (run HERE, before gpurun: the .so files travel with the snapshot):
#   (the no-split kernel's labs — libhqq_hip_kwave.so, libhqq_hip_kwnoarith.so — are built by tools/lab_kwave/build.sh)
#   tools/libhqq_hip_sknoarith.so   skinny.hip (both tiles) with -DSK_LAB_NOARITH -DSK_LAB_NOFIN (the split-K shape without arithmetic and without the finish)
#   tools/libhqq_hip_sknofin.so     skinny.hip with arithmetic, without the finish (what the three coherence trips cost)
set -e
cd "$(dirname "$0")/../hqq_amd/csrc"
make -j8 >/dev/null
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
$CC -DSK_LAB_NOARITH -DSK_LAB_NOFIN -c skinny.hip -o build/skinny_lab_noarith.o &
$CC -DSK_LAB_NOARITH -DSK_LAB_NOFIN -DSK_NARROW -c skinny.hip -o build/skinny_narrow_lab_noarith.o &
$CC -DSK_LAB_NOFIN -c skinny.hip -o build/skinny_lab_nofin.o &
$CC -DSK_LAB_NOFIN -DSK_NARROW -c skinny.hip -o build/skinny_narrow_lab_nofin.o &
wait
BASE=$(ls build/*.o | grep -v "_var_\|_lab")
link() { out=$1; shift; /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libhqq_hip_$out.so "$@"; echo built tools/libhqq_hip_$out.so; }
link sknoarith $(echo "$BASE" | grep -v "build/skinny") build/skinny_lab_noarith.o build/skinny_narrow_lab_noarith.o
link sknofin $(echo "$BASE" | grep -v "build/skinny") build/skinny_lab_nofin.o build/skinny_narrow_lab_nofin.o
