#!/bin/bash
# lab: tools/libhqq_hip_kwave.so = the library + the quarantined batched-decode kernel without a cross-workgroup K split (kwave.hip) as the
# default for 5..64 rows (HQQ_OPT_BATCH_SPLITK = 2048 forces skinny.hip; HQQ_OPT_SKINNY_KS(n) forces n 16-row tiles per workgroup);
# tools/libhqq_hip_kwnoarith.so = the same with -DKW_LAB_NOARITH (loads only).  Run HERE before gpurun (the .so files travel with the snapshot).
set -e
cd "$(dirname "$0")/../../hqq_amd/csrc"
make -j8 >/dev/null
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I."
for nb in 8 4 3 2; do
  $CC -DKW_NBITS=$nb -c ../../tools/lab_kwave/kwave.hip -o build/kwave_${nb}_lab.o &
  $CC -DKW_LAB_NOARITH -DKW_NBITS=$nb -c ../../tools/lab_kwave/kwave.hip -o build/kwave_${nb}_lab_noarith.o &
done
$CC -mllvm -amdgpu-mfma-vgpr-form -DGV_LAB_PRELOAD -mllvm -amdgpu-kernarg-preload-count=16 -DHQQ_LAB_KWAVE -c gemv.hip -o build/gemv_lab_kwave.o &
wait
BASE=$(ls build/*.o | grep -v "_var_\|_lab" | grep -v "build/gemv.o")
for v in "" _noarith; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libhqq_hip_kw${v#_}${v:+}.so $BASE build/gemv_lab_kwave.o build/kwave_8_lab$v.o build/kwave_4_lab$v.o build/kwave_3_lab$v.o build/kwave_2_lab$v.o
done
mv ../../tools/libhqq_hip_kw.so ../../tools/libhqq_hip_kwave.so
echo built tools/libhqq_hip_kwave.so tools/libhqq_hip_kwnoarith.so
