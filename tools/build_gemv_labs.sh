#!/bin/bash
# builds the stand-alone decode-kernel lab (tools/gemv_lab.hip) in its variants -> tools/gemv_lab_<name>.bin (run them with tools/r2_lab_gemv.sh):
#   plain       the kernel as shipped              ts          + in-kernel time stamps (gemv_lab_ts.bin N K [layers] [launches])
#   preload     kernel-argument preload, layer-0 fast path (GV_LAB_PRELOAD)          preload_ts   the same with time stamps
#   w8          8 waves x 2 workgroups per CU instead of 4 x 4
set -e
cd "$(dirname "$0")/../hqq_amd/csrc"
make -j8 >/dev/null
OBJS=$(ls build/*.o | grep -v "_var_\|_lab\|build/gemv.o")
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form"
build() {   # name, flags
  /opt/rocm/bin/hipcc $F $2 -c ../../tools/gemv_lab.hip -o build/gemv_lab_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $OBJS build/gemv_lab_$1.o -o ../../tools/gemv_lab_$1.bin
  echo built tools/gemv_lab_$1.bin
}
for v in ${@:-plain ts preload preload_ts w8}; do
  case $v in
    plain) build plain "" ;;
    ts) build ts "-DGV_LAB_TS" ;;
    preload) build preload "-DGV_LAB_PRELOAD -mllvm -amdgpu-kernarg-preload-count=16" ;;
    preload_ts) build preload_ts "-DGV_LAB_PRELOAD -DGV_LAB_TS -mllvm -amdgpu-kernarg-preload-count=16" ;;
    w8) build w8 "-DGV_WAVES_PER_WG=8 -DGV_WG_PER_CU=2" ;;
    *) echo "unknown variant $v"; exit 1 ;;
  esac
done
