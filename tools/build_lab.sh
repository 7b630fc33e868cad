#!/bin/bash
# lab build of the library: engine.hip with in-kernel time stamps (EN_LAB_TS); everything else as shipped
set -e
cd "$(dirname "$0")/../hqq_amd/csrc"
make -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DEN_LAB_TS ${LAB_FLAGS} -c engine.hip -o build/engine_lab.o
OBJS=$(ls build/*.o | grep -v engine)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libhqq_hip_lab.so $OBJS build/engine_lab.o
echo built tools/libhqq_hip_lab.so
