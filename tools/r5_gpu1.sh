#!/bin/bash
# first GPU call of round 5: parity of the new batched-decode kernel, then the bs = 32 probe (shipped / no-arithmetic labs), then bench legs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kwave_gpu.py -x -q 2>&1 | tail -15 | tee gpurun_out/r5/kwave_tests.txt
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_w3s_gpu.py -x -q -k "skinny or w3s" 2>&1 | tail -8 | tee gpurun_out/r5/skinny_tests.txt
(timeout 300 python tools/r5_bs32_probe.py 4 8,32,64
 for v in kwnoarith sknoarith sknofin; do HQQ_AMD_LIB=$PWD/tools/libhqq_hip_$v.so timeout 300 python tools/r5_bs32_probe.py 4 32; done
 timeout 200 python tools/r5_bs32_probe.py 2 32
 timeout 200 python tools/r5_bs32_probe.py 3 32) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/bs32_probe.txt
timeout 600 python bench.py --bs 32 --no-legs --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | tee gpurun_out/r5/bench_bs32.json
