#!/bin/bash
# the round's closing run on one box: the whole GPU suite, smoke(), then every profile artefact (tools/r6/profiles.sh)
mkdir -p gpurun_out/r6final
python -m pytest tests -m gpu -q > gpurun_out/r6final/pytest_gpu_full.txt 2>&1; tail -3 gpurun_out/r6final/pytest_gpu_full.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6final/smoke.txt 2>&1; tail -2 gpurun_out/r6final/smoke.txt
bash tools/r6/profiles.sh > gpurun_out/r6final/profiles_log.txt 2>&1; tail -40 gpurun_out/r6final/profiles_log.txt
