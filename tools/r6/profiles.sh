#!/bin/bash
# round 6: every profile artefact of the final build in ONE call (one box).   usage (on the GPU box): bash tools/r6/profiles.sh
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6prof; mkdir -p $OUT; export TMPDIR=/tmp
cd $R
( time python bench.py > $OUT/r06_bench_default.json 2>$OUT/bench.err ) 2> $OUT/bench_time.txt
for cfg in "int4:--nbits 4" "int4_bs32:--nbits 4 --bs 32" "int2:--nbits 2" "int3:--nbits 3"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs $args > $OUT/r06_decode_${tag}_bench_under_rocprof.json 2>$OUT/prof_$tag.err )
  f=$(find $OUT/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "^\"Name\"|gemv|skinny" $f | cut -c1-400 > $OUT/r06_decode_${tag}_kernel_stats.csv
  if [ "$tag" = int4 ]; then python tools/r6/per_launch.py $OUT/prof_$tag > $OUT/r06_per_launch.txt 2>&1; fi
  rm -rf $OUT/prof_$tag
done
bash tools/gpu_pmc.sh r6pmc > $OUT/pmc.log 2>&1; cp gpurun_out/r6pmc/pmc_summary.json $OUT/r06_decode_int4_pmc_summary.json 2>/dev/null
bash tools/r6/gpu_sq.sh > $OUT/sq.log 2>&1; cat gpurun_out/r6sq/sq_int4.txt gpurun_out/r6sq/sq_int2.txt > $OUT/r06_sq_counters_raw.txt 2>/dev/null
HQQ_AMD_LIB=$R/tools/r6/libhqq_hip_ts.so python tools/r6/ts_run.py 2>&1 | grep -v amdgpu.ids > $OUT/r06_decode_timeline.txt
./tools/r6/ramp_probe.bin > $OUT/r06_wave_launch_ramp_probe.txt 2>&1
./tools/r6/instr_bench.bin > $OUT/r06_instr_issue_rates.txt 2>&1
./tools/r6/range_probe.bin > $OUT/r06_buffer_range_check_probe.txt 2>&1
cat $OUT/bench_time.txt; head -c 400 $OUT/r06_bench_default.json; echo; cat $OUT/r06_per_launch.txt; cat $OUT/r06_decode_int4_kernel_stats.csv | cut -c1-200; cat $OUT/r06_decode_int4_pmc_summary.json | head -30
