#!/bin/bash
# GPU call 2 of round 5: the folded decoder-block launches (parity, token identity, kernels per block, end to end), then the headline's
# profile + SQ counters on THIS box (kernel stats and bench line from the same box: they must add up)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_block_fold_gpu.py -x -q 2>&1 | tail -12 | tee $OUT/fold_tests.txt
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "fused or bucket or glue" 2>&1 | tail -12 | tee $OUT/model_tests.txt
# kernels per decoder block
ATT=sdpa bash tools/r4_e2e_kernels.sh > /dev/null 2>&1; cp gpurun_out/r4/e2e_kernels_per_block.txt $OUT/e2e_kernels_per_block.txt; cat $OUT/e2e_kernels_per_block.txt | head -40
# the default bench line (all legs, end to end with identity check and the glue A/B)
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-700 $OUT/bench_default.json; echo
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5/bench_default.json"))
print(json.dumps(d.get("end_to_end"), indent=1)[:3000])
for l in d.get("legs", []): print(l.get("name"), l.get("ms_per_step"), l.get("frac"), l.get("frac_mfma"), l.get("tflops"))
PY
# kernel stats of the same command on the same box
cd /tmp
run() { tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs "$@" > $OUT/${tag}_bench_under_rocprof.json 2> $OUT/$tag.err
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${tag}_kernel_stats.csv
  rm -rf $OUT/$tag; head -5 $OUT/${tag}_kernel_stats.csv | cut -c1-200; cut -c1-200 $OUT/${tag}_bench_under_rocprof.json; echo; }
run decode_int4
run decode_int4_bs32 --bs 32
run decode_int2 --nbits 2
cd $R
# SQ counters of the final int4 / int2 decode kernels (separate --pmc passes, kernel-trace only beside them)
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
G2="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
G3="SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVES"
for nb in 4 2; do
  echo "== int$nb decode stack, bs = 1 (gemv_f16_kernel / gemv_f16_xp2_kernel instantiations together; 128 launches per step, mean per dispatch)"
  bash tools/gpu_pmc_kernel.sh "bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-legs --no-graph --nbits $nb" gemv_f16 "$G1" "$G2" "$G3"
done 2>&1 | tee $OUT/sq_counters_decode.txt
