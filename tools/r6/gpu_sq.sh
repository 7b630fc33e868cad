#!/bin/bash
# SQ counters of the bs = 1 decode kernels (separate --pmc passes, --kernel-trace only beside them) for int4 and int2
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6sq; mkdir -p $OUT
export TMPDIR=/tmp
for NB in 4 2; do
  cd /tmp
  i=0
  for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/nb${NB}_p$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-legs --no-graph --nbits $NB > /dev/null 2>$OUT/err_${NB}_$i.log
  done
  cd $GRAFT_REPO_ROOT
  python - <<PY > $OUT/sq_int$NB.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$OUT/nb${NB}_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemv_f16" not in r["Kernel_Name"]: continue
        k = r["Counter_Name"]; agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
print("== int$NB decode stack, bs = 1 (gemv_f16_kernel / gemv_f16_xp2_kernel instantiations together; 128 launches per step, mean per dispatch)")
for k, (n, v) in sorted(agg.items()): print(f"{k:32s} mean per dispatch {v / n:16.1f}  ({n} dispatches)")
PY
  cat $OUT/sq_int$NB.txt
done
rm -rf $OUT/nb*_p*
