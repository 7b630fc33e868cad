#!/bin/bash
# round 6, call 1: buffer range-check probe; baseline headline x3; no-dependency study; int2/int3
OUT=gpurun_out/r6a; mkdir -p $OUT
./tools/r6/range_probe.bin > $OUT/range_probe.txt 2>&1
for rep in 1 2 3; do
  python bench.py --no-legs --no-cpu-baseline --steps 30 --warmup 5 --random-codes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base', d['ms_per_step'], d['roofline']['frac'])"
done > $OUT/base.txt
python bench.py --no-legs --no-cpu-baseline --steps 30 --warmup 5 --random-codes --streams 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams3', d['ms_per_step'], d['roofline']['frac'])" >> $OUT/base.txt
python bench.py --no-legs --no-cpu-baseline --steps 30 --warmup 5 --random-codes --nbits 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('int2', d['ms_per_step'], d['roofline']['frac'])" >> $OUT/base.txt
python bench.py --no-legs --no-cpu-baseline --steps 30 --warmup 5 --random-codes --gemv-mode factored 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('factored', d['ms_per_step'], d['roofline']['frac'])" >> $OUT/base.txt
python bench.py --no-legs --no-cpu-baseline --steps 10 --warmup 3 --random-codes --workload decode70b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('70b', d['ms_per_step'], d['roofline']['frac'])" >> $OUT/base.txt
cat $OUT/range_probe.txt $OUT/base.txt
