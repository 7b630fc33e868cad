#!/bin/bash
# does the headline depend on how long the GPU has been busy before the timed region?  (clock ramp)
for w in 10 500 10 500; do
  python bench.py --no-legs --no-cpu-baseline --random-codes --steps 50 --warmup $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('warmup $w', d['ms_per_step'], d['roofline']['frac'])"
done
