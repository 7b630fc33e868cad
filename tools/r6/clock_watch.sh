#!/bin/bash
# What state is the box in while the 7B stack decodes?  Samples rocm-smi (clocks, power, temperature) every 0.25 s next to
# a 6000-step run of the headline leg, then prints the step time of that run.  Development aid: writes gpurun_out/r6clk/.
mkdir -p gpurun_out/r6clk; o=gpurun_out/r6clk
rocm-smi --showclocks --showpower --showtemp --showperflevel > $o/idle.txt 2>&1
( for i in $(seq 1 60); do echo "== t=$i"; rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -E "sclk|mclk|fclk|socclk|Power|Temperature \(Sensor (junction|memory|hbm)" ; sleep 0.25; done ) > $o/samples.txt &
S=$!
python bench.py --no-legs --no-cpu-baseline --steps 6000 --warmup 20 > $o/bench.json 2> $o/bench.err
wait $S
python - <<'PY'
import json,re
d=json.loads(open('gpurun_out/r6clk/bench.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'])
t=open('gpurun_out/r6clk/samples.txt').read()
for key in ('sclk','mclk','fclk','socclk','Power','junction','memory'):
    vals=[l.strip() for l in t.splitlines() if key in l]
    import collections
    c=collections.Counter(re.sub(r'GPU\[\d+\]\s*:\s*','',v) for v in vals)
    print(key, dict(c.most_common(6)))
PY
