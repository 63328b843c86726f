#!/bin/bash
# timing-only experiment: what would "no mirrors" buy?  fixed 4 iterations per step (values do not matter)
out=gpurun_out/r05_xp
mkdir -p $out
for N in 1024 2048; do
for xp in 0 1 2 3 0 3; do
  steps=300; [ $N = 2048 ] && steps=60
  QUFLOW_XP_NOMIRROR=$xp timeout -k 10 200 python bench.py --N $N --steps $steps --warmup 20 --fixed-iters 4 --cpu-seconds 0 --no-side-runs --no-config3 --no-kernel-events > $out/xp_${N}_$xp.json 2>/dev/null
  python -c "import json;d=json.load(open('$out/xp_${N}_$xp.json'));print('N=$N xp=$xp', round(d['value'],1), 'us/iteration', round(1e3*d['ms_per_step']/4,2))"
done; done 2>&1 | tee $out/summary.txt
