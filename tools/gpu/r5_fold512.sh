#!/bin/bash
# folded walk slots below N = 768 with this round's solve (QUFLOW_HIP_SOLVE_FOLD=1: from N = 256): stepper A/B on one box
for rep in 1 2 3; do for f in default 1; do
  if [ $f = default ]; then unset QUFLOW_HIP_SOLVE_FOLD; else export QUFLOW_HIP_SOLVE_FOLD=1; fi
  for n in 512 256; do
    timeout -k 10 200 python bench.py --N $n --steps 400 --warmup 20 --cpu-seconds 0 --no-side-runs --no-config3 --no-kernel-events > /tmp/f.json 2>/tmp/f.err
    python -c "import json;d=json.load(open('/tmp/f.json'));print('fold=$f N=$n rep $rep', round(d['value'],1))"
  done
done; done
