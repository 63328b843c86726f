#!/bin/bash
out=gpurun_out/r04x; mkdir -p $out
timeout -k 10 500 python -m pytest tests/test_hip_single.py -m gpu -x -q > $out/pytest_single.txt 2>&1; tail -3 $out/pytest_single.txt
for N in 333 1000 1500 1024 512; do
  K=$(( N <= 512 ? 400 : N <= 1024 ? 200 : 60 ))
  timeout -k 10 300 python bench.py --dtype c64 --N $N --steps $K --warmup 10 --cpu-seconds 0 --no-config3 --no-side-runs 2>> $out/sweep.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline'] or {}
print(json.dumps({'dtype': 'c64', 'N': $N, 'timesteps_per_s': round(d['value'],1), 'iterations_per_step': d['config']['iterations_per_step'], 'first_product_us': round(r.get('avg_launch_us',0),1), 'first_product_frac': round(r.get('frac',0),3)}))" | tee -a $out/sweep_c64_fix.jsonl
done
