#!/bin/bash
# size sweep of the stepper, complex128 and complex64, single trajectory: one line per size
export TMPDIR=/tmp
out=gpurun_out/r4w; mkdir -p $out; rm -f $out/sweep.jsonl
for dt in c128 c64; do for N in 128 256 333 384 512 640 704 768 800 896 1000 1024 1056 1280 1500 1536 1792 2048; do
  K=$(( N <= 512 ? 400 : N <= 1024 ? 200 : 60 ))
  timeout -k 10 300 python bench.py --dtype $dt --N $N --steps $K --warmup 10 --cpu-seconds 0 --no-config3 --no-side-runs 2>> $out/sweep.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline'] or {}
print(json.dumps({'dtype': '$dt', 'N': $N, 'timesteps_per_s': round(d['value'],1), 'iterations_per_step': d['config']['iterations_per_step'], 'first_product_us': round(r.get('avg_launch_us',0),1), 'first_product_frac': round(r.get('frac',0),3)}))" | tee -a $out/sweep.jsonl
done; done
