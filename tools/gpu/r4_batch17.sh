#!/bin/bash
out=gpurun_out/r04t; mkdir -p $out
for g in 4 2 8; do
  for n in 512 256; do
    for r in 1 2; do
      QUFLOW_HIP_SOLVE_G=$g timeout -k 10 120 python bench.py --N $n --steps 400 --warmup 20 --no-side-runs --no-config3 --cpu-seconds 0 > $out/b_${g}_${n}_$r.json 2> $out/b_${g}_${n}_$r.err
      python -c "
import json; d=json.loads(open('$out/b_${g}_${n}_$r.json').read().strip().splitlines()[-1]); print('G=$g N=$n', round(d['value'],1))"
    done
  done
done
