#!/bin/bash
out=gpurun_out/r04u; mkdir -p $out
for E in 2 0 4 8; do
  for r in 1 2; do
    QUFLOW_HIP_SK_EPI_UNITS=$E timeout -k 10 200 python bench.py --N 2048 --steps 60 --warmup 6 --no-side-runs --no-config3 --cpu-seconds 0 > $out/b_E${E}_$r.json 2> $out/b_E${E}_$r.err
    python -c "
import json; d=json.loads(open('$out/b_E${E}_$r.json').read().strip().splitlines()[-1]); print('N=2048 E=$E', round(d['value'],2))"
  done
done
for E in 2 0 4; do
  for n in 1536 960; do
    QUFLOW_HIP_SK_EPI_UNITS=$E timeout -k 10 200 python bench.py --N $n --steps 100 --warmup 10 --no-side-runs --no-config3 --cpu-seconds 0 > $out/b_E${E}_n$n.json 2> $out/b_E${E}_n$n.err
    python -c "
import json; d=json.loads(open('$out/b_E${E}_n$n.json').read().strip().splitlines()[-1]); print('N=$n E=$E', round(d['value'],2))"
  done
done
