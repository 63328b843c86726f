#!/bin/bash
out=gpurun_out/r04l; mkdir -p $out
for E in 2 0 -1 -2 -3 -4 -6; do
  for r in 1 2; do
  QUFLOW_HIP_SK_EPI_UNITS=$E timeout -k 10 120 python bench.py --no-side-runs --no-config3 --cpu-seconds 0 > $out/bench_E${E}_$r.json 2> $out/bench_E${E}_$r.err
  python - $out/bench_E${E}_$r.json $E <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("E", sys.argv[2], "%.1f"%d["value"])
PY
  done
done
