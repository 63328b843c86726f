#!/bin/bash
out=gpurun_out/r04f; mkdir -p $out
./tools/gemm_time 1024 > $out/gemm_time_1024.txt 2>&1; grep -v "tri32\|full" $out/gemm_time_1024.txt
QF_FUSED=1 ./tools/tri_probe_light 1024 > $out/tri_probe_light_1024.txt 2>&1; sed -n 5,30p $out/tri_probe_light_1024.txt
for E in 0 2 3 4 6; do
  for r in 1 2; do
  QUFLOW_HIP_SK_EPI_UNITS=$E timeout -k 10 120 python bench.py --no-side-runs --no-config3 --cpu-seconds 0 > $out/bench_E${E}_$r.json 2> $out/bench_E${E}_$r.err
  python - $out/bench_E${E}_$r.json $E <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("E", sys.argv[2], "%.1f"%d["value"])
PY
  done
done
