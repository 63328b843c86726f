#!/bin/bash
out=gpurun_out/r04g; mkdir -p $out
run() { # name, env...
  name=$1; shift
  for r in 1 2; do
    env "$@" timeout -k 10 120 python bench.py --no-side-runs --no-config3 --cpu-seconds 0 > $out/bench_${name}_$r.json 2> $out/bench_${name}_$r.err
    python - $out/bench_${name}_$r.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "%.1f"%d["value"])
PY
  done
}
run order0_E2 QUFLOW_HIP_TRI_ORDER=0 QUFLOW_HIP_SK_EPI_UNITS=2
run order4_E2 QUFLOW_HIP_TRI_ORDER=4 QUFLOW_HIP_SK_EPI_UNITS=2
run order2_E2 QUFLOW_HIP_TRI_ORDER=2 QUFLOW_HIP_SK_EPI_UNITS=2
run order8_E2 QUFLOW_HIP_TRI_ORDER=8 QUFLOW_HIP_SK_EPI_UNITS=2
run order4_E4 QUFLOW_HIP_TRI_ORDER=4 QUFLOW_HIP_SK_EPI_UNITS=4
run order0_E2b QUFLOW_HIP_TRI_ORDER=0 QUFLOW_HIP_SK_EPI_UNITS=2
timeout -k 10 200 python -m pytest tests -m gpu -x -q -k "vs_oracle_large or headline" > $out/pytest_order0.txt 2>&1; tail -2 $out/pytest_order0.txt
QUFLOW_HIP_TRI_ORDER=4 timeout -k 10 200 python -m pytest tests -m gpu -x -q -k "vs_oracle_large or headline or tri" > $out/pytest_order4.txt 2>&1; tail -2 $out/pytest_order4.txt
