#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3i; mkdir -p $out
for E in 4 -2 -4 -6 1 4; do
QUFLOW_HIP_SK_EPI_UNITS=$E timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -k "n64_golden or spot" > $out/pytest_$E.txt 2>&1 || { echo "pytest failed E=$E"; tail -20 $out/pytest_$E.txt; exit 1; }
QUFLOW_HIP_SK_EPI_UNITS=$E timeout -k 10 200 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('E=$E K=200', d['value'], d['roofline']['avg_launch_us'])"
done
