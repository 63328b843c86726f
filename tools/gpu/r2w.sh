#!/bin/bash
export TMPDIR=/tmp
for x in 1 2 0; do echo "== XCD order $x"; QUFLOW_HIP_SOLVE_XCD=$x timeout -k 10 60 tools/solve_probe 1024 | head -1; QUFLOW_HIP_SOLVE_XCD=$x timeout -k 10 60 tools/solve_probe 2048 | head -1; done
for x in 1 2; do QUFLOW_HIP_SOLVE_XCD=$x timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-config3 --no-side-runs --cpu-seconds 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('XCD=$x K=200', d['value'])"; done
