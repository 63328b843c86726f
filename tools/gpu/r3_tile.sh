#!/bin/bash
export TMPDIR=/tmp
run() { env "$@" timeout -k 10 300 python bench.py --N $N --steps 100 --warmup 10 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('N=$N', '$*', d['value'], 'gemm1', round(r['avg_launch_us'],1))"; }
for N in 768 832 896 960 1088 1152 1280 1536 1792; do run A=0; run QUFLOW_HIP_TILE64_MIN_N=4096; run QUFLOW_HIP_TILE64_MIN_N=4096 QUFLOW_HIP_TRI_MIN_N=4096; done
