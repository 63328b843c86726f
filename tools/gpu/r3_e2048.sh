#!/bin/bash
export TMPDIR=/tmp
run() { env "$@" timeout -k 10 300 python bench.py --N $N --steps 60 --warmup 10 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('N=$N', '$*', d['value'], 'gemm2', r.get('second_product',{}).get('avg_launch_us'))"; }
N=2048
for e in 0 2 4 6 8 12; do run QUFLOW_HIP_SK_EPI_UNITS=$e; done
run A=1
N=1536
for e in 0 4 8; do run QUFLOW_HIP_SK_EPI_UNITS=$e; done
