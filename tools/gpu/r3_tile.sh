#!/bin/bash
export TMPDIR=/tmp
run() { env "$@" timeout -k 10 300 python bench.py --dtype c64 --N $N --steps 100 --warmup 10 --cpu-seconds 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('c64 N=$N', '$*', d['value'], 'gemm1', round(r['avg_launch_us'],1), 'gemm2', round(r['second_product']['avg_launch_us'],1))"; }
for N in 1000 900 1500; do run A=0; run QUFLOW_HIP_C64_TILE64_MIN_N=4096; done
run2() { env "$@" timeout -k 10 300 python bench.py --N $N --steps 100 --warmup 10 --cpu-seconds 0 --no-config3 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('c128 N=$N', '$*', d['value'], 'gemm1', round(r['avg_launch_us'],1), 'gemm2', round(r['second_product']['avg_launch_us'],1))"; }
for N in 1000 1500; do run2 A=0; run2 QUFLOW_HIP_TILE64_MIN_N=4096; done
