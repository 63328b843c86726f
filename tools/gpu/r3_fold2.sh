#!/bin/bash
export TMPDIR=/tmp
run() { env "$@" timeout -k 10 300 python bench.py --N $N --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('N=$N', '$*', d['value'], 'solve', r.get('laplacian_inverse',{}).get('avg_launch_us'))"; }
for N in 768 1024; do
run A=0
run QUFLOW_HIP_SOLVE_FOLD=1
run QUFLOW_HIP_SOLVE_FOLD=1 QUFLOW_HIP_SOLVE_G=2
run QUFLOW_HIP_SOLVE_FOLD=1 QUFLOW_HIP_SOLVE_G=8
run A=1
done
N=2048
env QUFLOW_HIP_SOLVE_FOLD=0 timeout -k 10 300 python bench.py --N $N --steps 60 --warmup 10 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('N=$N FOLD=0', d['value'], 'solve', r.get('laplacian_inverse',{}).get('avg_launch_us'))"
env QUFLOW_HIP_SOLVE_FOLD=1 timeout -k 10 300 python bench.py --N $N --steps 60 --warmup 10 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('N=$N FOLD=1', d['value'], 'solve', r.get('laplacian_inverse',{}).get('avg_launch_us'))"
