#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3s; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_single.py -x -q > $out/pytest_single.txt 2>&1; tail -5 $out/pytest_single.txt
run() { env "$@" timeout -k 10 300 python bench.py --dtype c64 --N $N --steps 400 --warmup 10 --cpu-seconds 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('c64 N=$N', '$*', d['value'], 'gemm2', r.get('second_product',{}).get('avg_launch_us'), 'solve', r.get('laplacian_inverse',{}).get('avg_launch_us'))"; }
for N in 128 256 512; do run QUFLOW_HIP_DEFER=0; run QUFLOW_HIP_DEFER=1; run QUFLOW_HIP_DEFER=0; run QUFLOW_HIP_DEFER=1; done
