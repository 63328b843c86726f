#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3x; mkdir -p $out
timeout -k 10 800 python -m pytest tests/test_hip_single.py tests/test_hip_parity.py -x -q -k "c64 or single or complex64" > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
run() { env "$@" timeout -k 10 300 python bench.py --dtype c64 --N $N --steps 100 --warmup 10 --cpu-seconds 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('c64 N=$N', '$*', d['value'], 'gemm1', round(r['avg_launch_us'],1), 'gemm2', round(r['second_product']['avg_launch_us'],1), r['second_product']['kernel'][:16])"; }
for N in 512 768 896 1000 1024 1056 1280 1536 2048; do run A=0; done
