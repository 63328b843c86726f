#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3p; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -k "zgemm or fixedpoint or isomp_n64 or protocols or literal or options or chunking or deferred or spot" > $out/pytest.txt 2>&1; tail -5 $out/pytest.txt
run() { env "$@" timeout -k 10 300 python bench.py --N $N --steps 400 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('N=$N', '$*', d['value'], 'gemm1', r['avg_launch_us'], r['frac'])"; }
for N in 128 256 512 704; do run QUFLOW_HIP_ZGEMM_KS=1; run QUFLOW_HIP_ZGEMM_KS=2; run QUFLOW_HIP_ZGEMM_KS=1; run QUFLOW_HIP_ZGEMM_KS=2; done
