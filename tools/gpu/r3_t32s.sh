#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3t; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_faults.py -x -q -k "tri32 or protocols or deferred or fault or ensemble" > $out/pytest.txt 2>&1; tail -5 $out/pytest.txt
run() { env "$@" timeout -k 10 300 python bench.py --N $N --steps 400 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('N=$N', '$*', d['value'], 'gemm2', r.get('second_product',{}).get('avg_launch_us'))"; }
for N in 256 512 704; do for sp in 2,1 2,2 4,2 4,4 4,1; do run QUFLOW_HIP_TRI32_SPLIT=$sp; done; run A=1; run QUFLOW_HIP_TRI32_SPLIT=4,4 QUFLOW_HIP_DEFER=0; done
