#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3k; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_faults.py -x -q -k "deferred or fault or protocols" > $out/pytest.txt 2>&1; tail -8 $out/pytest.txt
run() { env "$@" timeout -k 10 300 python bench.py --N $N --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('N=$N', '$*', d['value'], d['config']['iterations_per_step'])"; }
for N in 768 1024; do
run A=0
run QUFLOW_HIP_DEFER=tri
run A=1
run QUFLOW_HIP_DEFER=tri
done
