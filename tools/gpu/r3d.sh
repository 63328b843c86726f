#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3d
timeout -k 10 900 python -m pytest tests/test_native_comm.py -x -q -m gpu > gpurun_out/r3d/native.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/r3d/native.log
timeout -k 10 300 python bench.py --N 512 --steps 200 --warmup 20 --no-config3 --cpu-seconds 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N512 line', d['value'])"
timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-config3 --cpu-seconds 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('other sizes', {k:v['value'] for k,v in d['other_sizes'].items()})"
