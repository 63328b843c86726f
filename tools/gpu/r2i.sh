#!/bin/bash
out=gpurun_out/r2i; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1 || { echo "pytest failed"; tail -40 $out/pytest_gpu.txt; exit 1; }
tail -2 $out/pytest_gpu.txt
for rep in 1 2 3; do
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-config3 --no-side-runs --cpu-seconds 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=20', d['value'], d['roofline']['avg_launch_us'], d['roofline']['launches_timed_with_events'])"
done
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-config3 --no-side-runs --cpu-seconds 0 --no-kernel-events | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=20 no events', d['value'])"
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-config3 --no-side-runs --cpu-seconds 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=200', d['value'], d['roofline']['avg_launch_us'])"
