#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3f; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1 || { echo "pytest failed"; tail -60 $out/pytest_gpu.txt; }
tail -3 $out/pytest_gpu.txt
for n in 512 1024 2048; do echo "== N=$n" >> $out/solve_probe.txt; timeout -k 10 60 tools/solve_probe $n >> $out/solve_probe.txt 2>&1; done
grep -E "^==|full " $out/solve_probe.txt
timeout -k 10 300 python bench.py --N 512 --steps 400 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=512 K=400', d['value'])"
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=1024 K=200', d['value'], d['roofline']['frac'])"
timeout -k 10 300 python bench.py --N 2048 --steps 60 --warmup 6 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=2048 K=60', d['value'], d['roofline']['frac'])"
