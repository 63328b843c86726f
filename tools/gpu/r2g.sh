#!/bin/bash
out=gpurun_out/r2g; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -k "poisson or solvers or f1 or n64_golden or spot or large or factor" > $out/pytest.txt 2>&1 || { echo "pytest failed"; tail -40 $out/pytest.txt; exit 1; }
tail -2 $out/pytest.txt
echo "== solve probe"; timeout -k 10 60 tools/solve_probe 1024 | tee $out/solve_probe.txt
echo "== N=512"; timeout -k 10 60 tools/solve_probe 512 | head -1
echo "== N=2048 L=32"; timeout -k 10 60 tools/solve_probe 2048 | head -1; echo "== N=2048 L=16"; QUFLOW_HIP_SOLVE_L=16 timeout -k 10 60 tools/solve_probe 2048 | head -1
for e in 0 4 8; do QUFLOW_HIP_SK_EPI_UNITS=$e timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-config3 --no-side-runs --cpu-seconds 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('E=$e K=200', d['value'], d['roofline']['avg_launch_us'])"; done
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-config3 --no-side-runs --cpu-seconds 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=20', d['value'], d['roofline']['avg_launch_us'])"
timeout -k 10 200 python bench.py --N 2048 --steps 40 --warmup 5 --no-config3 --no-side-runs --cpu-seconds 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=2048', d['value'], d['roofline']['avg_launch_us'])"
