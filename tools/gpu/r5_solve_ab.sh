#!/bin/bash
# the solve's trace handling / scan width change: tests of the solve, probe before/after, stepper A/B at three sizes
out=gpurun_out/r05_solve; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "poisson or solve or laplac" 2>&1 | tail -5
for n in 512 1024 2048 256; do
  echo "== base N=$n"; timeout -k 5 60 tools/solve_probe_base $n 2>&1 | head -7
  echo "== new  N=$n"; timeout -k 5 60 tools/solve_probe $n 2>&1 | head -22
done > $out/solve_probe_ab.txt 2>&1
grep -E "==|full|no trace  " $out/solve_probe_ab.txt
bash tools/gpu/r5_ab_lib.sh solve512 --N 512 --steps 400 --warmup 20
bash tools/gpu/r5_ab_lib.sh solve1024 --N 1024 --steps 200 --warmup 10
bash tools/gpu/r5_ab_lib.sh solve2048 --N 2048 --steps 60 --warmup 6
