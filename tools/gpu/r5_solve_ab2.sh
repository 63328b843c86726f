#!/bin/bash
out=gpurun_out/r05_solve; mkdir -p $out
for n in 512 256 384; do
  echo "== base N=$n"; timeout -k 5 60 tools/solve_probe_base $n 2>&1 | head -7
  echo "== new  N=$n"; timeout -k 5 60 tools/solve_probe $n 2>&1 | head -22
done > $out/solve_probe_ab2.txt 2>&1
grep -E "==|full|no trace  |life" $out/solve_probe_ab2.txt
bash tools/gpu/r5_ab_lib.sh solve512b --N 512 --steps 400 --warmup 20
bash tools/gpu/r5_ab_lib.sh solve256 --N 256 --steps 400 --warmup 20
bash tools/gpu/r5_suite.sh
