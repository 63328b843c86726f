#!/bin/bash
out=gpurun_out/r04j; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
for n in 512 1024 2048; do ./tools/solve_probe $n > $out/solve_probe_$n.txt 2>&1; head -3 $out/solve_probe_$n.txt; done
bash tools/gpu/r4_ab.sh $out 512 --steps 400 --warmup 20
bash tools/gpu/r4_ab.sh $out 1024
bash tools/gpu/r4_ab.sh $out 768 --steps 300 --warmup 20
