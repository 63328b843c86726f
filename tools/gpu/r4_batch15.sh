#!/bin/bash
out=gpurun_out/r04p; mkdir -p $out
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "config3 or int8 or i8 or zgemm_i8 or solve_poisson" > $out/pytest_i8.txt 2>&1; tail -3 $out/pytest_i8.txt
bash tools/gpu/r4_ab3.sh $out 1024 oz2 merge -- --products i8x65
bash tools/gpu/r4_ab3.sh $out 1024 oz2 merge -- --products i8x6f
bash tools/gpu/r4_ab3.sh $out 2048 oz2 merge -- --products i8x65 --steps 60 --warmup 6
bash tools/gpu/r4_ab3.sh $out 1024 oz2 merge --
