#!/bin/bash
out=gpurun_out/r04r; mkdir -p $out
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "config3 or int8 or i8 or zgemm_i8" > $out/pytest_i8.txt 2>&1; tail -5 $out/pytest_i8.txt
bash tools/gpu/r4_ab3.sh $out 1024 oz2 ks -- --products i8x65
bash tools/gpu/r4_ab3.sh $out 2048 oz2 ks -- --products i8x65 --steps 60 --warmup 6
bash tools/gpu/r4_ab3.sh $out 1024 oz2 ks -- --products i8
