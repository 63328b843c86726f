#!/bin/bash
out=gpurun_out/r04o; mkdir -p $out
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "config3 or int8 or i8 or zgemm_i8" > $out/pytest_i8.txt 2>&1; tail -3 $out/pytest_i8.txt
for p in i8x65 i8x6 i8; do
 bash tools/gpu/r4_ab3.sh $out 1024 mod oz2 -- --products $p | sed "s/^/$p /"
done
bash tools/gpu/r4_ab3.sh $out 2048 mod oz2 -- --products i8x65 --steps 60 --warmup 6
