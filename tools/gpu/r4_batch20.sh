#!/bin/bash
out=gpurun_out/r04y; mkdir -p $out
bash tools/gpu/r4_ab3.sh $out 1000 base cur -- --dtype c64 --steps 200 --warmup 10
bash tools/gpu/r4_ab3.sh $out 1500 base cur -- --dtype c64 --steps 60 --warmup 10
bash tools/gpu/r4_ab3.sh $out 333 base cur -- --dtype c64 --steps 400 --warmup 10
bash tools/gpu/r4_ab3.sh $out 1000 base cur -- --steps 200 --warmup 10
