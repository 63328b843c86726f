#!/bin/bash
out=gpurun_out/r04k; mkdir -p $out
bash tools/gpu/r4_ab3.sh $out 512 base t32off t32on -- --steps 400 --warmup 20
bash tools/gpu/r4_ab3.sh $out 256 base t32off t32on -- --steps 400 --warmup 20
bash tools/gpu/r4_ab3.sh $out 768 base t32off t32on -- --steps 300 --warmup 20
