#!/bin/bash
out=gpurun_out/r04n; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
bash tools/gpu/r4_ab3.sh $out 1024 sched1 mod --
bash tools/gpu/r4_ab3.sh $out 512 sched1 mod -- --steps 400 --warmup 20
QF_FUSED=1 ./tools/tri_probe_light 1024 > $out/tri_probe_light_1024.txt 2>&1; sed -n 9,18p $out/tri_probe_light_1024.txt
