#!/bin/bash
out=gpurun_out/r04e; mkdir -p $out
./tools/gemm_time 1024 > $out/gemm_time_1024.txt 2>&1; cat $out/gemm_time_1024.txt
QF_FUSED=1 ./tools/tri_probe_light 1024 > $out/tri_probe_light_1024.txt 2>&1; tail -22 $out/tri_probe_light_1024.txt
bash tools/gpu/r4_ab.sh $out 1024
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "tri or fused or deferred or n64_golden or vs_oracle_large" > $out/pytest_tri.txt 2>&1; tail -3 $out/pytest_tri.txt
