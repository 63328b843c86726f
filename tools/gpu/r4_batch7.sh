#!/bin/bash
out=gpurun_out/r04h; mkdir -p $out
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "tri or fused or deferred or n64_golden or vs_oracle_large or zgemm or headline or fixedpoint_products" > $out/pytest_tri.txt 2>&1; tail -3 $out/pytest_tri.txt
./tools/gemm_time 1024 > $out/gemm_time_1024.txt 2>&1; grep -v "tri32\|full" $out/gemm_time_1024.txt
./tools/gemm_time 2048 50 > $out/gemm_time_2048.txt 2>&1; grep -v "tri32\|full" $out/gemm_time_2048.txt
QF_FUSED=1 ./tools/tri_probe_light 1024 > $out/tri_probe_light_1024.txt 2>&1; sed -n 9,18p $out/tri_probe_light_1024.txt
export QUFLOW_HIP_SK_EPI_UNITS=2
bash tools/gpu/r4_ab.sh $out 1024
bash tools/gpu/r4_ab.sh $out 2048 --steps 60 --warmup 6
