#!/bin/bash
out=gpurun_out/r2d; mkdir -p $out
export TMPDIR=/tmp
for v in spread1 spread2; do echo "== $v"; timeout -k 10 120 tools/gemm_time_$v 1024 | tee $out/gemm_time_$v.txt; done
echo "== solve probe (G default)"; timeout -k 10 60 tools/solve_probe 1024 | tee $out/solve_probe.txt
for g in 2 8 16; do echo "== solve probe G=$g"; QUFLOW_HIP_SOLVE_G=$g timeout -k 10 60 tools/solve_probe 1024 | head -3; done
echo "== solve probe N=512/2048"; timeout -k 10 60 tools/solve_probe 512 | head -2; timeout -k 10 60 tools/solve_probe 2048 | head -2
