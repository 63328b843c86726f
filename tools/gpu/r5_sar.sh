#!/bin/bash
# experiment: the re+im operand of the 3M form made behind the fragment's LDS read (one v_add_f64 per fragment) instead of a third LDS plane
out=gpurun_out/r05_sar; mkdir -p $out
for r in 1 2; do
  timeout -k 5 120 tools/gemm_time 1024 2>&1 | grep -E "first product|triangle \+ fused" | sed "s/^/base /"
  timeout -k 5 120 tools/gemm_time_sar 1024 2>&1 | grep -E "first product|triangle \+ fused" | sed "s/^/sar  /"
done | tee $out/gemm_time_ab.txt
timeout -k 5 120 tools/gemm_time 2048 2>&1 | grep -E "first product|triangle \+ fused" | sed "s/^/base 2048 /" | tee -a $out/gemm_time_ab.txt
timeout -k 5 120 tools/gemm_time_sar 2048 2>&1 | grep -E "first product|triangle \+ fused" | sed "s/^/sar  2048 /" | tee -a $out/gemm_time_ab.txt
