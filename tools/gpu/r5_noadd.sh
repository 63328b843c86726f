#!/bin/bash
# ceiling of "the re+im planes come from the producers": the K loops without their v_add_f64 (timing only; results wrong)
out=gpurun_out/r05_noadd; mkdir -p $out
for r in 1 2; do
  timeout -k 5 120 tools/gemm_time 1024 2>&1 | grep -E "first product|triangle \+ fused" | sed "s/^/base  /"
  timeout -k 5 120 tools/gemm_time_noadd 1024 2>&1 | grep -E "first product|triangle \+ fused" | sed "s/^/noadd /"
done | tee $out/gemm_time_ab.txt
QF_FUSED=1 timeout -k 5 60 tools/tri_probe_light 1024 2>&1 | grep -E "K-tile|life" | sed "s/^/base  /" | tee $out/tri_probe_ab.txt
QF_FUSED=1 timeout -k 5 60 tools/tri_probe_light_noadd 1024 2>&1 | grep -E "K-tile|life" | sed "s/^/noadd /" | tee -a $out/tri_probe_ab.txt
timeout -k 5 120 tools/gemm_time 2048 2>&1 | grep -E "first product|triangle \+ fused" | sed "s/^/base  2048 /" | tee -a $out/gemm_time_ab.txt
timeout -k 5 120 tools/gemm_time_noadd 2048 2>&1 | grep -E "first product|triangle \+ fused" | sed "s/^/noadd 2048 /" | tee -a $out/gemm_time_ab.txt
