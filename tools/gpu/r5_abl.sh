#!/bin/bash
# where the K-tile's 70.4 cycles per MFMA (64 of pure issue) go: the K loop with parts switched off (timing only; results wrong)
out=gpurun_out/r05_abl; mkdir -p $out
for v in "" _NOSTORE _NOGLOAD _NOBARRIER _NOSTAGE _MFMAONLY; do
  timeout -k 5 120 tools/gemm_time$v 1024 2>&1 | grep -E "first product" | sed "s/^/base$v /"
done | tee $out/first_product_ablations.txt
