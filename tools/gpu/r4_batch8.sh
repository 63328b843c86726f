#!/bin/bash
out=gpurun_out/r04i; mkdir -p $out
for v in "" _l2; do
./tools/gemm_time$v 1024 > $out/gemm_time${v}_1024.txt 2>&1; grep "triangle" $out/gemm_time${v}_1024.txt
QF_FUSED=1 ./tools/tri_probe_light$v 1024 > $out/tri_probe_light${v}_1024.txt 2>&1; sed -n 9,18p $out/tri_probe_light${v}_1024.txt
done
