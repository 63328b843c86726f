#!/bin/bash
out=gpurun_out/r2h; mkdir -p $out
for i in 1 2 3; do QF_FUSED=1 timeout -k 10 60 tools/tri_probe_light 1024 > /dev/null 2>&1; done
QF_FUSED=1 timeout -k 10 60 tools/tri_probe_light 1024 | tee $out/tri_light_fused.txt
timeout -k 10 60 tools/tri_probe_light 1024 | grep -v "^full" | tee $out/tri_light_nonfused.txt
