#!/bin/bash
out=gpurun_out/r05_e
mkdir -p $out
for rep in 1 2; do for lib in base new; do
  L=$PWD/quflow_amd/libquflow_hip.so; [ $lib = base ] && L=$PWD/tools/ab/libquflow_hip_base.so
  QUFLOW_HIP_LIB=$L timeout -k 10 200 python bench.py --N 1024 --steps 200 --warmup 20 --cpu-seconds 0 --no-side-runs --no-config3 --kernel-table > $out/kt_${lib}_$rep.json 2>$out/kt_${lib}_$rep.err
  echo "== $lib rep $rep"; grep kernel-table $out/kt_${lib}_$rep.err
done; done
QF_FUSED=1 timeout -k 5 60 tools/tri_probe_light 1024 > $out/tri_probe_new.txt 2>&1; grep -v "^tri\|^block" $out/tri_probe_new.txt
