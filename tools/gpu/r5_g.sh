#!/bin/bash
out=gpurun_out/r05_g
mkdir -p $out
for lib in base new xp1 xp2 xp3 xp7; do
  L=$PWD/tools/ab/libquflow_hip_$lib.so; [ $lib = new ] && L=$PWD/quflow_amd/libquflow_hip.so
  QUFLOW_HIP_LIB=$L timeout -k 10 200 python bench.py --N 1024 --steps 300 --warmup 20 --products i8x65 --fixed-iters 4 --cpu-seconds 0 --no-side-runs --kernel-table > $out/$lib.json 2>$out/$lib.err
  echo "== $lib: $(python -c "import json;d=json.load(open('$out/$lib.json'));print(round(d['value'],1))") timesteps/s (4 iterations per step)"; grep "kernel-table" $out/$lib.err | grep -v " 0  total" | cut -c1-80
done
