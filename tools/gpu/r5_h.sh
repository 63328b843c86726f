#!/bin/bash
out=gpurun_out/r05_h
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_faults.py -x -q -k "i8 or config3 or int8 or fault or hybrid or plan or config5" > $out/pytest_i8.txt 2>&1; rc=$?; tail -6 $out/pytest_i8.txt; echo "rc=$rc"
for lib in base new; do
  L=$PWD/tools/ab/libquflow_hip_$lib.so; [ $lib = new ] && L=$PWD/quflow_amd/libquflow_hip.so
  QUFLOW_HIP_LIB=$L timeout -k 10 200 python bench.py --N 1024 --steps 300 --warmup 20 --products i8x65 --fixed-iters 4 --cpu-seconds 0 --no-side-runs --kernel-table > $out/$lib.json 2>$out/$lib.err
  echo "== $lib: $(python -c "import json;d=json.load(open('$out/$lib.json'));print(round(d['value'],1))") timesteps/s (4 iterations per step)"; grep "kernel-table" $out/$lib.err | grep -v " 0  total" | cut -c1-80
done
for rep in 1 2; do for lib in base new; do
  L=$PWD/tools/ab/libquflow_hip_$lib.so; [ $lib = new ] && L=$PWD/quflow_amd/libquflow_hip.so
  for N in 1024 2048; do st=300; [ $N = 2048 ] && st=60
  QUFLOW_HIP_LIB=$L timeout -k 10 200 python bench.py --N $N --steps $st --warmup 20 --products i8x65 --cpu-seconds 0 --no-side-runs --no-kernel-events > $out/${lib}_${N}_$rep.json 2>/dev/null
  echo "$lib N=$N rep $rep: $(python -c "import json;d=json.load(open('$out/${lib}_${N}_$rep.json'));print(round(d['value'],1))") timesteps/s"
  done
done; done
