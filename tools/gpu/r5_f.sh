#!/bin/bash
out=gpurun_out/r05_f
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_faults.py -x -q -k "i8 or config3 or int8 or fault or hybrid or plan" > $out/pytest_i8.txt 2>&1; rc=$?; tail -8 $out/pytest_i8.txt; echo "rc=$rc"
for rep in 1 2; do for lib in base new; do
  L=$PWD/quflow_amd/libquflow_hip.so; [ $lib = base ] && L=$PWD/tools/ab/libquflow_hip_base.so
  QUFLOW_HIP_LIB=$L timeout -k 10 200 python bench.py --N 1024 --steps 300 --warmup 20 --products i8x65 --cpu-seconds 0 --no-side-runs --kernel-table > $out/${lib}_$rep.json 2>$out/${lib}_$rep.err
  python -c "import json;d=json.load(open('$out/${lib}_$rep.json'));print('$lib rep $rep i8x65 N=1024', round(d['value'],1), 'timesteps/s')"; grep "kernel-table" $out/${lib}_$rep.err | grep -v " 0  total"
done; done
