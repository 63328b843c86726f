#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3c; mkdir -p $out
for n in 512 256; do
  for cfg in "" "QUFLOW_HIP_SOLVE_L=8" "QUFLOW_HIP_SOLVE_L=8 QUFLOW_HIP_SOLVE_G=2" "QUFLOW_HIP_SOLVE_L=8 QUFLOW_HIP_SOLVE_G=8" "QUFLOW_HIP_SOLVE_G=2" "QUFLOW_HIP_SOLVE_G=8"; do
    echo "== N=$n $cfg" >> $out/solve_probe.txt
    env $cfg timeout -k 10 60 tools/solve_probe $n >> $out/solve_probe.txt 2>&1
  done
done
grep -E "^==|full|no stores" $out/solve_probe.txt
for cfg in "" "QUFLOW_HIP_SOLVE_L=8"; do
  env $cfg timeout -k 10 200 python bench.py --N 512 --steps 400 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=512 K=400 $cfg', d['value'])"
done
o2=$out/trace512; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $o2 -- python3 bench.py --N 512 --steps 400 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs --no-kernel-events > $o2.json 2> $o2.err; python3 tools/iter_timeline.py $o2 4000 | tee $out/iter_timeline_512.txt
o2=$out/trace1024; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $o2 -- python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs --no-kernel-events > $o2.json 2> $o2.err; python3 tools/iter_timeline.py $o2 2000 | tee $out/iter_timeline_1024.txt
