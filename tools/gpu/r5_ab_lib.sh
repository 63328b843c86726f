#!/bin/bash
# A/B of two builds of the library on one box: tools/ab/libquflow_hip_base.so against tools/ab/libquflow_hip_xp.so
# usage: r5_ab_lib.sh <tag> [bench args...]   (default: N = 1024, fixed 4 iterations, 300 steps)
tag=$1; shift
out=gpurun_out/r05_ab_$tag
mkdir -p $out
args="$@"
[ -z "$args" ] && args="--N 1024 --steps 300 --warmup 20 --fixed-iters 4"
for rep in 1 2 3; do for lib in base xp; do
  QUFLOW_HIP_LIB=$PWD/tools/ab/libquflow_hip_$lib.so timeout -k 10 200 python bench.py $args --cpu-seconds 0 --no-side-runs --no-config3 --no-kernel-events > $out/${lib}_$rep.json 2>$out/${lib}_$rep.err
  python -c "import json;d=json.load(open('$out/${lib}_$rep.json'));print('$lib rep $rep', round(d['value'],1), 'timesteps/s', round(1e3*d['ms_per_step']/max(d['config']['iterations_per_step'],1e-9),2), 'us/iteration')"
done; done 2>&1 | tee $out/summary.txt
