#!/bin/bash
# round 6, third GPU call: the N = 6144 iteration count against the oracle; the changed areas under test (qf_commutator, 4-entry
# chunks of k_solve at N <= 256 in both precisions, whole-stack finite check); A/B of the 4-entry chunks at N = 128 / 256 / 192
export TMPDIR=/tmp
out=gpurun_out/r06_c; mkdir -p $out
timeout -k 10 400 python tools/gpu/r6_n6144.py 6144 > $out/n6144_vs_oracle.txt 2>&1; cat $out/n6144_vs_oracle.txt | tail -3
timeout -k 10 900 python -m pytest tests/test_hip_envelope.py tests/test_hip_single.py tests/test_hip_parity.py -x -q -m gpu -k "not config5 and not fuzz and not randomised" > $out/pytest_changed.txt 2>&1; rc=$?
tail -5 $out/pytest_changed.txt; echo "tests rc=$rc"; [ $rc = 0 ] || exit $rc
BASE=$PWD/tools/ab/libquflow_hip_base.so
for N in 256 128 192; do for rep in 1 2 3; do for lib in base new; do
  if [ $lib = base ]; then env="QUFLOW_HIP_LIB=$BASE"; else env="QUFLOW_DUMMY=1"; fi
  env $env timeout -k 10 200 python bench.py --N $N --steps 800 --warmup 20 --cpu-seconds 0 --no-side-runs --no-config3 --no-kernel-events > $out/ab_${N}_${lib}_$rep.json 2>$out/ab_${N}_${lib}_$rep.err
  python -c "import json;d=json.load(open('$out/ab_${N}_${lib}_$rep.json'));print('N=$N $lib rep $rep', round(d['value'],1), 'timesteps/s', round(1e3*d['ms_per_step']/max(d['config']['iterations_per_step'],1e-9),2), 'us/iteration', d['roofline'] and d['roofline']['launched']['laplacian_inverse']['kernel'] if d.get('roofline') else '')"
done; done; done 2>&1 | tee $out/l4_ab_small_sizes.txt
for dt in c64; do for N in 256 128; do for lib in base new; do
  if [ $lib = base ]; then env="QUFLOW_HIP_LIB=$BASE"; else env="QUFLOW_DUMMY=1"; fi
  env $env timeout -k 10 200 python bench.py --dtype c64 --N $N --steps 800 --warmup 20 --cpu-seconds 0 --no-side-runs --no-config3 --no-kernel-events > $out/abc64_${N}_${lib}.json 2>$out/abc64_${N}_${lib}.err
  python -c "import json;d=json.load(open('$out/abc64_${N}_${lib}.json'));print('c64 N=$N $lib', round(d['value'],1), 'timesteps/s')"
done; done; done 2>&1 | tee -a $out/l4_ab_small_sizes.txt
t0=$(date +%s.%N); timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_form_k20.json 2> $out/bench_driver_form_k20.err; python3 -c "import time; print('driver form wall s:', round(time.time() - $t0, 1))"
python -c "import json;d=json.load(open('$out/bench_driver_form_k20.json'));print('driver form:', round(d['value'],1), 'noprewarm', round(d['config']['value_without_prewarm'],1)); print({N:{k:(round(1e3*v['seconds_per_call'],3), round(1e3*v['oracle_seconds_per_call'],3)) for k,v in d['per_call'][N].items()} for N in ('N512','N1024','N2048')})"
