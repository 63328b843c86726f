#!/bin/bash
# round 6: first operand / sweep loads ahead of the tag look-up in k_zgemm (B operand), k_zgemm_tri32 and k_solve -- tests, then
# same-box A/B of four builds: base (HEAD before), products_only, solve_only, new (both)
export TMPDIR=/tmp
out=gpurun_out/r06_early; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_hip_faults.py tests/test_hip_envelope.py -x -q -m gpu -k "deferred or golden or large or chunking or fault or nonfinite or ensemble or plan or fold or protocols or variants or compsum or config5 or threads" > $out/pytest.txt 2>&1; rc=$?
tail -4 $out/pytest.txt; echo "tests rc=$rc"; [ $rc = 0 ] || exit $rc
for N in 512 1024 256 768 2048; do
  K=$(( N <= 512 ? 600 : N <= 1024 ? 300 : 60 ))
  for rep in 1 2 3; do for lib in base products_only solve_only new; do
    if [ $lib = new ]; then env="QUFLOW_DUMMY=1"; else env="QUFLOW_HIP_LIB=$PWD/tools/ab/libquflow_hip_$lib.so"; fi
    env $env timeout -k 10 200 python bench.py --N $N --steps $K --warmup 20 --cpu-seconds 0 --no-side-runs --no-config3 --no-kernel-events > $out/ab_${N}_${lib}_$rep.json 2>$out/ab_${N}_${lib}_$rep.err
    python -c "import json;d=json.load(open('$out/ab_${N}_${lib}_$rep.json'));print('N=$N %-14s rep $rep' % '$lib', round(d['value'],1), 'timesteps/s', round(1e3*d['ms_per_step']/max(d['config']['iterations_per_step'],1e-9),2), 'us/iteration')"
  done; done
done 2>&1 | tee $out/ab_summary.txt
for lib in base new; do
  if [ $lib = new ]; then env="QUFLOW_DUMMY=1"; else env="QUFLOW_HIP_LIB=$PWD/tools/ab/libquflow_hip_$lib.so"; fi
  env $env timeout -k 10 200 python tools/ensemble_rate.py 512 4 300 2>/dev/null | tail -1 | cut -c1-300
done | tee -a $out/ab_summary.txt
