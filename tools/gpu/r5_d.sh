#!/bin/bash
# (record of how the shared-epilogue attempt was measured, profiles/r05_shared_epilogue_attempt.txt; QUFLOW_HIP_SK_EPI_UNITS existed at that
# commit and was removed later in the round -- the E sweep below no longer has a switch to drive)
out=gpurun_out/r05_d
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_faults.py -x -q -k "fixedpoint or golden or oracle_large or fault or headline or ensemble or full_size or config5 or stepper or protocols or plan" > $out/pytest_tri.txt 2>&1; rc=$?; tail -12 $out/pytest_tri.txt; echo "rc=$rc"
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do for lib in base new; do
  L=$PWD/quflow_amd/libquflow_hip.so; [ $lib = base ] && L=$PWD/tools/ab/libquflow_hip_base.so
  for E in 2; do
  QUFLOW_HIP_LIB=$L timeout -k 10 200 python bench.py --N 1024 --steps 300 --warmup 20 --cpu-seconds 0 --no-side-runs --no-config3 --no-kernel-events > $out/${lib}_$rep.json 2>$out/${lib}_$rep.err
  python -c "import json;d=json.load(open('$out/${lib}_$rep.json'));print('$lib rep $rep', round(d['value'],1), 'timesteps/s')"
  done
done; done
for E in 0 1 2 3; do
  QUFLOW_HIP_SK_EPI_UNITS=$E timeout -k 10 200 python bench.py --N 1024 --steps 300 --warmup 20 --cpu-seconds 0 --no-side-runs --no-config3 --no-kernel-events > $out/new_E$E.json 2>$out/new_E$E.err
  python -c "import json;d=json.load(open('$out/new_E$E.json'));print('new E=$E', round(d['value'],1), 'timesteps/s')"
done
