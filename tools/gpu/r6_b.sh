#!/bin/bash
# round 6, second GPU call: the split api_*.hip library under a quick test subset; the k_solve L = 4 experiment;
# the bench line in the driver's form and the default form with the new entries; the size sweep up to the context limit
export TMPDIR=/tmp
out=gpurun_out/r06_b; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_faults.py tests/test_hip_parity.py tests/test_hip_envelope.py -x -q -m gpu -k "fault or golden or nonfinite or plan or chunking or contract or passive or stack or states or hooks or magmp" > $out/pytest_subset.txt 2>&1; rc=$?
tail -4 $out/pytest_subset.txt; echo "subset rc=$rc"; [ $rc = 0 ] || exit $rc

# ---- k_solve with 4-step chunks (128 chunks per walk, two per scanning lane) at N <= 512
L4=$PWD/tools/ab/libquflow_hip_l4.so
QUFLOW_HIP_LIB=$L4 QUFLOW_HIP_SOLVE_L4=4 timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "poisson or solve or laplac or deferred or n64_golden" > $out/pytest_l4.txt 2>&1; rc=$?
tail -3 $out/pytest_l4.txt; echo "L4 tests rc=$rc"
if [ $rc = 0 ]; then
  for N in 512 256; do for rep in 1 2 3; do for mode in base 4 2; do
    if [ $mode = base ]; then env="QUFLOW_HIP_LIB=$L4"; else env="QUFLOW_HIP_LIB=$L4 QUFLOW_HIP_SOLVE_L4=$mode"; fi
    env $env timeout -k 10 200 python bench.py --N $N --steps 600 --warmup 20 --cpu-seconds 0 --no-side-runs --no-config3 --no-kernel-events > $out/l4_${N}_${mode}_$rep.json 2>$out/l4_${N}_${mode}_$rep.err
    python -c "import json;d=json.load(open('$out/l4_${N}_${mode}_$rep.json'));print('N=$N k_solve chunks: $mode rep $rep', round(d['value'],1), 'timesteps/s', round(1e3*d['ms_per_step']/max(d['config']['iterations_per_step'],1e-9),2), 'us/iteration')"
  done; done; done 2>&1 | tee $out/l4_summary.txt
fi

# ---- the bench line: driver's form (x3), default form
for rep in a b c; do
  /usr/bin/time -f "%e s wall" -o $out/driver_form_$rep.time timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_form_k20_$rep.json 2> $out/bench_driver_form_k20_$rep.err
  python -c "import json;d=json.load(open('$out/bench_driver_form_k20_$rep.json'));print('driver form $rep:', round(d['value'],1), 'timesteps/s; N512', round(d['other_sizes']['N512']['value'],1), 'N2048', round(d['other_sizes']['N2048']['value'],1), 'IC-B', round(d['smooth_data']['N1024']['value'],1), d['smooth_data']['N1024']['oracle_check'])"; cat $out/driver_form_$rep.time
done
timeout -k 10 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "default bench rc=$?"
python -c "import json;d=json.load(open('$out/bench_default.json'));print('default:', round(d['value'],1)); print(json.dumps(d['per_call'])[:1500]); print(json.dumps(d['other_sizes']['N512']['laplacian_inverse'])); print(json.dumps(d['other_sizes']['N2048']['laplacian_inverse']))"

# ---- size sweep up to the context limit
rm -f $out/size_sweep.jsonl
for N in 256 512 768 1024 1536 2048 3072 4096 6144 8192; do
  K=$(( N <= 512 ? 400 : N <= 1024 ? 200 : N <= 2048 ? 60 : N <= 3072 ? 30 : N <= 4096 ? 20 : N <= 6144 ? 10 : 6 ))
  for prod in f64 i8x65; do
    [ $prod = i8x65 ] && { [ $N -lt 1024 ] || [ $N -gt 4096 ]; } && continue
    timeout -k 10 400 python bench.py --N $N --steps $K --warmup 3 --cpu-seconds 0 --no-config3 --products $prod 2>> $out/sweep.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline'] or {}
sp=r.get('second_product') or {}; li=r.get('laplacian_inverse') or {}; ws=r.get('whole_step') or {}
print(json.dumps({'N': $N, 'products': '$prod', 'timesteps_per_s': round(d['value'],2), 'ms_per_step': round(d['ms_per_step'],4), 'iterations_per_step': d['config']['iterations_per_step'],
  'first_product': {'kernel': (r.get('launched',{}).get('first_product') or {}).get('kernel'), 'us': round(r.get('avg_launch_us',0),1), 'frac': round(r.get('frac',0),3)},
  'second_product': {'kernel': sp.get('kernel'), 'us': round(sp.get('avg_launch_us',0),1), 'frac': round(sp.get('frac',0),3)},
  'laplacian_inverse': {'kernel': li.get('kernel'), 'us': round(li.get('avg_launch_us',0),1), 'achieved_GBs': round(li.get('achieved_GBs',0),1), 'frac': round(li.get('frac',0),3)},
  'whole_step_frac': round(ws.get('frac',0),3) if ws else None}))" | tee -a $out/size_sweep.jsonl
  done
done
