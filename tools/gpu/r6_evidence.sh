#!/bin/bash
# round-6 evidence set (profiles/README_r06.md): smoke, the GPU suite (default products and under auto), bench lines (default x2, the
# driver's form x3), rocprofv3 kernel traces of the same commands, PMC passes (separate --pmc runs beside --kernel-trace only),
# config 5's long run, the k-replicas line.  Usage: gpurun --timeout 1200 -- bash tools/gpu/r6_evidence.sh [a|b|c|d|e]
export TMPDIR=/tmp
part=${1:-all}
out=gpurun_out/r06_ev; mkdir -p $out
cd /tmp && cd $GRAFT_REPO_ROOT
if [ $part = all ] || [ $part = a ]; then
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=8 > $out/pytest_gpu.txt 2>&1; rc=$?; tail -14 $out/pytest_gpu.txt; [ $rc = 0 ] || exit $rc
for t in a b; do
  timeout -k 10 500 python bench.py > $out/bench_default_$t.json 2> $out/bench_default_$t.err || { echo "bench failed"; tail -20 $out/bench_default_$t.err; exit 1; }
done
for t in a b c; do
  t0=$(date +%s.%N); timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_form_k20_$t.json 2> $out/bench_driver_form_k20_$t.err; python3 -c "import time; print(round(time.time() - $t0, 1))" > $out/driver_form_$t.wall
done
python - <<'PY'
import json
for f in ("bench_default_a","bench_default_b","bench_driver_form_k20_a","bench_driver_form_k20_b","bench_driver_form_k20_c"):
    d=json.loads(open("gpurun_out/r06_ev/%s.json"%f).read().strip().splitlines()[-1]); r=d["roofline"]
    o=d.get("other_sizes") or {}
    print(f, "%.1f"%d["value"], "noprewarm %.1f"%(d["config"].get("value_without_prewarm") or 0), "g1 %.1f frac %.3f g2 %.1f solve %.1f (%.2f)"%(r["avg_launch_us"], r["frac"], r["second_product"]["avg_launch_us"], r["laplacian_inverse"]["avg_launch_us"], r["laplacian_inverse"]["frac"]),
          "whole %.3f"%r["whole_step"]["frac"], "cfg3 x%.3f N2048 x%s"%(d["config3_lowprecision_products"]["vs_fp64_headline"], (d["config3_lowprecision_products"].get("N2048") or {}).get("vs_fp64_same_size")),
          "| N512 %.0f solve %.1fus %.2f whole %.2f"%(o["N512"]["value"], o["N512"]["laplacian_inverse"]["avg_launch_us"], o["N512"]["laplacian_inverse"]["frac"], o["N512"]["whole_step_frac"]),
          "| N2048 %.1f solve %.1fus %.2f whole %.2f"%(o["N2048"]["value"], o["N2048"]["laplacian_inverse"]["avg_launch_us"], o["N2048"]["laplacian_inverse"]["frac"], o["N2048"]["whole_step_frac"]),
          "| IC-B %.0f its %.2f oracle-equal %s"%(d["smooth_data"]["N1024"]["value"], d["smooth_data"]["N1024"]["iterations_per_step"], d["smooth_data"]["N1024"]["oracle_check"].get("equal")),
          "| x4 N512 %.0f"%d["replicas_per_gpu"]["N512_x4"]["sum_timesteps_per_s"])
PY
cat $out/driver_form_*.wall | tr '\n' ' '; echo "s wall (driver form)"
fi
if [ $part = all ] || [ $part = b ]; then
trace() {  # name, bench args...
  n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$n -- python3 bench.py --cpu-seconds 0 --no-kernel-events --no-config3 --no-side-runs "$@" > $out/bench_${n}_under_rocprof.json 2> $out/bench_${n}_under_rocprof.err
  python3 tools/trace_summary.py $out/prof_$n > $out/bench_kernel_trace_summary_$n.txt 2>&1; head -6 $out/bench_kernel_trace_summary_$n.txt
}
trace n1024
cp $out/prof_n1024/*/*kernel_stats.csv $out/bench_kernel_stats.csv 2>/dev/null
python3 tools/iter_timeline.py $out/prof_n1024 > $out/iter_timeline_n1024.txt 2>&1 || true
trace k20 --steps 20 --warmup 5
trace n512 --N 512 --steps 400 --warmup 20
python3 tools/iter_timeline.py $out/prof_n512 > $out/iter_timeline_n512.txt 2>&1 || true
trace n2048 --N 2048 --steps 60 --warmup 6
trace n4096 --N 4096 --steps 12 --warmup 2
trace i8x65 --products i8x65
rm -rf $out/prof_*
fi
if [ $part = all ] || [ $part = c ]; then
pm=gpurun_out/r06_pmc; mkdir -p $pm
bash tools/pmc_pass.sh $pm/f64 > $pm/f64_passes.txt 2>&1
python3 tools/pmc_summary.py $pm/f64 > $pm/pmc_summary.txt 2>&1; head -30 $pm/pmc_summary.txt
bash tools/pmc_pass.sh $pm/n512 --N 512 --steps 400 --warmup 20 > $pm/n512_passes.txt 2>&1
python3 tools/pmc_summary.py $pm/n512 > $pm/pmc_summary_n512.txt 2>&1
bash tools/pmc_pass.sh $pm/n2048 --N 2048 --steps 40 --warmup 4 > $pm/n2048_passes.txt 2>&1
python3 tools/pmc_summary.py $pm/n2048 > $pm/pmc_summary_n2048.txt 2>&1; head -30 $pm/pmc_summary_n2048.txt
bash tools/pmc_pass.sh $pm/i8x65 --products i8x65 > $pm/i8x65_passes.txt 2>&1
python3 tools/pmc_summary.py $pm/i8x65 > $pm/pmc_summary_i8x65.txt 2>&1
bash tools/pmc_pass.sh $pm/c64 --dtype c64 > $pm/c64_passes.txt 2>&1
python3 tools/pmc_summary.py $pm/c64 > $pm/pmc_summary_c64.txt 2>&1
find $pm -name "*.csv" -delete; find $pm -name "*.db" -delete; du -sh $pm
fi
if [ $part = all ] || [ $part = d ]; then
timeout -k 10 500 python tools/longrun.py 2048 10000 1000 > $out/longrun_n2048_10k_steps.json 2> $out/longrun_n2048_10k_steps.err; tail -c 900 $out/longrun_n2048_10k_steps.json; echo
timeout -k 10 300 python tools/longrun.py 512 100000 10000 > $out/longrun_n512_100k_steps.json 2> $out/longrun_n512_100k_steps.err; tail -c 600 $out/longrun_n512_100k_steps.json; echo
fi
if [ $part = all ] || [ $part = e ]; then
QUFLOW_HIP_GEMM=auto timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_zz_perf_guard.py > $out/pytest_gpu_under_auto_products.txt 2>&1; tail -3 $out/pytest_gpu_under_auto_products.txt
fi
