#!/bin/bash
# round-4 evidence, part A: GPU suite, smoke, bench lines (default and the driver's form), kernel trace
export TMPDIR=/tmp
out=gpurun_out/r04_ev; mkdir -p $out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1 || { echo "pytest failed"; tail -40 $out/pytest_gpu.txt; exit 1; }
tail -2 $out/pytest_gpu.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err || { echo "bench failed"; tail -20 $out/bench_default.err; exit 1; }
for t in a b c; do
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/bench_driver_form_k20_$t.json 2> $out/bench_driver_form_k20_$t.err
done
python - <<'PY'
import json
for f in ("bench_default","bench_driver_form_k20_a","bench_driver_form_k20_b","bench_driver_form_k20_c"):
    d=json.loads(open("gpurun_out/r04_ev/%s.json"%f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, "%.1f"%d["value"], "noprewarm", d["config"].get("value_without_prewarm"), "g1 %.1f frac %.3f g2 %.1f solve %.1f"%(r["avg_launch_us"], r["frac"], r["second_product"]["avg_launch_us"], r["laplacian_inverse"]["avg_launch_us"]),
          "cfg3 %.1f (x%.3f)"%(d["config3_lowprecision_products"]["value"], d["config3_lowprecision_products"]["vs_fp64_headline"]),
          "N512", (d.get("other_sizes") or {}).get("N512",{}).get("value"), "N2048", (d.get("other_sizes") or {}).get("N2048",{}).get("value"),
          "x2", (d.get("replicas_per_gpu") or {}).get("N1024_x2",{}).get("ratio"))
PY
cd /tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_f64 -- python3 bench.py --cpu-seconds 0 --no-kernel-events --no-config3 --no-side-runs > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
python3 tools/trace_summary.py $out/prof_f64 > $out/bench_kernel_trace_summary.txt 2>&1; head -8 $out/bench_kernel_trace_summary.txt
cp $out/prof_f64/*/*kernel_stats.csv $out/bench_kernel_stats.csv
python3 tools/iter_timeline.py $out/prof_f64 > $out/iter_timeline_n1024.txt 2>&1 || true
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_k20 -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-kernel-events --no-config3 --no-side-runs > $out/bench_k20_under_rocprof.json 2> $out/bench_k20_under_rocprof.err
python3 tools/trace_summary.py $out/prof_k20 > $out/bench_k20_kernel_trace_summary.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_i8x65 -- python3 bench.py --cpu-seconds 0 --no-kernel-events --no-config3 --no-side-runs --products i8x65 > $out/bench_i8x65_under_rocprof.json 2> $out/bench_i8x65_under_rocprof.err
python3 tools/trace_summary.py $out/prof_i8x65 > $out/i8x65_kernel_trace_summary.txt 2>&1; head -7 $out/i8x65_kernel_trace_summary.txt
rm -rf $out/prof_f64 $out/prof_k20 $out/prof_i8x65
