#!/bin/bash
# round-5 evidence, part A: smoke, bench lines (default and the driver's form), kernel traces, probes
export TMPDIR=/tmp
out=gpurun_out/r05_ev; mkdir -p $out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for t in a b; do
  timeout -k 10 500 python bench.py > $out/bench_default_$t.json 2> $out/bench_default_$t.err || { echo "bench failed"; tail -20 $out/bench_default_$t.err; exit 1; }
done
for t in a b c; do
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/bench_driver_form_k20_$t.json 2> $out/bench_driver_form_k20_$t.err
done
python - <<'PY'
import json
for f in ("bench_default_a","bench_default_b","bench_driver_form_k20_a","bench_driver_form_k20_b","bench_driver_form_k20_c"):
    d=json.loads(open("gpurun_out/r05_ev/%s.json"%f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, "%.1f"%d["value"], "noprewarm", d["config"].get("value_without_prewarm"), "g1 %.1f frac %.3f g2 %.1f solve %.1f"%(r["avg_launch_us"], r["frac"], r["second_product"]["avg_launch_us"], r["laplacian_inverse"]["avg_launch_us"]),
          "cfg3 %.1f (x%.3f) N2048 x%s"%(d["config3_lowprecision_products"]["value"], d["config3_lowprecision_products"]["vs_fp64_headline"], (d["config3_lowprecision_products"].get("N2048") or {}).get("vs_fp64_same_size")),
          "N512", (d.get("other_sizes") or {}).get("N512",{}).get("value"), "N2048", (d.get("other_sizes") or {}).get("N2048",{}).get("value"))
PY
cd /tmp && cd $GRAFT_REPO_ROOT
trace() {  # name, bench args...
  n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$n -- python3 bench.py --cpu-seconds 0 --no-kernel-events --no-config3 --no-side-runs "$@" > $out/bench_${n}_under_rocprof.json 2> $out/bench_${n}_under_rocprof.err
  python3 tools/trace_summary.py $out/prof_$n > $out/bench_kernel_trace_summary_$n.txt 2>&1; head -7 $out/bench_kernel_trace_summary_$n.txt
}
trace n1024
cp $out/prof_n1024/*/*kernel_stats.csv $out/bench_kernel_stats.csv
python3 tools/iter_timeline.py $out/prof_n1024 > $out/iter_timeline_n1024.txt 2>&1 || true
trace k20 --steps 20 --warmup 5
trace n512 --N 512 --steps 400 --warmup 20
trace n2048 --N 2048 --steps 60 --warmup 6
trace i8x65 --products i8x65
trace c64 --dtype c64
rm -rf $out/prof_*
timeout -k 5 60 tools/solve_probe 512 > $out/solve_probe.txt 2>&1; timeout -k 5 60 tools/solve_probe 1024 >> $out/solve_probe.txt 2>&1; timeout -k 5 60 tools/solve_probe 2048 >> $out/solve_probe.txt 2>&1
QF_FUSED=1 timeout -k 5 60 tools/tri_probe_light 1024 > $out/tri_probe_light_n1024.txt 2>&1
timeout -k 5 120 tools/gemm_time 1024 > $out/gemm_time_n1024.txt 2>&1; timeout -k 5 120 tools/gemm_time 512 > $out/gemm_time_n512.txt 2>&1
ls $out | head -50
