#!/bin/bash
out=gpurun_out/r2k; mkdir -p $out
export TMPDIR=/tmp
python -c "import bench; print('visible_gpu_count', bench.visible_gpu_count())"
python bench.py --gpus 2 --steps 2 --warmup 0 > $out/gpus2.out 2> $out/gpus2.err; echo "rc for --gpus 2 on a 1-GPU box: $?"; cat $out/gpus2.err | tail -2
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/bench_driver_form_k20.json 2> $out/bench_driver_form_k20.err || { echo "bench failed"; tail -20 $out/bench_driver_form_k20.err; exit 1; }
timeout -k 10 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err || { echo "bench failed"; tail -20 $out/bench_default.err; exit 1; }
python - <<'PY'
import json
for f in ("bench_driver_form_k20", "bench_default"):
    d = json.loads(open("gpurun_out/r2k/%s.json" % f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, "value %.1f" % d["value"], "gemm1 %.1f us" % r["avg_launch_us"], "exec_frac %.3f" % r["executed_frac"], "timed", r["launches_timed_with_events"],
          "tri %.1f (%.3f)" % (r["second_product"]["avg_launch_us"], r["second_product"]["executed_frac"]), "solve %.1f (%.3f)" % (r["laplacian_inverse"]["avg_launch_us"], r["laplacian_inverse"]["frac"]),
          "whole %.3f" % r["whole_step"]["frac"], "fixed10 %.1f" % r["fixed_iterations_10"]["value"])
    print("  config3", {k: d["config3_lowprecision_products"][k] for k in ("value", "casimir_drift", "casimir_drift_f64_run", "spectrum_drift", "spectrum_drift_f64_run", "max_abs_state_diff_vs_f64_run")})
    print("  other", {k: (v["value"], v["roofline_executed_frac_first_product"]) for k, v in d["other_sizes"].items()})
    print("  replicas", {k: (v["sum_timesteps_per_s"], v["ratio"]) for k, v in d["replicas_per_gpu"].items()})
    print("  cpu", d["cpu_baseline"]["value"])
PY
tools/kstats.sh $out/kstats > $out/kstats_summary.txt 2>&1; python3 tools/trace_summary.py $out/kstats > $out/trace_summary.txt 2>&1; head -8 $out/trace_summary.txt
