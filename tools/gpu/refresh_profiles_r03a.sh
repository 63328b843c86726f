#!/bin/bash
# Round 3 evidence, part A: the GPU suite, smoke, the bench lines (driver form, default, other workloads, c64).
# Usage: gpurun --timeout 1200 -- bash tools/gpu/refresh_profiles_r03a.sh ; then copy from gpurun_out/refresh3/ (profiles/README_r03.md)
out=gpurun_out/refresh3; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1 || { echo "pytest failed"; tail -40 $out/pytest_gpu.txt; exit 1; }
tail -2 $out/pytest_gpu.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/bench_driver_form_k20.json 2> $out/bench_driver_form_k20.err || { echo "bench failed"; tail -20 $out/bench_driver_form_k20.err; exit 1; }
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/bench_driver_form_k20_b.json 2>> $out/bench_driver_form_k20.err
timeout -k 10 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err || { echo "bench failed"; exit 1; }
for n in 512 2048; do timeout -k 10 300 python bench.py --N $n --steps $([ $n = 512 ] && echo 400 || echo 60) --warmup 10 --no-config3 --cpu-seconds 8 >> $out/bench_lines.jsonl 2>> $out/bench_lines.err; done
timeout -k 10 300 python bench.py --N 256 --steps 400 --warmup 10 --no-config3 --cpu-seconds 0 >> $out/bench_lines.jsonl 2>> $out/bench_lines.err
timeout -k 10 300 python bench.py --ic B --no-config3 --cpu-seconds 0 >> $out/bench_lines.jsonl 2>> $out/bench_lines.err
timeout -k 10 300 python bench.py --fixed-iters 4 --no-config3 --cpu-seconds 0 >> $out/bench_lines.jsonl 2>> $out/bench_lines.err
timeout -k 10 300 python bench.py --compsum --no-config3 --cpu-seconds 0 >> $out/bench_lines.jsonl 2>> $out/bench_lines.err
for p in i8x6 i8 i8h; do timeout -k 10 300 python bench.py --products $p --no-config3 --cpu-seconds 0 >> $out/bench_lines.jsonl 2>> $out/bench_lines.err; done
for n in 512 1024 2048; do timeout -k 10 300 python bench.py --dtype c64 --N $n --steps $([ $n = 2048 ] && echo 60 || echo 200) --warmup 10 --cpu-seconds 6 >> $out/bench_c64.jsonl 2>> $out/bench_c64.err; done
python - <<'PY'
import json
for f in ("bench_lines", "bench_c64"):
    for l in open("gpurun_out/refresh3/%s.jsonl" % f):
        d = json.loads(l); c = d["config"]; r = d.get("roofline") or {}
        print("N=%d %s ic=%s products=%s fixed=%s compsum=%s: %.1f steps/s  its %.3f  gemm1 %.1f us frac %.3f  gemm2 %s solve %s" % (
            c["N"], c.get("state_dtype"), c["ic"], c["products"], c["fixed_iters"], c["compsum"], d["value"], c["iterations_per_step"],
            r.get("avg_launch_us", 0), r.get("frac", 0), (r.get("second_product") or {}).get("avg_launch_us"), (r.get("laplacian_inverse") or {}).get("avg_launch_us")))
for f in ("bench_driver_form_k20", "bench_driver_form_k20_b", "bench_default"):
    d = json.loads(open("gpurun_out/refresh3/%s.json" % f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, d["value"], "no prewarm", d["config"].get("value_without_prewarm"), "gemm1", r["avg_launch_us"], "frac", r["frac"], "alg", r["algorithmic_frac"],
          "whole", r.get("whole_step", {}).get("frac"), "N512", (d.get("other_sizes") or {}).get("N512", {}).get("value"),
          (d.get("other_sizes") or {}).get("N512", {}).get("whole_step_frac"))
PY
