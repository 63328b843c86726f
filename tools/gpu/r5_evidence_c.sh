#!/bin/bash
# round-5 evidence, part C: bench lines of the variants (sizes, IC-B, fixed iterations, compsum, config-3 products, complex64),
# soak runs of the round's kernels, replicas per GPU
out=gpurun_out/r05_evc; mkdir -p $out
rm -f $out/bench_lines.jsonl
b() { timeout -k 10 300 python bench.py "$@" >> $out/bench_lines.jsonl 2>> $out/bench_lines.err; }
b --N 512 --steps 400 --warmup 20 --no-config3 --cpu-seconds 4
b --N 2048 --steps 60 --warmup 6 --no-config3 --cpu-seconds 4
b --N 256 --steps 400 --warmup 20 --no-config3 --cpu-seconds 0
b --ic B --no-config3 --no-side-runs --cpu-seconds 0
b --fixed-iters 4 --no-config3 --no-side-runs --cpu-seconds 0
b --fixed-iters 10 --no-config3 --no-side-runs --cpu-seconds 0
b --compsum --no-config3 --no-side-runs --cpu-seconds 0
for p in i8x65 i8x6 i8x6f i8; do b --products $p --no-side-runs --cpu-seconds 0; done
b --N 2048 --steps 60 --warmup 6 --products i8x65 --no-side-runs --cpu-seconds 0
b --dtype c64 --N 1024 --steps 200 --warmup 10 --cpu-seconds 0
b --dtype c64 --N 512 --steps 400 --warmup 20 --cpu-seconds 0
python - <<'PY'
import json
for l in open("gpurun_out/r05_evc/bench_lines.jsonl"):
    d=json.loads(l); c=d["config"]; r=d.get("roofline") or {}
    print("N=%s %s prod=%s ic=%s: %.1f steps/s (no prewarm %s) its %.3f g1 %s" % (c.get("N"), d["dtype"][:8], c.get("products"), c.get("initial_condition", c.get("ic")), d["value"], c.get("value_without_prewarm"), c.get("iterations_per_step",0), r.get("avg_launch_us")))
PY
timeout -k 10 200 python tools/longrun.py 1024 50000 5000 > $out/longrun_n1024_50k_steps.json 2> $out/longrun_n1024_50k_steps.progress
timeout -k 10 200 python tools/longrun.py 512 100000 10000 > $out/longrun_n512_100k_steps.json 2> $out/longrun_n512_100k_steps.progress
python -c "
import json
for f in ('longrun_n1024_50k_steps','longrun_n512_100k_steps'):
    d=json.load(open('$out/'+f+'.json')); print(f, d['timesteps_per_s'], d['casimir_drift_k234'], d['skew_hermitian_defect'], d['trace'])"
timeout -k 10 200 python tools/ensemble_rate.py 512 1,2,4 300 > $out/ensemble_on_one_gpu.jsonl 2>&1
timeout -k 10 200 python tools/ensemble_rate.py 1024 1,2 150 >> $out/ensemble_on_one_gpu.jsonl 2>&1
cut -c1-300 $out/ensemble_on_one_gpu.jsonl
