#!/bin/bash
# round-4 evidence, part C: bench lines (sizes, variants), the long run of config 5, replicas per GPU, probes
out=gpurun_out/r04_evc; mkdir -p $out
rm -f $out/bench_lines.jsonl
b() { timeout -k 10 300 python bench.py "$@" >> $out/bench_lines.jsonl 2>> $out/bench_lines.err; }
b --N 512 --steps 400 --warmup 20 --no-config3 --cpu-seconds 4
b --N 2048 --steps 60 --warmup 6 --no-config3 --cpu-seconds 4
b --N 256 --steps 400 --warmup 20 --no-config3 --cpu-seconds 0
b --N 768 --steps 300 --warmup 20 --no-config3 --cpu-seconds 0
b --N 1536 --steps 100 --warmup 10 --no-config3 --cpu-seconds 0
b --ic B --no-config3 --no-side-runs --cpu-seconds 0
b --fixed-iters 4 --no-config3 --no-side-runs --cpu-seconds 0
b --compsum --no-config3 --no-side-runs --cpu-seconds 0
for p in i8x65 i8x6 i8x6f i8 i8hx6; do b --products $p --no-side-runs --cpu-seconds 0; done
for p in i8x65 i8x6; do b --N 2048 --steps 60 --warmup 6 --products $p --no-side-runs --cpu-seconds 0; done
b --dtype c64 --N 1024 --steps 200 --warmup 10 --cpu-seconds 0
b --dtype c64 --N 512 --steps 400 --warmup 20 --cpu-seconds 0
python - <<'PY'
import json
for l in open("gpurun_out/r04_evc/bench_lines.jsonl"):
    d=json.loads(l); c=d["config"]; r=d.get("roofline") or {}
    print("N=%s %s prod=%s ic=%s: %.1f steps/s its %.3f g1 %s" % (c.get("N"), d["dtype"][:8], c.get("products"), c.get("initial_condition", c.get("ic")), d["value"], c.get("iterations_per_step",0), r.get("avg_launch_us")))
PY
python tools/longrun.py 2048 10000 1000 > $out/longrun_n2048_10k_steps.json 2> $out/longrun_n2048_10k_steps.progress
python -c "import json; d=json.load(open('$out/longrun_n2048_10k_steps.json')); print('longrun', d['timesteps_per_s'], d['casimir_drift_k234'], d['skew_hermitian_defect'])"
python tools/ensemble_rate.py 512 1,2,4 300 > $out/ensemble_on_one_gpu.jsonl 2>&1
python tools/ensemble_rate.py 1024 1,2,4 150 >> $out/ensemble_on_one_gpu.jsonl 2>&1
cat $out/ensemble_on_one_gpu.jsonl | cut -c1-300
./tools/gemm_time 1024 > $out/gemm_time_n1024.txt 2>&1
./tools/gemm_time 512 > $out/gemm_time_n512.txt 2>&1
QF_FUSED=1 ./tools/tri_probe_light 1024 > $out/tri_probe_light_n1024.txt 2>&1
for n in 512 1024 2048; do ./tools/solve_probe $n; done > $out/solve_probe.txt 2>&1
./tools/bf16_split_probe 1024 > $out/bf16_split_probe_n1024.json 2>&1
./tools/bf16_split_probe 2048 > $out/bf16_split_probe_n2048.json 2>&1
