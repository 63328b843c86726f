#!/bin/bash
# round 4, batch 1: new parity tests, bf16 probe, solve probe, config-3 variants
out=gpurun_out/r04b; mkdir -p $out
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "config3 or headline or foreign_hamiltonian or compsum_with_tri32 or vs_oracle_large" > $out/pytest_new.txt 2>&1; tail -3 $out/pytest_new.txt
./tools/bf16_split_probe 1024 > $out/bf16_split_probe_n1024.json 2>&1
./tools/bf16_split_probe 2048 > $out/bf16_split_probe_n2048.json 2>&1
for n in 512 1024 2048; do ./tools/solve_probe $n > $out/solve_probe_$n.txt 2>&1; done
for p in f64 i8x6 i8x6f i8hx6; do
  timeout -k 10 120 python bench.py --products $p --no-side-runs --cpu-seconds 0 > $out/bench_${p}_1024.json 2> $out/bench_${p}_1024.err
done
for p in f64 i8x6 i8x6f; do
  timeout -k 10 200 python bench.py --N 2048 --steps 60 --warmup 6 --products $p --no-side-runs --cpu-seconds 0 > $out/bench_${p}_2048.json 2> $out/bench_${p}_2048.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04b/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get("roofline") or {}
        print(f.split('/')[-1], "%.1f"%d["value"], d["config"].get("iterations_per_step"), r.get("avg_launch_us"), (r.get("second_product") or {}).get("avg_launch_us"), (r.get("laplacian_inverse") or {}).get("avg_launch_us"))
    except Exception as e: print(f, "ERR", e)
PY
