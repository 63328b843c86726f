#!/bin/bash
out=gpurun_out/r04c; mkdir -p $out
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "config3 or int8 or i8" > $out/pytest_i8.txt 2>&1; tail -3 $out/pytest_i8.txt
for p in f64 i8x6 i8x6f i8x65; do
  timeout -k 10 120 python bench.py --products $p --no-side-runs --cpu-seconds 0 > $out/bench_${p}_1024.json 2> $out/bench_${p}_1024.err
done
for p in f64 i8x6 i8x65; do
  timeout -k 10 200 python bench.py --N 2048 --steps 60 --warmup 6 --products $p --no-side-runs --cpu-seconds 0 > $out/bench_${p}_2048.json 2> $out/bench_${p}_2048.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04c/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get("roofline") or {}
        print(f.split('/')[-1], "%.1f"%d["value"], d["config"].get("iterations_per_step"), r.get("avg_launch_us"), (r.get("second_product") or {}).get("avg_launch_us"), (r.get("laplacian_inverse") or {}).get("avg_launch_us"))
    except Exception as e: print(f, "ERR", e)
PY
