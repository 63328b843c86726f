#!/bin/bash
# round 5, first contact: config 3's trace fix (fp64 diagonal of the first int8 product), fault injection in k_oz_gemm
out=gpurun_out/r05_a
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_hip_faults.py -q -k "i8 or config3 or config5 or int8 or full_size or fault or hybrid" > $out/pytest_i8.txt 2>&1
rc=$?
tail -15 $out/pytest_i8.txt
echo pytest rc=$rc
QUFLOW_HIP_GEMM=i8x65 timeout -k 10 300 python tools/longrun.py 2048 10000 1000 > $out/longrun_n2048_10k_steps_i8x65.json 2> $out/longrun_i8x65.err || exit 1
python - <<'PY'
import json
for f in ("gpurun_out/r05_a/longrun_n2048_10k_steps_i8x65.json",):
    r = json.load(open(f)); print(f, r["trace"], r["timesteps_per_s"], r["casimir_drift_k234"])
PY
timeout -k 10 300 python bench.py --N 1024 --steps 200 --warmup 20 --products i8x65 --cpu-seconds 0 --no-side-runs > $out/bench_i8x65_n1024.json 2> $out/bench_i8x65_n1024.err || exit 1
python -c "import json;r=json.load(open('$out/bench_i8x65_n1024.json'));print('i8x65 N=1024', r['value'])"
timeout -k 10 300 python bench.py --N 1024 --steps 200 --warmup 20 --cpu-seconds 0 --no-side-runs --no-config3 > $out/bench_f64_n1024.json 2> $out/bench_f64_n1024.err || exit 1
python -c "import json;r=json.load(open('$out/bench_f64_n1024.json'));print('f64 N=1024', r['value'])"
