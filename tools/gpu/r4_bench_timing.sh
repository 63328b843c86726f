T0=$(date +%s.%N); timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_timing.json 2> gpurun_out/r04_bench_timing.err; T1=$(date +%s.%N); python -c "print('bench wall s', $T1 - $T0)"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_bench_timing.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["other_sizes"]["N512"]["cpu_baseline"]["value"], d["other_sizes"]["N2048"]["cpu_baseline"]["value"])
PY
