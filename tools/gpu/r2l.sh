#!/bin/bash
out=gpurun_out/r2l; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -k "n64_golden or chunking or contract or options or ensemble or spot or large" > $out/pytest.txt 2>&1 || { echo "pytest failed"; tail -40 $out/pytest.txt; exit 1; }
tail -2 $out/pytest.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/bench_driver_form_k20.json 2> $out/bench_driver_form_k20.err || { echo "bench failed"; tail -20 $out/bench_driver_form_k20.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2l/bench_driver_form_k20.json").read().strip().splitlines()[-1])
print("K=20 value %.1f" % d["value"], {k: (round(v["sum_timesteps_per_s"]), round(v["ratio"], 3)) for k, v in d["replicas_per_gpu"].items()})
PY
tools/kstats.sh $out/kstats > $out/kstats_summary.txt 2>&1; python3 tools/trace_summary.py $out/kstats > $out/trace_summary.txt 2>&1; head -7 $out/trace_summary.txt
