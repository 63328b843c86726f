#!/bin/bash
# quick A/B: selected parity tests, kernel trace summary, bench at K = 200 (x2) and K = 20
export TMPDIR=/tmp
out=gpurun_out/ab; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -k "${QF_AB_TESTS:-fixedpoint_products or n64_golden or chunking or spot or large or ensemble or contract or zgemm or hooks}" > $out/pytest.txt 2>&1 || { echo "pytest failed"; tail -40 $out/pytest.txt; exit 1; }
tail -2 $out/pytest.txt
tools/kstats.sh $out/kstats --steps 200 --warmup 20 > $out/kstats_summary.txt 2>&1; python3 tools/trace_summary.py $out/kstats > $out/trace_summary.txt 2>&1; head -5 $out/trace_summary.txt
for i in 1 2; do timeout -k 10 200 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=200', d['value'], d['roofline']['avg_launch_us'])"; done
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=20', d['value'], d['roofline']['avg_launch_us'])"
