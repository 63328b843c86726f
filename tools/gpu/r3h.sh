#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3h; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -k "fixedpoint_products or n64_golden or chunking or spot or large or ensemble or contract or i8 or hooks" > $out/pytest.txt 2>&1 || { echo "pytest failed"; tail -40 $out/pytest.txt; exit 1; }
tail -2 $out/pytest.txt
for E in 4 2 6 0 8; do
QUFLOW_HIP_SK_EPI_UNITS=$E timeout -k 10 200 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('E=$E K=200', d['value'], d['roofline']['avg_launch_us'])"
done
QUFLOW_HIP_SK_EPI_UNITS=4 timeout -k 10 200 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-config3 --no-side-runs | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('E=4 K=200', d['value'], d['roofline']['avg_launch_us'])"
tools/kstats.sh $out/kstats --steps 200 --warmup 20 > $out/kstats_summary.txt 2>&1; python3 tools/trace_summary.py $out/kstats > $out/trace_summary.txt 2>&1; head -5 $out/trace_summary.txt
