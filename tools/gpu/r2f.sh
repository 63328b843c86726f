#!/bin/bash
out=gpurun_out/r2f; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -k "ensemble or n64_golden or chunking or contract or hooks" > $out/pytest.txt 2>&1 || { echo "pytest failed"; tail -40 $out/pytest.txt; exit 1; }
tail -2 $out/pytest.txt
timeout -k 10 300 python tools/ensemble_rate.py 512 1,2,3,4,6 400 | tee $out/ens512.jsonl
timeout -k 10 300 python tools/ensemble_rate.py 1024 1,2,4 200 | tee $out/ens1024.jsonl
timeout -k 10 300 python tools/ensemble_rate.py 256 1,2,4,8 400 | tee $out/ens256.jsonl
