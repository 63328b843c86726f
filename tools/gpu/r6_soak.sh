#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r06_soak; mkdir -p $out
timeout -k 10 600 python tools/gpu/r6_soak_threads.py 400 > $out/soak_threads.json 2> $out/soak_threads.err; echo "threads soak rc=$?"; cat $out/soak_threads.json
( timeout -k 10 200 python tools/ensemble_rate.py 512 1,2,4 300; timeout -k 10 200 python tools/ensemble_rate.py 1024 1,2 200; timeout -k 10 200 python tools/ensemble_rate.py 256 1,4 400 ) 2>/dev/null | tee $out/ensemble_on_one_gpu.jsonl
timeout -k 10 500 python tools/gpu/r4_soak_ensemble.py 512 4 100000 10000 2>/dev/null | tail -1 | cut -c1-600 | tee $out/soak_ensemble.json
timeout -k 10 200 python -m pytest tests/test_zz_perf_guard.py -q -m gpu -s 2>&1 | tail -4
