#!/bin/bash
out=gpurun_out/r2e; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1 || { echo "pytest failed"; tail -60 $out/pytest_gpu.txt; exit 1; }
tail -3 $out/pytest_gpu.txt
