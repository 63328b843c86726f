#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3g; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1 || { echo "pytest failed"; tail -60 $out/pytest_gpu.txt; }
tail -3 $out/pytest_gpu.txt
