#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r2x
timeout -k 10 300 python -m pytest tests/test_native_comm.py -x -q -m gpu > gpurun_out/r2x/native.log 2>&1; echo "native rc=$?"; tail -15 gpurun_out/r2x/native.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2x/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/r2x/gpu_suite.log
