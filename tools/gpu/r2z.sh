#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r2z
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "single_precision or yielding or native or diagnostics or ensemble" > gpurun_out/r2z/sel.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/r2z/sel.log
