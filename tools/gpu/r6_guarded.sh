#!/bin/bash
# round 6: the GPU suite and the every-size sweep with every device allocation of the library fenced by guard zones
# (QUFLOW_HIP_DEBUG_GUARD=1, csrc/guard.hip; tests/conftest.py fails the test after which a zone is damaged).
# Usage: gpurun --timeout 1200 -- bash tools/gpu/r6_guarded.sh [suite|sweep|all]
export TMPDIR=/tmp
part=${1:-all}
out=gpurun_out/r06_guard; mkdir -p $out
export QUFLOW_HIP_DEBUG_GUARD=1
if [ $part = all ] || [ $part = suite ]; then
  timeout -k 10 1000 python -m pytest tests -x -q -m gpu -rP > $out/pytest_gpu_guarded.txt 2>&1; rc=$?
  grep -h 'perf guard' $out/pytest_gpu_guarded.txt | cut -c1-200; tail -6 $out/pytest_gpu_guarded.txt; [ $rc = 0 ] || exit $rc
fi
if [ $part = all ] || [ $part = sweep ]; then
  timeout -k 10 1000 python tools/gpu/r6_every_size.py ${2:-1200} ${3:-700} ${4:-400} > $out/every_size_guarded.txt 2>&1; rc=$?
  tail -8 $out/every_size_guarded.txt; [ $rc = 0 ] || exit $rc
fi
