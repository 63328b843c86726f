#!/bin/bash
# round 3: folded walk slots in k_solve -- tests, then the probe with and without
export TMPDIR=/tmp
out=gpurun_out/r3j; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_single.py -x -q -k "solve or poisson or tridiagonal or laplac" > $out/pytest.txt 2>&1; tail -15 $out/pytest.txt


