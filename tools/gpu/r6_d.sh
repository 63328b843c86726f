#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r06_d; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_hip_envelope.py tests/test_hip_parity.py -x -q -m gpu -k "distinct or commutator or laplace or bracket or geometry or interfaces" 2>&1 | tail -3
