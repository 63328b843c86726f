#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -k "$1" 2>&1 | tail -15
