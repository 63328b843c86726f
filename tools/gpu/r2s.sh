#!/bin/bash
export TMPDIR=/tmp
for w in other config3 passes none; do timeout -k 10 300 python tools/gpu/r2s_one.py $w; done
timeout -k 10 300 python tools/gpu/r2s_one.py none --no-kernel-events
timeout -k 10 300 python tools/gpu/r2s_one.py none --prewarm-ms 0
