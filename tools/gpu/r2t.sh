#!/bin/bash
export TMPDIR=/tmp
for m in sleep tiny_first plain; do timeout -k 10 300 python tools/gpu/r2t_one.py $m; done
