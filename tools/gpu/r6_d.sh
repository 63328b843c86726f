#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r06_d; mkdir -p $out
timeout -k 10 300 python tools/gpu/r6_d2h_fresh.py > $out/d2h_fresh_array.txt 2>&1; cat $out/d2h_fresh_array.txt
