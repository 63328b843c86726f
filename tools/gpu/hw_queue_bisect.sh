#!/bin/bash
export TMPDIR=/tmp
for p in "" setdev scratch diag events timer "setdev,scratch,diag,events,timer"; do timeout -k 10 200 python tools/gpu/hw_queue_bisect_one.py "$p"; done
