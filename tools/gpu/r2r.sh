#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 600 python - <<'PY'
import os, sys, time, json, io, contextlib
sys.path.insert(0, ".")
import bench
orig = bench.replicas_per_gpu_run
def twice(args, qfa, N, k, steps, device, warmup=20):
    a = orig(args, qfa, N, k, steps, device, warmup)
    b = orig(args, qfa, N, k, steps, device, warmup)
    print("pass 1: %.0f (%.3f)  pass 2: %.0f (%.3f)" % (a["sum_timesteps_per_s"], a["ratio"], b["sum_timesteps_per_s"], b["ratio"]), file=sys.stderr, flush=True)
    return b
bench.replicas_per_gpu_run = twice
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--cpu-seconds", "0"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
PY
