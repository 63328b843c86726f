#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 400 python - <<'PY'
import os, sys, time, json
sys.path.insert(0, ".")
import bench
args = bench.parse(["--steps", "20", "--warmup", "5"])
import numpy as np
import quflow_amd as qfa
from quflow_amd import _lib
def rep(tag):
    r = bench.replicas_per_gpu_run(args, qfa, 512, 4, 300, 0)
    print(tag, round(r["sum_timesteps_per_s"]), round(r["ratio"], 3), flush=True)
rep("fresh")
N = 1024
W0 = qfa.ensemble.make_W0(N, 0); dt = 0.25 * qfa.hbar(N)
tr = qfa.DeviceTrajectory(W0); tr.advance(dt, 25)
bench.instrumented_pass(qfa, _lib, W0, dt, 20, {}, 0); rep("after instrumented_pass")
bench.fixed_iteration_run(qfa, _lib, W0, N, 0); rep("after fixed_iteration_run")
bench.config3_side_run(args, qfa, tr, W0, dt, {}, 0, "i8x6"); rep("after config3_side_run")
bench.other_size_run(args, qfa, 512, 200, 20, 0); rep("after other_size_run 512")
bench.other_size_run(args, qfa, 2048, 60, 6, 0); rep("after other_size_run 2048")
PY
