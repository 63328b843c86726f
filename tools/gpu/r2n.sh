#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 200 python - <<'PY'
import os, sys, time, json, ctypes
sys.path.insert(0, ".")
import numpy as np
import quflow_amd as qfa
from quflow_amd import _lib
sys.path.insert(0, "tools")
import ensemble_rate as er
print("fresh", er.rate(512, 4, 300))
# what bench.py does before: a main trajectory with events, scratch trajectories created and closed
W0 = qfa.ensemble.make_W0(1024, 0)
dt = 0.25 * qfa.hbar(1024)
tr = qfa.DeviceTrajectory(W0)
lib, h = tr.ctx._lib, tr.ctx.handle
scratch = qfa.DeviceTrajectory(W0); scratch.advance(dt, 10); scratch.sync(); scratch.ctx.close()
_lib.check(lib.qf_profile_reset(h)); _lib.check(lib.qf_profile_stride(h, 4)); _lib.check(lib.qf_profile_enable(h, 2))
tr.advance(dt, 20); tr.sync(); _lib.check(lib.qf_profile_enable(h, 0))
print("after main trajectory with events + a closed scratch", er.rate(512, 4, 300))
for i in range(6):
    t = qfa.DeviceTrajectory(qfa.ensemble.make_W0(512, 0)); t.advance(0.25 * qfa.hbar(512), 5); t.sync(); t.ctx.close()
print("after 6 more created/closed contexts", er.rate(512, 4, 300))
os.environ["QUFLOW_HIP_GEMM"] = "i8x6"; os.environ["QUFLOW_HIP_I8_MIN_N"] = "64"
t = qfa.DeviceTrajectory(W0); t.advance(dt, 5); t.sync(); t.ctx.close()
os.environ.pop("QUFLOW_HIP_GEMM"); os.environ.pop("QUFLOW_HIP_I8_MIN_N")
print("after an i8x6 context", er.rate(512, 4, 300))
PY
