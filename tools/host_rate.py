#!/usr/bin/env python3
"""Diagnostic: wall vs device time per isomp step over repeated calls in one process
(small N: the stepper is launch-rate bound; shows whether a slow mode is per process or per call)."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quflow_amd as qfa
from quflow_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
W0 = qfa.ensemble.make_W0(N, 0)
dt = 0.25 * qfa.hbar(N)
tr = qfa.DeviceTrajectory(W0)
lib, h = tr.ctx._lib, tr.ctx.handle
tr.advance(dt, 20)
if os.environ.get("HR_DIAG"):
    tr.diagnostics()
if os.environ.get("HR_SLEEP"):
    time.sleep(float(os.environ["HR_SLEEP"]))
out = []
for r in range(reps):
    tr.sync()
    t0 = time.perf_counter()
    _lib.check(lib.qf_timer_start(h))
    st = tr.advance(dt, steps)
    ev = ctypes.c_double()
    _lib.check(lib.qf_timer_stop(h, ctypes.byref(ev)))
    tr.sync()
    el = time.perf_counter() - t0
    out.append("%.3f/%.3f" % (1e3 * el / steps, ev.value / steps))
print("N=%d wall/device ms per step:" % N, " ".join(out), "| ncpu", len(os.sched_getaffinity(0)), "first", min(os.sched_getaffinity(0)))
