import os, sys, time, json, ctypes
sys.path.insert(0, ".")
import bench
args = bench.parse(["--steps", "20", "--warmup", "5"])
import numpy as np
import quflow_amd as qfa
from quflow_amd import _lib
pieces = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] else []
N = 1024
if "setdev" in pieces:
    qfa.set_device(0)
W0 = qfa.ensemble.make_W0(N, 0); dt = 0.25 * qfa.hbar(N)
tr = qfa.DeviceTrajectory(W0, device=0)
lib, h = tr.ctx._lib, tr.ctx.handle
if "scratch" in pieces:
    s = qfa.DeviceTrajectory(W0, device=0)
    t_end = time.perf_counter() + 0.15
    while time.perf_counter() < t_end:
        s.advance(dt, 10)
    s.sync(); s.ctx.close()
tr.advance(dt, 5)
if "diag" in pieces:
    tr.diagnostics()
if "events" in pieces:
    _lib.check(lib.qf_profile_reset(h)); _lib.check(lib.qf_profile_stride(h, 4)); _lib.check(lib.qf_profile_enable(h, 2))
if "timer" in pieces:
    _lib.check(lib.qf_timer_start(h))
tr.advance(dt, 20)
if "diag" in pieces:
    tr.diagnostics()
if "timer" in pieces:
    ms = ctypes.c_double(); _lib.check(lib.qf_timer_stop(h, ctypes.byref(ms)))
tr.sync()
if "events" in pieces:
    _lib.check(lib.qf_profile_enable(h, 0))
    n = ctypes.c_longlong(); m = ctypes.c_double()
    _lib.check(lib.qf_profile_read(h, 1, ctypes.byref(n), ctypes.byref(m)))
r = bench.replicas_per_gpu_run(args, qfa, 512, 4, 300, 0)
print("%-40s x4 %.0f ratio %.3f" % (",".join(pieces) or "(none)", r["sum_timesteps_per_s"], r["ratio"]), flush=True)
