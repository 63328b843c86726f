#!/usr/bin/env python3
"""Is an int8-products trajectory a function of (W0, dt, steps) alone?  The same call from differently used heaps."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quflow_amd as qfa
from quflow_amd.context import release_contexts
os.environ["QUFLOW_HIP_I8_MIN_N"] = "64"
def run(N, steps, mode, seed=0):
    if mode == "f64": os.environ.pop("QUFLOW_HIP_GEMM", None)
    else: os.environ["QUFLOW_HIP_GEMM"] = mode
    release_contexts()
    W0 = qfa.ensemble.make_W0(N, seed)
    st = {"iterations": 0.0}
    W = qfa.isomp(W0.copy(), 0.25 * qfa.hbar(N), steps=steps, stats=st)
    return hashlib.sha1(W.tobytes()).hexdigest()[:12], st["iterations"], W
for mode in sys.argv[1].split(","):
    h = []
    run(256, 10, mode); a = run(1024, 40, mode); h.append(a[:2])
    b = run(1024, 40, mode); h.append(b[:2])
    run(1024, 5, "f64"); c = run(1024, 40, mode); h.append(c[:2])
    run(2048, 1, mode); d = run(1024, 40, mode); h.append(d[:2])
    print(mode, h, "maxdiff a-b %.2e a-c %.2e a-d %.2e" % (np.abs(a[2]-b[2]).max(), np.abs(a[2]-c[2]).max(), np.abs(a[2]-d[2]).max()), flush=True)
release_contexts()
