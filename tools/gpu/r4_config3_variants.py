#!/usr/bin/env python3
"""Config 3 variants measured side by side (round 3; round 4 adds i8x65 -- six digits for the first product, five for the second -- and i8x6f -- int8 first product, fp64 triangle second): fp64 products, six / five int8 digits for both products, and
the hybrid (fp64 first product, digit-split second product) -- distance from the CPU oracle's fp64 run on the same
W0 after `steps` steps, spectrum / Casimir drift, iteration counts, and the rate over 200 resident steps."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
from oracle import isomp_oracle as oracle
W0 = oracle.make_W0(N, 0)
spec0, cas0 = oracle.spectrum(W0), oracle.casimirs(W0)
import quflow_amd as qfa
from quflow_amd.context import release_contexts
dt = 0.25 * qfa.hbar(N)
sc = {"iterations": 0.0}
Wc = oracle.isomp(W0.copy(), dt, steps=steps, stats=sc)
print(json.dumps({"mode": "cpu oracle", "its": sc["iterations"], "spec_drift": float(np.abs(oracle.spectrum(Wc) - spec0).max()),
                  "cas_drift": float(np.abs(oracle.casimirs(Wc) - cas0).max())}))
for mode in ("f64", "i8x65", "i8x6", "i8x6f", "i8", "i8hx6", "i8h"):
    if mode == "f64":
        os.environ.pop("QUFLOW_HIP_GEMM", None)
    else:
        os.environ["QUFLOW_HIP_GEMM"] = mode
    os.environ["QUFLOW_HIP_I8_MIN_N"] = "64"
    release_contexts()
    sg = {"iterations": 0.0}
    Wg = qfa.isomp(W0.copy(), dt, steps=steps, stats=sg)
    tr = qfa.DeviceTrajectory(W0)
    for _ in range(6):
        tr.advance(dt, 50)
    tr.sync()
    t0 = time.perf_counter()
    st = tr.advance(dt, 200)
    tr.sync()
    el = time.perf_counter() - t0
    tr.ctx.close()
    print(json.dumps({"mode": mode, "N": N, "steps": steps, "its": sg["iterations"], "max_diff_vs_cpu": float(np.abs(Wg - Wc).max()),
                      "spec_drift": float(np.abs(oracle.spectrum(Wg) - spec0).max()),
                      "cas_drift": float(np.abs(oracle.casimirs(Wg) - cas0).max()),
                      "skew_exact": bool(np.array_equal(Wg, -Wg.conj().T)),
                      "timesteps_per_s": 200 / el, "its_200": st["iterations"]}))
release_contexts()
