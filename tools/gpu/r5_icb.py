#!/usr/bin/env python3
"""Config 3's products in the smooth regime (IC-B: ~7 iterations per step, a large stream function): distance from the
fp64 products' trajectory and conservation over `steps` steps at N, both from the same W0 on a resident trajectory."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import isomp_oracle as oracle
import quflow_amd as qfa
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["f64", "i8x65", "i8x6"]
W0 = oracle.make_W0_smooth(N, 0)
spec0, cas0 = oracle.spectrum(W0), oracle.casimirs(W0)
dt = 0.25 * qfa.hbar(N)
ref = {}
for mode in modes:
    if mode == "f64": os.environ.pop("QUFLOW_HIP_GEMM", None)
    else: os.environ["QUFLOW_HIP_GEMM"] = mode
    tr = qfa.DeviceTrajectory(W0)
    e0, s0 = tr.diagnostics()
    row = {"mode": mode, "N": N, "ic": "B (smoothed)", "checkpoints": []}
    done = 0
    for chunk in (2, 8, 40, steps - 50):
        t0 = time.perf_counter(); st = tr.advance(dt, chunk, diagnostics=True); el = time.perf_counter() - t0
        done += chunk
        W = tr.download()
        cp = {"step": done, "its": st["iterations"], "energy_drift": st["energy"] - e0, "enstrophy_drift": st["enstrophy"] - s0,
              "spec_drift": float(np.abs(oracle.spectrum(W) - spec0).max()), "cas_drift": float(np.abs(oracle.casimirs(W) - cas0).max()),
              "trace": float(abs(np.trace(W))), "timesteps_per_s": chunk / el}
        if mode != "f64": cp["maxdiff_vs_f64"] = float(np.abs(W - ref[done]).max())
        row["checkpoints"].append(cp)
        if mode == "f64":
            ref[done] = W
    tr.ctx.close()
    print(json.dumps(row), flush=True)
