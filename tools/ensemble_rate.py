#!/usr/bin/env python3
"""Sum of timesteps/s of k independent replicas advanced together on ONE GPU (DeviceEnsemble,
qf_isomp_multi) against the single-trajectory rate.  Usage: tools/ensemble_rate.py N k [steps] [c64]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(v, "8")
import quflow_amd as qfa  # noqa: E402


def rate(N, k, steps, warmup=20, c64=False):
    import numpy as np
    dt = 0.25 * qfa.hbar(N)
    W0s = [qfa.ensemble.make_W0(N, s) for s in range(k)]
    if c64:
        W0s = [W.astype(np.complex64) for W in W0s]
    ens = qfa.DeviceEnsemble(W0s)
    t_end = time.perf_counter() + 0.15
    while time.perf_counter() < t_end:
        ens.advance(dt, 10)
    ens.advance(dt, warmup)
    ens.sync()
    t0 = time.perf_counter()
    st = ens.advance(dt, steps)
    ens.sync()
    el = time.perf_counter() - t0
    ens.close()
    return k * steps / el, sum(s["iterations"] for s in st) / k


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    ks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    c64 = len(sys.argv) > 4 and sys.argv[4] == "c64"
    base = None
    for k in ks:
        r, its = rate(N, k, steps, c64=c64)
        if k == 1:
            base = r
        print(json.dumps({"N": N, "replicas_on_one_gpu": k, "sum_timesteps_per_s": r, "iterations_per_step": its,
                          "vs_single": (r / base) if base else None}))
