#!/usr/bin/env python3
"""BASELINE.json config 5: N=2048 fp64 long-time run (10k steps) on one MI355X, with the
conservation record (energy, enstrophy, Casimirs via host eigen-free traces every chunk)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import quflow_amd as qfa  # noqa: E402


def casimirs(W):
    H = 1j * W
    N = W.shape[-1]
    H2 = H @ H
    return [float(np.trace(H2).real / N), float(np.trace(H2 @ H).real / N), float(np.trace(H2 @ H2).real / N)]


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    W0 = qfa.ensemble.make_W0(N, 0)
    dt = 0.25 * qfa.hbar(N)
    tr = qfa.DeviceTrajectory(W0)
    e0, s0 = tr.diagnostics()
    c0 = casimirs(W0)
    rows = []
    t0 = time.perf_counter()
    tgpu = 0.0
    for k in range(0, steps, chunk):
        t1 = time.perf_counter()
        st = tr.advance(dt, chunk)
        tr.sync()
        tgpu += time.perf_counter() - t1
        e, s = tr.diagnostics()
        rows.append({"step": k + chunk, "energy_drift": e - e0, "enstrophy_drift": s - s0,
                     "iterations": st["iterations"], "number_of_maxit": st["number_of_maxit"]})
        print(rows[-1], flush=True)
    W = tr.download()
    c1 = casimirs(W)
    out = {"N": N, "steps": steps, "chunk": chunk, "stepsize": 0.25, "timesteps_per_s": steps / tgpu,
           "wall_s": time.perf_counter() - t0, "energy0": e0, "enstrophy0": s0,
           "casimir_drift_k234": [a - b for a, b in zip(c1, c0)],
           "skew_hermitian_defect": float(np.abs(W + W.conj().T).max()), "trace": complex(np.trace(W)).__repr__(),
           "chunks": rows}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
