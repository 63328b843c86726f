#!/usr/bin/env python3
"""BASELINE.json config 5: N=2048 fp64 long-time run (10k steps) on one MI355X, with the
conservation record (energy, enstrophy, Casimirs via host eigen-free traces every chunk).
stdout = one JSON object (redirect it into profiles/); per-chunk progress lines go to stderr."""
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
    c64 = len(sys.argv) > 4 and sys.argv[4] == "c64"       # complex64 state: the float32 arithmetic path
    W0 = qfa.ensemble.make_W0(N, 0)
    if c64:
        W0 = W0.astype(np.complex64)
    peak = 157.3e12 if c64 else 78.6e12
    csize = 8.0 if c64 else 16.0
    dt = 0.25 * qfa.hbar(N)
    tr = qfa.DeviceTrajectory(W0)
    e0, s0 = tr.diagnostics()
    c0 = casimirs(W0.astype(np.complex128))
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
        print(json.dumps(rows[-1]), file=sys.stderr, flush=True)     # progress; stdout carries ONE JSON object
    W = tr.download()
    c1 = casimirs(W.astype(np.complex128))
    # roofline report of the run (BASELINE.json config 5 asks for the HBM-bandwidth one): algorithmic bytes
    # and executed flops per step (per iteration (40 + 240) N^2 bytes of the minimal fused
    # schedule and the 3M products -- 6 N^3 for the first, the upper-triangle tile share of 6 N^3 for the
    # second; per step the W update, 48 N^2 bytes) at the measured rate, against 8 TB/s and 78.6 TFLOP/s
    its = float(np.mean([r["iterations"] for r in rows]))
    nt = N // 64
    share2 = (nt * (nt + 1) / 2) / (nt * nt) if (N % 64 == 0 and N >= 768) else 1.0
    bytes_step = (its * 280.0 * N * N + 48.0 * N * N) * csize / 16.0
    flops_step = its * 6.0 * N ** 3 * (1.0 + share2)
    rate = steps / tgpu
    bound_s = flops_step / peak + bytes_step / 8e12
    roof = {"iterations_per_step": its, "algorithmic_bytes_per_step": bytes_step, "executed_flops_per_step": flops_step,
            "achieved_GBs_algorithmic": bytes_step * rate / 1e9, "hbm_peak_GBs": 8000.0,
            "hbm_frac": bytes_step * rate / 8e12,
            "achieved_TFLOPs_executed": flops_step * rate / 1e12, "mfma_peak_TFLOPs": peak / 1e12,
            "mfma_frac_executed": flops_step * rate / peak,
            "bound_ms_per_step": 1e3 * bound_s, "measured_ms_per_step": 1e3 / rate, "whole_step_frac": bound_s * rate,
            "reading": "the step is MFMA-bound at this size (arithmetic intensity N/6 flop per byte): the HBM-side "
                       "fraction is the share of the 8 TB/s the minimal fused schedule's bytes would need at the measured rate"}
    out = {"N": N, "dtype": "complex64" if c64 else "complex128", "steps": steps, "chunk": chunk, "stepsize": 0.25, "timesteps_per_s": steps / tgpu,
           "wall_s": time.perf_counter() - t0, "energy0": e0, "enstrophy0": s0,
           "casimir_drift_k234": [a - b for a, b in zip(c1, c0)],
           "skew_hermitian_defect": float(np.abs(W + W.conj().T).max()), "trace": complex(np.trace(W)).__repr__(),
           "roofline": roof, "chunks": rows}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
