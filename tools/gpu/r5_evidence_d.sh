#!/bin/bash
# round-5 evidence, part D: config 3 at the round's last kernel change (diagonal tiles of the int8 second product mirror in-tile):
# config 5's long run with config 3's products on white-noise data, a long run on SMOOTH data in both product families
out=gpurun_out/r05_evd; mkdir -p $out
QUFLOW_HIP_GEMM=i8x65 timeout -k 10 400 python tools/longrun.py 2048 10000 1000 > $out/longrun_n2048_10k_steps_i8x65.json 2> $out/longrun_i8x65.progress
python -c "
import json; d=json.load(open('$out/longrun_n2048_10k_steps_i8x65.json')); print('i8x65 N=2048 10k', d['timesteps_per_s'], d['trace'], d['casimir_drift_k234'], d['skew_hermitian_defect'])"
python - <<'PY' > $out/smooth_data_2000_steps_n1024.jsonl
import os, sys, json, time
import numpy as np
sys.path.insert(0, ".")
import quflow_amd as qfa
from quflow_amd.context import release_contexts
N, steps = 1024, 2000
W0 = qfa.ensemble.make_W0(N, 7)
W0 = qfa.solve_poisson(W0).copy(); W0 /= np.linalg.norm(W0, "fro") / np.sqrt(N)       # IC-B: smooth data, 7-8 passes per step
dt = 0.25 * qfa.hbar(N)
for prod in ("f64", "i8x65", "i8x6"):
    if prod == "f64": os.environ.pop("QUFLOW_HIP_GEMM", None)
    else: os.environ["QUFLOW_HIP_GEMM"] = prod
    release_contexts()
    tr = qfa.DeviceTrajectory(W0)
    e0, s0 = tr.diagnostics()
    t0 = time.perf_counter(); st = tr.advance(dt, steps); tr.sync(); el = time.perf_counter() - t0
    W = tr.download(); e1, s1 = tr.diagnostics(); tr.ctx.close()
    H = 1j * W; H2 = H @ H
    print(json.dumps({"N": N, "steps": steps, "products": prod, "initial_data": "IC-B (Poisson-smoothed)", "timesteps_per_s": steps / el,
                      "iterations_per_step": st["iterations"], "abs_trace": abs(np.trace(W)), "skew_hermitian_exact": bool(np.array_equal(W, -W.conj().T)),
                      "energy_drift": e1 - e0, "enstrophy_drift": s1 - s0, "C2": float(np.trace(H2).real / N), "C3": float(np.trace(H2 @ H).real / N)}), flush=True)
PY
cat $out/smooth_data_2000_steps_n1024.jsonl | cut -c1-400
