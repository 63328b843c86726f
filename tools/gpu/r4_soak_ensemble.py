#!/usr/bin/env python3
"""Soak of k replicas sharing one GPU (DeviceEnsemble / qf_isomp_multi) with the round's kernels: the stream-K waits of
k_zgemm_tri under co-scheduling.  Usage: r4_soak_ensemble.py N k steps chunk"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(v, "8")
import numpy as np
import quflow_amd as qfa

N, k, steps, chunk = (int(x) for x in sys.argv[1:5])
dt = 0.25 * qfa.hbar(N)
W0s = [qfa.ensemble.make_W0(N, s) for s in range(k)]
ens = qfa.DeviceEnsemble(W0s)
t0 = time.perf_counter()
its = []
for c in range(0, steps, chunk):
    st = ens.advance(dt, chunk)
    ens.sync()
    its.append(st["iterations"] if isinstance(st, dict) else [s["iterations"] for s in st])
    print(json.dumps({"done": c + chunk, "elapsed_s": round(time.perf_counter() - t0, 2)}), file=sys.stderr, flush=True)
el = time.perf_counter() - t0
Ws = ens.download()
# every member against its own single-trajectory run of the same chunking (bit-identical by construction)
tr = qfa.DeviceTrajectory(W0s[k - 1])
for c in range(0, steps, chunk):
    tr.advance(dt, chunk)
tr.sync()
same = bool(np.array_equal(tr.download(), Ws[k - 1]))
print(json.dumps({"N": N, "replicas": k, "steps": steps, "chunk": chunk, "sum_timesteps_per_s": k * steps / el,
                  "skew_hermitian_exact": [bool(np.array_equal(W, -W.conj().T)) for W in Ws],
                  "last_member_bit_identical_to_its_single_run": same}))
