#!/usr/bin/env python3
"""How noisy is "spectrum / Casimir drift after n steps" as a statistic?  For several seeds: the CPU oracle's drift, the
fp64 products' and the int8 modes' on the same W0 (N, steps from argv), and the 2-norm of the state difference that
Weyl's inequality bounds the difference of two drifts with."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
seeds = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 2, 3]
from oracle import isomp_oracle as oracle
import quflow_amd as qfa
from quflow_amd.context import release_contexts
dt = 0.25 * qfa.hbar(N)
for seed in seeds:
    W0 = oracle.make_W0(N, seed)
    spec0, cas0 = oracle.spectrum(W0), oracle.casimirs(W0)
    Wc = oracle.isomp(W0.copy(), dt, steps=steps)
    row = {"N": N, "steps": steps, "seed": seed, "cpu_spec": float(np.abs(oracle.spectrum(Wc) - spec0).max()),
           "cpu_cas": float(np.abs(oracle.casimirs(Wc) - cas0).max())}
    for mode in ("f64", "i8x65", "i8x6", "i8x6f"):
        if mode == "f64":
            os.environ.pop("QUFLOW_HIP_GEMM", None)
        else:
            os.environ["QUFLOW_HIP_GEMM"] = mode
        os.environ["QUFLOW_HIP_I8_MIN_N"] = "64"
        release_contexts()
        Wg = qfa.isomp(W0.copy(), dt, steps=steps)
        row[mode] = {"spec": float(np.abs(oracle.spectrum(Wg) - spec0).max()), "cas": float(np.abs(oracle.casimirs(Wg) - cas0).max()),
                     "maxdiff": float(np.abs(Wg - Wc).max()), "norm2_diff": float(np.linalg.norm(Wg - Wc, 2)),
                     "trace": float(abs(np.trace(Wg)))}
        row[mode]["spec_ratio"] = row[mode]["spec"] / row["cpu_spec"]
        row[mode]["cas_ratio"] = row[mode]["cas"] / row["cpu_cas"]
    print(json.dumps(row), flush=True)
release_contexts()
