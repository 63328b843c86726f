#!/usr/bin/env python3
"""Round 6: EVERY size, not a sample.  The suite's parametrised sizes sit on both sides of each kernel-selection rule
(DESIGN 3.0); this run takes every N of a range through the device path against the CPU oracle on the same W0:

  solve   every N in [2, NS]    qfa.solve_poisson (skew-Hermitian AND general input) and qfa.laplace
  step    every N in [2, NT]    qfa.isomp, 2 adaptive steps (default options), state + iteration statistics + tol_auto
  c64     every N in [2, NC]    the same for a complex64 state (float32 arithmetic path) against the oracle in complex64

One line per failing size, a summary line per leg.  Usage (GPU box): python tools/gpu/r6_every_size.py [NS] [NT] [NC]
(test infrastructure: imports oracle/)."""
import json
import os
import sys
import time

for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quflow_amd as qfa  # noqa: E402
from oracle import isomp_oracle as oracle  # noqa: E402

oracle.build()
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2300
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 1100
NC = int(sys.argv[3]) if len(sys.argv) > 3 else 600
eps = np.finfo(np.float64).eps


def forget():
    """Contexts (device) and the oracle's per-N buffers (host) are cached for the life of the process: a sweep drops them."""
    qfa.release_contexts()
    oracle._lap_cache.clear()
    oracle._buf_cache.clear()


def leg_solve():
    bad, worst, t0 = [], 0.0, time.time()
    rng = np.random.default_rng(7)
    for N in range(2, NS + 1):
        W = oracle.make_W0(N, N)
        scale = float(np.abs(W).max())
        Pg = qfa.solve_poisson(W).copy()
        Pc = oracle.solve_poisson(W).copy()
        e = float(np.abs(Pg - Pc).max())
        # the reference's own bound (tests/test_laplacian.py:252) and 256 ulp of the data scale
        ok = e <= 1e-14 * N * N and e <= 256 * eps * scale * max(1.0, np.abs(Pc).max() / scale)
        G = (rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N)))
        old = qfa.laplacian.select_skewherm(False)      # the general solver (cpu.py:200-278) on both sides
        oracle.select_skewherm(False)
        try:
            Gg = qfa.solve_poisson(G).copy()
            Gc = oracle.solve_poisson(G).copy()
        finally:
            qfa.laplacian.select_skewherm(old)
            oracle.select_skewherm(True)
        eg = float(np.abs(Gg - Gc).max())
        ok = ok and eg <= 1e-14 * N * N
        Lg = qfa.laplace(Pc).copy()
        Lc = oracle.laplace(Pc).copy()
        ok = ok and np.array_equal(Lg, Lc)
        worst = max(worst, e / (N * N), eg / (N * N))
        if not ok:
            bad.append(N)
            print(json.dumps({"leg": "solve", "N": N, "err_skewh": e, "err_general": eg, "laplace_bit_exact": bool(np.array_equal(Lg, Lc))}), flush=True)
        forget()
        if N % 250 == 0:
            print("solve: up to N = %d, %d failing, %.0f s" % (N, len(bad), time.time() - t0), flush=True)
    print(json.dumps({"leg": "solve", "sizes": NS - 1, "failing": bad, "worst_err_over_N2": worst, "seconds": round(time.time() - t0, 1)}), flush=True)
    return bad


def leg_step(dtype, top, tol_state):
    bad, worst, t0 = [], 0.0, time.time()
    for N in range(2, top + 1):
        W0 = oracle.make_W0(N, 1000 + N).astype(dtype)
        dt = 0.25 * qfa.hbar(N)
        sg, sc = {}, {}
        Wg = qfa.isomp(W0.copy(), dt, steps=2, stats=sg)
        Wc = oracle.isomp(W0.copy(), dt, steps=2, stats=sc)
        e = float(np.abs(Wg.astype(np.complex128) - Wc.astype(np.complex128)).max())
        same = (sg.get("iterations") == sc.get("iterations") and sg.get("number_of_maxit") == sc.get("number_of_maxit"))
        ta, tb = sg.get("tol_auto"), sc.get("tol_auto")
        tol_ok = (ta is None and tb is None) or (ta is not None and tb is not None and abs(ta - tb) <= (1e-12 if dtype == np.complex128 else 1e-6) * abs(tb))
        skew = bool(np.array_equal(Wg, -Wg.conj().T))
        bound = tol_state if dtype == np.complex128 else tol_state * float(np.abs(Wc).max())   # c64: relative to the state's scale
        ok = e <= bound and same and tol_ok and skew and Wg.dtype == Wc.dtype
        worst = max(worst, e)
        if not ok:
            bad.append(N)
            print(json.dumps({"leg": "step", "dtype": np.dtype(dtype).name, "N": N, "err": e, "device": sg, "oracle": sc, "skew_hermitian_exact": skew}, default=float), flush=True)
        forget()
        if N % 100 == 0:
            print("step %s: up to N = %d, %d failing, worst %.2e, %.0f s" % (np.dtype(dtype).name, N, len(bad), worst, time.time() - t0), flush=True)
    print(json.dumps({"leg": "step", "dtype": np.dtype(dtype).name, "sizes": top - 1, "failing": bad, "worst_state_err": worst, "seconds": round(time.time() - t0, 1)}), flush=True)
    return bad


if __name__ == "__main__":
    b = []
    if NS >= 2:
        b += leg_solve()
    if NT >= 2:
        b += leg_step(np.complex128, NT, 1e-11)
    if NC >= 2:
        b += leg_step(np.complex64, NC, 1e-5)
    print("every-size run:", "0 disagreements" if not b else "%d FAILING sizes" % len(b))
    damaged = 0
    if os.environ.get("QUFLOW_HIP_DEBUG_GUARD", "0") not in ("", "0"):
        allocs, damaged, first = qfa.guard_report()
        print("guard zones (QUFLOW_HIP_DEBUG_GUARD): %d device allocations fenced, %d damaged zones %s" % (allocs, damaged, first))
    sys.exit(1 if (b or damaged) else 0)
