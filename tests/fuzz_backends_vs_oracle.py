#!/usr/bin/env python3
"""Randomised differential run of the Laplacian backends and the other steppers against the CPU oracle (test infrastructure,
lives under tests/: it imports oracle/): solve_poisson (complex128 / complex64, states with a trace, general matrices with
select_skewherm(False)), laplace, helmholtz / heat / viscdamp / globalqg, euler / heun / rk4 (with forcing), isomp_simple,
isomp_quasinewton, magmp, isomp on complex64 states -- random sizes around the tile and chunk edges.  Prints one line per
case and a summary; returns the number of disagreements.  Usage: python tests/fuzz_backends_vs_oracle.py [cases] [seed]"""
import json
import os
import sys

for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    # the oracle's BLAS sizes its pool by the VISIBLE cpus (256 on the GPU host, 16 usable): oversubscribed, a case takes seconds
    os.environ.setdefault(_v, "8")
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quflow_amd as qfa  # noqa: E402
from oracle import isomp_oracle as oracle  # noqa: E402

SIZES = [2, 3, 5, 8, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 96, 100, 127, 128, 129, 160, 200, 255, 256, 257]
EPS = np.finfo(np.float64).eps
EPS32 = float(np.finfo(np.float32).eps)


def _mx(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max())


def one_case(rng, sizes):
    kind = str(rng.choice(["poisson", "poisson_trace", "poisson_c64", "poisson_general", "laplace", "helmholtz", "heat", "viscdamp",
                           "globalqg", "euler", "heun", "rk4", "rk4_forcing", "simple", "quasinewton", "magmp", "isomp_c64"]))
    N = int(rng.choice(sizes))
    seed = int(rng.integers(0, 1000))
    W = oracle.make_W0(N, seed)
    lap = qfa.laplacian
    info = {"kind": kind, "N": N}
    if kind == "poisson":
        d, c = lap.solve_poisson(W).copy(), oracle.solve_poisson(W).copy()
        return info, _mx(d, c), 1e-14 * N * N
    if kind == "poisson_trace":
        Wt = W + (0.3 + 0.7j) * np.eye(N)              # tr(W)/N is removed from the right-hand side, cpu.py:311-317
        d, c = lap.solve_poisson(Wt).copy(), oracle.solve_poisson(Wt).copy()
        return info, _mx(d, c), 1e-14 * N * N
    if kind == "poisson_c64":
        W32 = W.astype(np.complex64)
        d, c = lap.solve_poisson(W32).copy(), oracle.solve_poisson(W32).copy()
        assert d.dtype == np.complex64
        return info, _mx(d, c), 64 * EPS32 * max(1.0, float(np.abs(c).max())) * max(1, N // 16)
    if kind == "poisson_general":
        A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
        A -= np.trace(A) / N * np.eye(N)
        lap.select_skewherm(False)
        oracle.select_skewherm(False)
        try:
            d, c = lap.solve_poisson(A).copy(), oracle.solve_poisson(A).copy()
        finally:
            lap.select_skewherm(True)
            oracle.select_skewherm(True)
        return info, _mx(d, c), 1e-14 * N * N * max(1.0, float(np.abs(A).max()))
    if kind == "laplace":
        P = oracle.solve_poisson(W).copy()
        return info, _mx(lap.laplace(P), oracle.laplace(P)), 64 * EPS * N * N * float(np.abs(P).max())
    if kind == "helmholtz":
        a = float(rng.choice([0.1, 1.0, 7.5]))
        d, c = lap.solve_helmholtz(W, alpha=a).copy(), oracle.solve_helmholtz(W, alpha=a).copy()
        return info, _mx(d, c), 1e-13 * max(1.0, float(np.abs(c).max()))
    if kind == "heat":
        h = float(rng.choice([1e-4, 1e-2, 0.3]))
        d, c = lap.solve_heat(h, W).copy(), oracle.solve_heat(h, W).copy()
        return info, _mx(d, c), 1e-13 * max(1.0, float(np.abs(c).max()))
    if kind == "viscdamp":
        h = float(rng.choice([1e-3, 0.05]))
        d, c = lap.solve_viscdamp(h, W, nu=1e-3, alpha=0.05).copy(), oracle.solve_viscdamp(h, W, nu=1e-3, alpha=0.05).copy()
        return info, _mx(d, c), 1e-13 * max(1.0, float(np.abs(c).max()))
    if kind == "globalqg":
        g = float(rng.choice([0.5, 2.0]))
        d, c = lap.solve_globalqg(W, gamma=g).copy(), oracle.solve_globalqg(W, gamma=g).copy()
        return info, _mx(d, c), 1e-13 * max(1.0, float(np.abs(c).max()))
    steps = int(rng.integers(1, 5))
    dt = float(rng.choice([0.05, 0.25, 0.5])) * qfa.hbar(N)
    info["steps"] = steps
    if kind in ("euler", "heun", "rk4", "rk4_forcing"):
        name = "rk4" if kind == "rk4_forcing" else kind
        kw = {}
        if kind == "rk4_forcing":
            kw["forcing"] = lambda P, Wx: -0.05 * Wx + 0.02 * P
        d = getattr(qfa, name)(W.copy(), dt, steps, **kw)
        c = getattr(oracle, name)(W.copy(), dt, steps, **kw)
        return info, _mx(d, c), 1e-12
    if kind == "simple":
        return info, _mx(qfa.isomp_simple(W.copy(), dt, steps), oracle.isomp_simple(W.copy(), dt, steps)), 1e-11
    if kind == "quasinewton":
        sd, sc = {}, {}
        d = qfa.isomp_quasinewton(W.copy(), dt, steps, stats=sd)
        c = oracle.isomp_quasinewton(W.copy(), dt, steps, stats=sc)
        info["its"] = [sd.get("iterations"), sc.get("iterations")]
        # the exit test compares a rounding-level residual with a rounding-level tolerance (eps * stepsize * |W|, isospectral.py:190-191):
        # it asks for a bit-level fixed point, which LAPACK's LU and the Newton-Schulz inverse reach within a pass or two of each other
        # (round 5: the inverse is left alone once E moves by less than its rounding noise; before that the device took one to three
        # passes more); the STATE must agree, the counts may differ by two (test_lu_steppers_vs_oracle_large)
        # (4,000 more cases found two outliers -- N = 2: 5.0 against 7.3 passes, N = 257: 10 against 6 -- so the counts are
        # reported and only a gross difference is flagged; the state is what must agree)
        bad_counts = abs(sd.get("iterations") - sc.get("iterations")) > 5.0
        return info, (np.inf if bad_counts else _mx(d, c)), 1e-10
    if kind == "magmp":
        # a magnetic potential as smooth as the stream function and a step the fixed point converges for (white noise in both
        # slots at dt = 0.5 hbar diverges in the reference as well: nothing to compare)
        S = np.stack([W, oracle.solve_poisson(oracle.make_W0(N, seed + 1)).copy()])
        dtm = float(rng.choice([0.05, 0.1])) * qfa.hbar(N)
        sd, sc = {"iterations": 0.0}, {"iterations": 0.0}
        d = qfa.magmp(S.copy(), dtm, steps, stats=sd)
        c = oracle.magmp_fixedpoint(S.copy(), dtm, steps, stats=sc)
        info["its"] = [sd.get("iterations"), sc.get("iterations")]
        return info, (np.inf if sd.get("iterations") != sc.get("iterations") else _mx(d, c)), 1e-12 * max(1.0, float(np.abs(c).max()))
    if kind == "isomp_c64":
        W32 = W.astype(np.complex64)
        sd, sc = {"iterations": 0.0}, {"iterations": 0.0}
        d = qfa.isomp(W32.copy(), dt, steps=steps, stats=sd)
        c = oracle.isomp(W32.copy(), dt, steps=steps, stats=sc)
        info["its"] = [sd.get("iterations"), sc.get("iterations")]
        assert d.dtype == np.complex64
        return info, (np.inf if sd.get("iterations") != sc.get("iterations") else _mx(d, c)), 2e-4
    raise AssertionError(kind)


def main(cases=200, seed=0, sizes=SIZES, quiet=False):
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(cases):
        state = rng.bit_generator.state
        try:
            info, diff, bound = one_case(rng, sizes)
            err = None
        except Exception as e:      # noqa: BLE001
            probe = np.random.default_rng(0)
            probe.bit_generator.state = state            # replay the draw to name the case that raised
            info = {"kind": str(probe.choice(["poisson", "poisson_trace", "poisson_c64", "poisson_general", "laplace", "helmholtz", "heat",
                                              "viscdamp", "globalqg", "euler", "heun", "rk4", "rk4_forcing", "simple", "quasinewton",
                                              "magmp", "isomp_c64"])), "N": int(probe.choice(sizes))}
            diff, bound, err = np.inf, 0.0, "%s: %s" % (type(e).__name__, e)
        ok = err is None and diff <= bound
        bad += not ok
        if not quiet or not ok:
            print(json.dumps(dict(info, case=c, ok=bool(ok), diff=(None if not np.isfinite(diff) else diff), bound=bound, error=err)), flush=True)
    print("cases %d, disagreements %d" % (cases, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
