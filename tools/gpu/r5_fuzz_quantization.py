#!/usr/bin/env python3
"""Randomised property run of the spherical-harmonics <-> matrix transforms on the device (no oracle): round trips
mat2shr(shr2mat(w)) = w and mat2shc(shc2mat(w)) = w for random coefficient vectors (full and truncated, with and without the
Berezin multipliers), shr2mat's result is skew-Hermitian, shr / shc agree through the real <-> complex coefficient map, linearity,
and the analytic Laplacian eigenvalues: laplace(shr2mat(e_lm)) = -l(l+1) shr2mat(e_lm).  Usage: r5_fuzz_quantization.py [cases] [seed]"""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
import quflow_amd as qfa  # noqa: E402
from quflow_amd import quantization as qz  # noqa: E402


def main(cases=120, seed=0):
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(cases):
        N = int(rng.choice([2, 3, 5, 8, 16, 17, 32, 33, 48, 64, 65, 96, 128, 129, 200, 256]))
        full = rng.random() < 0.5
        # (shorter vectors are read in whole shells: elmax = int(sqrt(len)) - 1, quflow/quantization.py:144-148)
        n = N * N if full else int(rng.integers(1, N + 1)) ** 2
        berezin = bool(rng.random() < 0.25)
        w = rng.standard_normal(n)
        W = qz.shr2mat(w, N=N, berezin=berezin)
        back = qz.mat2shr(W, berezin=berezin)
        scale = max(1.0, float(np.abs(w).max()))
        e_rt = float(np.abs(back[:n] - w).max()) / scale
        e_tail = float(np.abs(back[n:]).max()) if n < N * N else 0.0
        skew = float(np.abs(W + W.conj().T).max())
        # linearity
        w2 = rng.standard_normal(n)
        a = float(rng.standard_normal())
        e_lin = float(np.abs(qz.shr2mat(w + a * w2, N=N, berezin=berezin) - (W + a * qz.shr2mat(w2, N=N, berezin=berezin))).max())
        # complex coefficients: round trip
        wc = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        Wc = qz.shc2mat(wc, N=N, berezin=berezin)
        backc = qz.mat2shc(Wc, berezin=berezin)
        e_rtc = float(np.abs(backc[:n] - wc).max()) / max(1.0, float(np.abs(wc).max()))
        # one basis element is an eigenvector of the quantised Laplacian: Delta T_lm = -l(l+1) T_lm
        ind = int(rng.integers(0, N * N))
        el, m = qz.ind2elm(ind)
        e1 = np.zeros(ind + 1); e1[ind] = 1.0
        T = qz.shr2mat(e1, N=N)
        e_lap = float(np.abs(qfa.laplacian.laplace(T) + el * (el + 1) * T).max()) / max(1.0, el * (el + 1) * float(np.abs(T).max()))
        tol = 2e-12 if N <= 64 else 2e-11
        if berezin:      # the multipliers divide: rounding is amplified by their smallest (quantization.py:48-60; the reference's too)
            tol *= 1.0 / float(np.abs(qz.berezin_multipliers(N)[:n]).min())
        ok = e_rt <= tol and e_tail <= tol * scale and skew <= 1e-13 * scale and e_lin <= (tol if berezin else 1e-12) * scale * max(1.0, abs(a)) and e_rtc <= tol and e_lap <= 1e-10
        bad += not ok
        print(json.dumps({"case": c, "ok": bool(ok), "N": N, "n": n, "berezin": berezin, "round_trip": e_rt, "tail": e_tail, "skew": skew,
                          "linearity": e_lin, "round_trip_complex": e_rtc, "laplace_eigen": e_lap, "l": int(el)}), flush=True)
    print("cases %d, properties violated %d" % (cases, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
