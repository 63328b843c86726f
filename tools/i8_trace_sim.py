#!/usr/bin/env python3
"""Where config 3's trace leak came from (round 5; CPU only, numpy).

The stepper adds 2 (PW - PW^H) to W each step (isospectral.py:509,547); its trace is 4i sum_i Im (Phalf @ Whalf)_ii, and for
skew-Hermitian operands the terms (i, k) and (k, i) of that sum cancel exactly.  This script restates the digit slicing of
csrc/ozaki.hip (row scale s >= 4 max, KD base-128 digits in [-64, 63], pairs a + b < KD kept) for the diagonal entries only
and prints sum_i Im (PW)_ii of the truncated series for four conventions:

    base 128 / row scales      what k_oz_slice does
    base 128 / one scale       the same digits with one power-of-two scale per matrix
    base 127 / row scales      odd-symmetric digits d(-x) = -d(x) in [-63, 63] (the round-4 verdict's "mean-zero digits")
    base 127 / one scale       odd digits AND equal scales: the pairs cancel in the integers (sum = fp64 rounding only)

Finding (N = 1024, seeds 0..3): the sum is 1e-15 ... 2e-14 per product with row scales WHATEVER the digit convention
(three power-of-two scales among the rows of Phalf: the digits of P_ik in row i and of P_ki in row k are then different
numbers and their truncations do not cancel), of either sign depending on W0, and the same from step to step because W
moves slowly -- a linear drift, not a bias of the digit set.  Only "odd digits and one scale" removes it, at the price of
the row scaling's accuracy; the library keeps its slicing and forms the N diagonal imaginary parts in fp64 instead
(k_oz_slice PAIR mode), which is exact to rounding for any scales.

Usage: python tools/i8_trace_sim.py [N] [seeds]"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import isomp_oracle as oracle  # noqa: E402


def digits128(Y, KD):
    B = sum(64 * 128.0 ** -(t + 1) for t in range(KD))
    q = np.rint((Y + B) * 128.0 ** KD).astype(np.int64)
    return [((q >> (7 * (KD - 1 - t))) & 127) - 64 for t in range(KD)]


def digits127(Y, KD):
    q = np.rint(Y * 127.0 ** KD)
    D = [None] * KD
    for t in range(KD - 1, 0, -1):
        c = np.rint(q * (1.0 / 127))
        D[t] = (q - 127 * c).astype(np.int64)
        q = c
    D[0] = q.astype(np.int64)
    return D


def row_scales(X):
    m = np.maximum(np.abs(X.real), np.abs(X.imag)).max(axis=1)
    _, e = np.frexp(m)
    return np.ldexp(1.0, e + 2)


def one_scale(X):
    s = row_scales(X)
    return np.full_like(s, s.max())


def diag_im(P, W, dig, base, KD, scales):
    sP, sW = scales(P), scales(W)
    Pr, Pi = dig(P.real / sP[:, None], KD), dig(P.imag / sP[:, None], KD)
    Wr, Wi = dig(W.real / sW[:, None], KD), dig(W.imag / sW[:, None], KD)
    out = np.zeros(P.shape[0])
    for s in range(KD - 1, -1, -1):
        g = np.zeros(P.shape[0], dtype=np.int64)
        for a in range(s + 1):
            g += (Pr[a] * Wi[s - a] - Pi[a] * Wr[s - a]).sum(axis=1)
        out += g.astype(np.float64) * float(base) ** -(s + 2)
    return out * sP * sW, len(np.unique(sP)), len(np.unique(sW))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    seeds = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3]
    vareps = 0.125                                   # dt / (2 hbar) at dt = 0.25 hbar
    for seed in seeds:
        W = oracle.make_W0(N, seed)
        P = oracle.solve_poisson(W).copy() * vareps
        exact = (P.real * W.imag - P.imag * W.real).sum(axis=1)
        cells = []
        for name, dig, base in (("base128", digits128, 128), ("base127", digits127, 127)):
            for sname, sc in (("row scales", row_scales), ("one scale", one_scale)):
                v, a, b = diag_im(P, W, dig, base, 6, sc)
                if sc is row_scales:
                    nP, nW = a, b
                cells.append("%s/%s %+.2e" % (name, sname, math.fsum(v)))
        print("N=%d seed %d  exact sum %+.1e  scales among rows: Phalf %d, Whalf %d | %s" % (N, seed, math.fsum(exact), nP, nW, " | ".join(cells)))


if __name__ == "__main__":
    main()
