#!/usr/bin/env python3
"""Randomised differential run of the device-resident trajectory against the CPU oracle (test infrastructure under tests/: it
imports oracle/): a DeviceTrajectory advanced in random chunks -- every chunk is one stepper call, which starts from dW = 0 as
the reference's does -- against the same chain of oracle calls, at sizes that take every protocol (32x32 and 64x64 tiles,
deferred / fused step end, stream-K triangle), with diagnostics in between and a download at the end; and the same chain through
isomp() on host arrays (upload / download per call).  Usage: python tests/fuzz_trajectory_vs_oracle.py [cases] [seed]"""
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

SIZES = [48, 64, 96, 100, 128, 160, 256, 320, 512, 704, 768, 1024]


def main(cases=40, seed=0, sizes=SIZES, quiet=False):
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(cases):
        N = int(rng.choice(sizes))
        budget = 12 if N <= 256 else 6 if N <= 512 else 3
        chunks = []
        while sum(chunks) < budget:
            chunks.append(int(rng.integers(1, 4)))
        dt = float(rng.choice([0.1, 0.25, 0.5])) * qfa.hbar(N)
        kw = {}
        if rng.random() < 0.25:
            kw["compsum"] = True
        if rng.random() < 0.25:
            kw["maxit"] = int(rng.integers(2, 5))
        W0 = oracle.make_W0(N, int(rng.integers(0, 1000)))
        if rng.random() < 0.3:
            W0 = oracle.solve_poisson(W0).copy()
            W0 /= np.linalg.norm(W0, "fro") / np.sqrt(N)
        tr = qfa.DeviceTrajectory(W0)
        Wc, Wh = W0.copy(), W0.copy()
        ok = True
        notes = []
        for n in chunks:
            sd, sc, sh = {"iterations": 0.0}, {"iterations": 0.0}, {"iterations": 0.0}
            st = tr.advance(dt, n, **kw)
            oracle.isomp(Wc, dt, steps=n, stats=sc, **kw)
            qfa.isomp(Wh, dt, steps=n, stats=sh, **kw)
            e, s = tr.diagnostics()
            ec, scs = oracle.energy_euler(Wc), oracle.enstrophy(Wc)
            if st["iterations"] != sc["iterations"] or sh["iterations"] != sc["iterations"]:
                ok = False
                notes.append(("iterations", st["iterations"], sh["iterations"], sc["iterations"]))
            if abs(e - ec) > 1e-12 * max(1e-30, abs(ec)) + 1e-18 or abs(s - scs) > 1e-12 * abs(scs):
                ok = False
                notes.append(("diagnostics", e, ec, s, scs))
        Wd = tr.download()
        tr.ctx.close()
        d1, d2 = float(np.abs(Wd - Wc).max()), float(np.abs(Wh - Wc).max())
        bitwise = bool(np.array_equal(Wd, Wh))          # resident chain == upload/download chain, bit for bit
        ok = ok and d1 <= 2e-11 and d2 <= 2e-11 and bitwise
        bad += not ok
        if not quiet or not ok:
            print(json.dumps({"case": c, "ok": bool(ok), "N": N, "chunks": chunks, "kw": kw, "diff_resident": d1, "diff_host_calls": d2,
                              "resident_equals_host_calls_bitwise": bitwise, "notes": notes}), flush=True)
    print("cases %d, disagreements %d" % (cases, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
