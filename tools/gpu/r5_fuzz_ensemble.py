#!/usr/bin/env python3
"""Randomised check that every member of a DeviceEnsemble (k trajectories sharing one GPU, one host thread feeding their
streams) is bit-identical to its own single-trajectory run, chunk for chunk: random k, sizes (every protocol), chunkings.
No oracle involved.  Usage: python tools/gpu/r5_fuzz_ensemble.py [cases] [seed]"""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
import quflow_amd as qfa  # noqa: E402


def main(cases=30, seed=0):
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(cases):
        N = int(rng.choice([64, 96, 100, 128, 256, 320, 512, 768, 1024]))
        k = int(rng.integers(2, 5))
        budget = 12 if N <= 256 else 8 if N <= 512 else 4
        chunks = []
        while sum(chunks) < budget:
            chunks.append(int(rng.integers(1, 5)))
        dt = float(rng.choice([0.1, 0.25, 0.5])) * qfa.hbar(N)
        seeds = [int(s) for s in rng.integers(0, 1000, size=k)]
        W0s = [qfa.ensemble.make_W0(N, s) for s in seeds]
        ens = qfa.DeviceEnsemble(W0s)
        stats_e = []
        for n in chunks:
            stats_e.append([st["total_iterations"] for st in ens.advance(dt, n)])
        We = ens.download()
        ok = True
        for j, W0 in enumerate(W0s):
            tr = qfa.DeviceTrajectory(W0)
            its = [tr.advance(dt, n)["total_iterations"] for n in chunks]
            Ws = tr.download()
            tr.ctx.close()
            if not np.array_equal(Ws, We[j]) or its != [row[j] for row in stats_e]:
                ok = False
        ens.close()
        bad += not ok
        print(json.dumps({"case": c, "ok": bool(ok), "N": N, "k": k, "chunks": chunks}), flush=True)
    print("cases %d, members not bit-identical to their single runs: %d" % (cases, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
