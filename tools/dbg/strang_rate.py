import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quflow_amd as qfa
N = 1024
W0 = qfa.ensemble.make_W0(N, 0)
dt = 0.25 * qfa.hbar(N)
for name, s in (("ViscDampStep (resident)", qfa.ViscDampStep(nu=1e-4, alpha=0.01)),
                ("lambda (host hook)", lambda h, W: qfa.laplacian.solve_viscdamp(h, W, nu=1e-4, alpha=0.01))):
    qfa.isomp(W0.copy(), dt, steps=5, strang_splitting=s)
    t0 = time.perf_counter()
    W = qfa.isomp(W0.copy(), dt, steps=100, strang_splitting=s)
    el = time.perf_counter() - t0
    print("%-26s N=%d: %.1f steps/s" % (name, N, 100 / el), "enstrophy %.12f" % qfa.enstrophy(W))
