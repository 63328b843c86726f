#!/bin/bash
export TMPDIR=/tmp
echo "== standalone, 8 BLAS threads"; timeout -k 10 120 python tools/ensemble_rate.py 512 1,4 300
echo "== standalone, 16 BLAS threads"; OPENBLAS_NUM_THREADS=16 OMP_NUM_THREADS=16 timeout -k 10 120 python tools/ensemble_rate.py 512 1,4 300
echo "== standalone after an eigvalsh"; timeout -k 10 120 python - <<'PY'
import os, sys, time, json
sys.path.insert(0, ".")
os.environ["OPENBLAS_NUM_THREADS"] = "16"
import numpy as np
import quflow_amd as qfa
sys.path.insert(0, "tools")
import ensemble_rate as er
A = np.random.randn(1024, 1024); A = A + A.T
t0 = time.time(); np.linalg.eigvalsh(A); print("eigvalsh", time.time() - t0)
for k in (1, 4):
    print(k, er.rate(512, k, 300))
# with a big context alive
tr = qfa.DeviceTrajectory(qfa.ensemble.make_W0(2048, 0)); tr.advance(0.25 * qfa.hbar(2048), 5)
for k in (1, 4):
    print("with N=2048 context alive", k, er.rate(512, k, 300))
PY
