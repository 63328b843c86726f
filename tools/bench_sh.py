#!/usr/bin/env python3
"""Measures the spherical-harmonics <-> matrix transforms on the device:
state-resident forms (NULL matrix pointer: W stays in HBM), HIP-event stopwatch on the ctx
stream, algorithmic bytes = the basis block sweep + W + omega.  One JSON line per (N, direction).

    python tools/bench_sh.py 512 1024
"""
import ctypes
import json
import os
import sys
import time

for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS"):
    os.environ.setdefault(_v, "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import quflow_amd as qfa
from quflow_amd import _lib, quantization as q
from quflow_amd.context import ptr

HBM_PEAK_GBS = 8000.0

for N in [int(a) for a in sys.argv[1:]] or [512]:
    t0 = time.time()
    basis = q.get_basis(N)
    t_basis = time.time() - t0
    ctx = q._resident_context(N)
    lib, h = ctx._lib, ctx.handle
    rng = np.random.default_rng(0)
    omega = rng.standard_normal(N * N)
    out = np.zeros(N * N)
    nn = ctypes.c_longlong(N * N)
    bytes_alg = basis.nbytes + 16 * N * N + 8 * N * N
    for name, call in (("shr2mat", lambda: lib.qf_shr2mat(h, ptr(omega), nn, None)),
                       ("mat2shr", lambda: lib.qf_mat2shr(h, None, ptr(out), nn)),
                       ("mat2shr_resident", lambda: lib.qf_mat2shr(h, None, None, nn)),
                       ("shr2mat_resident", lambda: lib.qf_shr2mat(h, None, nn, None))):
        for _ in range(3):
            _lib.check(call())
        reps = 10
        ms_best, ms_sum = 1e30, 0.0
        for _ in range(reps):
            # the timed region holds the H2D/D2H of omega (8 N^2 bytes) as well: reported separately below
            _lib.check(lib.qf_timer_start(h))
            _lib.check(call())
            ms = ctypes.c_double()
            _lib.check(lib.qf_timer_stop(h, ctypes.byref(ms)))
            ms_best = min(ms_best, ms.value)
            ms_sum += ms.value
        ms_avg = ms_sum / reps
        print(json.dumps({"workload": name, "N": N, "ms_avg": ms_avg, "ms_best": ms_best,
                          "algorithmic_bytes": bytes_alg, "GBps_avg": bytes_alg / ms_avg / 1e6,
                          "frac_of_hbm_peak": bytes_alg / ms_avg / 1e6 / HBM_PEAK_GBS,
                          "includes": ("pack/unpack kernels; omega stays on the device" if name.endswith("_resident")
                                       else "PCIe copy of omega (8 N^2 B) + pack/unpack kernels"),
                          "basis_GB": basis.nbytes / 1e9,
                          "basis_compute_plus_download_seconds": t_basis}), flush=True)
    # round trip sanity (after the resident pair the state is still shr2mat(omega))
    _lib.check(lib.qf_mat2shr(h, None, ptr(out), nn))
    assert np.abs(out - omega).max() <= 1e-10 * np.abs(omega).max()
