"""One-off (round 6): what a host-in / host-out call pays for handing back a FRESH 16 N^2-byte ndarray (qfa.laplace at
N = 2048 read 25.7 ms where solve_poisson, which returns a persistent buffer as the reference's does, reads 2.4 ms).
Variants of the destination: np.empty (untouched pages), np.empty + one store per 4 KiB page before the download,
download into a persistent buffer + ndarray.copy()."""
import sys, time
sys.path.insert(0, ".")
import ctypes
import numpy as np
import quflow_amd as qfa
from quflow_amd import _lib
from quflow_amd.context import get_context, ptr

for N in (1024, 2048, 4096):
    P = qfa.solve_poisson(qfa.ensemble.make_W0(N, 0)).copy()
    ctx = get_context(N)
    lib = ctx._lib
    keep = np.empty_like(P)

    def v_empty():
        W = np.empty_like(P)
        _lib.check(lib.qf_laplace(ctx.handle, ptr(P), ptr(W)))
        return W

    def v_touch():
        W = np.empty_like(P)
        W.reshape(-1)[::256] = 0          # one store per 4 KiB page
        _lib.check(lib.qf_laplace(ctx.handle, ptr(P), ptr(W)))
        return W

    def v_persistent_copy():
        _lib.check(lib.qf_laplace(ctx.handle, ptr(P), ptr(keep)))
        return keep.copy()

    def v_persistent():
        _lib.check(lib.qf_laplace(ctx.handle, ptr(P), ptr(keep)))
        return keep

    ref = v_persistent().copy()
    for name, fn in (("np.empty", v_empty), ("np.empty + touch", v_touch), ("persistent + copy", v_persistent_copy), ("persistent (no fresh array)", v_persistent)):
        fn()
        reps = 20 if N <= 2048 else 6
        held = []
        t0 = time.perf_counter()
        for _ in range(reps):
            held.append(fn())            # (results are kept alive, as a caller would: freed pages are not recycled)
            if len(held) > 3:
                held.pop(0)
        t = (time.perf_counter() - t0) / reps
        assert np.array_equal(held[-1], ref)
        print("N=%d %-28s %.3f ms per call" % (N, name, 1e3 * t))
