import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quflow_amd as qfa
from quflow_amd import _lib
from quflow_amd.context import get_context, ptr
for N in (64, 128, 256, 1024):
    rng = np.random.default_rng(N)
    A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    B = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    B = np.ascontiguousarray(B - B.conj().T)
    C = np.zeros_like(A)
    ctx = get_context(N)
    _lib.check(ctx._lib.qf_zgemm_i8(ctx.handle, ptr(A), ptr(B), ptr(C)))
    ref = A @ B
    err = np.abs(C - ref)
    rel = err / (np.abs(A).max(axis=1, keepdims=True) * np.abs(B).max(axis=0, keepdims=True))
    print(os.environ.get("QUFLOW_HIP_GEMM"), N, "max rel err %.3e (2^%.1f)" % (rel.max(), np.log2(rel.max())),
          "re %.2e im %.2e" % (np.abs((C - ref).real).max(), np.abs((C - ref).imag).max()))
