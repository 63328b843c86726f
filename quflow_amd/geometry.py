"""Geometry helpers of quflow next to the hot path (quflow/geometry.py): `hbar` and the matrix
norms / inner products the diagnostics are made of on the host (O(N^2) reductions of a state that
is already there), `bracket` with its two products on the device."""
import numpy as np


def hbar(N):
    """hbar(N) = 2/sqrt(N^2-1)  (quflow/geometry.py:7-9)."""
    return 2.0 / np.sqrt(N ** 2 - 1)


def qtime2seconds(qtime, N):
    """Quantum time units -> seconds: qtime * hbar(N)  (quflow/utils.py:206-221)."""
    return qtime * (2.0 / np.sqrt(N ** 2 - 1))


def seconds2qtime(t, N):
    """Seconds -> quantum time units: t / hbar(N)  (quflow/utils.py:224-239)."""
    return t / (2.0 / np.sqrt(N ** 2 - 1))


def _device_matmul(A, B):
    from . import _lib
    from .context import as_c128, get_context, ptr, result_array
    A = as_c128(A, "A")
    B = as_c128(B, "B")
    C = result_array(A.shape, A.dtype, "matmul")
    ctx = get_context(A.shape[-1])
    _lib.check(ctx._lib.qf_zgemm(ctx.handle, ptr(A), ptr(B), ptr(C)))
    return C


def bracket(P, W):
    """[P, W]/hbar (quflow/geometry.py:41-49, dense branch), both products on the device."""
    A = _device_matmul(P, W)
    A -= _device_matmul(W, P)
    A /= hbar(np.asarray(P).shape[-1])
    return A


def norm_L2(W):
    """Scaled Frobenius norm (quflow/geometry.py:53-68)."""
    W = np.asarray(W)
    return np.linalg.norm(W, ord='fro') / np.sqrt(W.shape[-1])


def inner_L2(P, W):
    """Re sum P conj(W) / N (quflow/geometry.py:72-76)."""
    P = np.asarray(P)
    W = np.asarray(W)
    return (P * W.conj()).sum().real / W.shape[-1]


def norm_Linf(W):
    """Spectral norm (quflow/geometry.py:80-92)."""
    return np.linalg.norm(np.asarray(W), ord=2)


def norm_L1(W):
    """Scaled nuclear norm through the eigenvalues (quflow/geometry.py:95-110)."""
    W = np.asarray(W)
    sW = np.abs(np.linalg.eigvals(W))
    sW /= W.shape[-1]
    return sW.sum()


def integral(W):
    """Re(-i tr(W)/N) (quflow/geometry.py:113-129)."""
    W = np.asarray(W)
    trW = np.trace(W) / W.shape[-1]
    return np.real(-1j * trW)
