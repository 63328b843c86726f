"""Diagnostics of the hot path (quflow/physics.py:26-38) evaluated on the device, and the Sobolev
inner products next to them (quflow/physics.py:9-21: one device solve / stencil, one host reduction)."""
import ctypes

import numpy as np

from . import _lib
from .context import as_c128, get_context, ptr


def _diagnostics(W):
    W = np.asarray(W)
    if W.ndim >= 3:
        W = W[(0,) * (W.ndim - 2) + (Ellipsis,)]
    Wc = as_c128(W, "W")
    ctx = get_context(Wc.shape[-1])
    _lib.check(ctx._lib.qf_upload_W(ctx.handle, ptr(Wc)))
    e = ctypes.c_double()
    s = ctypes.c_double()
    _lib.check(ctx._lib.qf_diagnostics(ctx.handle, ctypes.byref(e), ctypes.byref(s)))
    return e.value, s.value


def energy_euler(W):
    """E = -<W, Delta^-1 W>/2, quflow/physics.py:26-32."""
    return _diagnostics(W)[0]


def enstrophy(W):
    """S = <W, W>/2, quflow/physics.py:34-38."""
    return _diagnostics(W)[1]


def inner_Hm1(W1, W2):
    """-<W1, Delta^-1 W2>, quflow/physics.py:9-11."""
    from .geometry import inner_L2
    from .laplacian import solve_poisson
    return -inner_L2(W1, solve_poisson(W2))


def norm_Hm1(W):
    """quflow/physics.py:13-14."""
    return np.sqrt(inner_Hm1(W, W))


def inner_H1(P1, P2):
    """-<P1, Delta P2>, quflow/physics.py:16-18."""
    from .geometry import inner_L2
    from .laplacian import laplace
    return -inner_L2(P1, laplace(P2))


def norm_H1(P):
    """quflow/physics.py:20-21."""
    return np.sqrt(inner_H1(P, P))
