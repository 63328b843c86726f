"""Diagnostics of the hot path (quflow/physics.py:26-38) evaluated on the device."""
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
