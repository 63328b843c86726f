"""Laplacian backend module protocol of quflow, on the MI355X.

Mirrors `quflow.laplacian` (default backend quflow/laplacian/cpu.py re-exported by
quflow/laplacian/__init__.py:1): an object with `solve_poisson(W)`, `laplace(P)`,
`laplacian(N, bc)`, `select_skewherm(flag)` and `__name__` -- the interface the
reference's tests parametrise over (tests/test_laplacian.py:134-152,226-252).
Everything below runs hand-written HIP kernels through the C ABI
(include/quflow_hip.h); there is no CPU path.
"""
import collections
import ctypes

import numpy as np

from . import _lib
from . import geometry as _geometry
from .context import as_c128, get_context, ptr, result_array

_SKEW_HERM_ = True
_out_cache = {}


def select_skewherm(flag):
    """quflow/laplacian/cpu.py:563-591: returns the previous flag."""
    global _SKEW_HERM_
    old = _SKEW_HERM_
    _SKEW_HERM_ = bool(flag)
    return old


def select_first(W):
    """quflow/laplacian/cpu.py:672-674: state 0 of a (..., N, N) stack, contiguous -- the default `reduce` of
    solve_poisson (:681, 696-697)."""
    return np.ascontiguousarray(W[(0,) * (W.ndim - 2) + (Ellipsis,)])


def select_sum(W):
    """quflow/laplacian/cpu.py:677-678: the sum of the states -- `reduce=select_sum` makes every state of a stack a
    source of the one stream matrix."""
    return W.sum(axis=tuple(range(W.ndim - 2)))


_reduce_first = select_first


def allocate_buffer(W):
    """quflow/laplacian/cpu.py:594-601 warms the solver's per-N buffers for W's size and dtype.  Here: the device
    context of that size (tables, factors, device buffers) and the persistent result array of solve_poisson."""
    W = np.asarray(W)
    N = W.shape[0]
    get_context(N)
    _out_buffer(N, np.complex64 if W.dtype == np.complex64 else np.complex128)


def _out_buffer(N, dtype):
    """solve_poisson returns the SAME ndarray on every call (cpu.py:24-32,726,734)."""
    key = (N, np.dtype(dtype).str)
    if key not in _out_cache:
        _out_cache[key] = np.zeros((N, N), dtype=dtype)
    return _out_cache[key]


def single_precision_on_device():
    """complex64 data is computed in float32 on the device, as the reference computes it (float32 tables and
    solve, cpu.py:725; complex64 products).  QUFLOW_HIP_C64=f64 restores the double-precision evaluation with a
    cast of the result (A/B runs)."""
    import os
    return os.environ.get("QUFLOW_HIP_C64", "f32") != "f64"


def laplacian(N, bc=False, dtype=np.float64):
    """Coefficient table (N,N,2) of the quantized Laplacian, quflow/laplacian/cpu.py:55-95,604-625.
    dtype=float32: the table the reference builds for complex64 input (the integer diagonal cast to float32,
    the double-precision square root rounded once, the boundary condition subtracted in float32)."""
    ctx = get_context(N)
    if np.dtype(dtype) == np.float32 and single_precision_on_device():
        lap32 = np.zeros((N, N, 2), dtype=np.float32)
        _lib.check(ctx._lib.qf_c64_laplacian_table(ctx.handle, int(bool(bc)), ptr(lap32)))
        return lap32
    lap = np.zeros((N, N, 2), dtype=np.float64)
    _lib.check(ctx._lib.qf_laplacian_table(ctx.handle, int(bool(bc)), ptr(lap)))
    return lap.astype(dtype, copy=False)


def solve_poisson(W, reduce=select_first):
    """Solve Delta P = W (quflow/laplacian/cpu.py:681-734).  The returned array is a
    persistent buffer that the caller may mutate and that the next call overwrites."""
    W = np.asarray(W)
    if W.ndim >= 3:
        W = reduce(W)
    in_dtype = W.dtype if W.dtype in (np.complex64, np.complex128) else np.complex128
    if in_dtype == np.complex64 and single_precision_on_device():
        # float32 tables, float32 Thomas solve, complex64 result (cpu.py:725-734)
        if W.ndim != 2 or W.shape[0] != W.shape[1]:
            raise ValueError("W must be a square matrix, got shape %s" % (W.shape,))
        W32 = np.ascontiguousarray(W, dtype=np.complex64)
        N = W32.shape[-1]
        ctx = get_context(N)
        P32 = _out_buffer(N, np.complex64)
        _lib.check(ctx._lib.qf_c64_solve_poisson(ctx.handle, ptr(W32), ptr(P32), int(_SKEW_HERM_)))
        return P32
    Wc = as_c128(W, "W")
    N = Wc.shape[-1]
    ctx = get_context(N)
    P = _out_buffer(N, np.complex128)
    _lib.check(ctx._lib.qf_solve_poisson(ctx.handle, ptr(Wc), ptr(P), int(_SKEW_HERM_)))
    if in_dtype == np.complex64:
        P32 = _out_buffer(N, np.complex64)
        P32[...] = P
        return P32
    return P


def laplace(P):
    """Apply the quantized Laplacian (quflow/laplacian/cpu.py:628-669, dense branch)."""
    if np.asarray(P).dtype == np.complex64 and single_precision_on_device():
        P32 = np.ascontiguousarray(P, dtype=np.complex64)
        if P32.ndim != 2 or P32.shape[0] != P32.shape[1]:
            raise ValueError("P must be a square matrix, got shape %s" % (P32.shape,))
        W32 = result_array(P32.shape, P32.dtype, "laplace")
        ctx = get_context(P32.shape[-1])
        _lib.check(ctx._lib.qf_c64_laplace(ctx.handle, ptr(P32), ptr(W32)))
        return W32
    Pc = as_c128(P, "P")
    N = Pc.shape[-1]
    ctx = get_context(N)
    W = result_array(Pc.shape, Pc.dtype, "laplace")      # (a new array per call unless the last one was dropped)
    _lib.check(ctx._lib.qf_laplace(ctx.handle, ptr(Pc), ptr(W)))
    return W.astype(np.asarray(P).dtype, copy=False) if np.asarray(P).dtype == np.complex64 else W


def _table_key(*parts):
    key = hash(parts) & 0xFFFFFFFFFFFFFFFF
    return key or 1


def _is_c64(W):
    return np.asarray(W).dtype == np.complex64 and single_precision_on_device()


def _solve_with_table(table, key, W):
    if _is_c64(W):
        # complex64 data: float32 table (built in float32 by the callers, as cpu.py:760,809 do with
        # dtype=type(W[0,0].real)) and the float32 solve
        W32 = np.ascontiguousarray(W, dtype=np.complex64)
        if W32.ndim != 2 or W32.shape[0] != W32.shape[1]:
            raise ValueError("W must be a square matrix, got shape %s" % (W32.shape,))
        ctx = get_context(W32.shape[-1])
        tab32 = np.ascontiguousarray(table, dtype=np.float32)
        P32 = np.zeros_like(W32)
        _lib.check(ctx._lib.qf_c64_solve_tridiagonal(ctx.handle, ptr(tab32), ptr(W32), ptr(P32), int(_SKEW_HERM_)))
        return P32
    Wc = as_c128(W, "W")
    N = Wc.shape[-1]
    ctx = get_context(N)
    table = np.ascontiguousarray(table, dtype=np.float64)
    P = np.zeros_like(Wc)
    _lib.check(ctx._lib.qf_solve_tridiagonal(ctx.handle, ptr(table), ctypes.c_ulonglong(key), ptr(Wc), ptr(P),
                                             int(_SKEW_HERM_)))
    return P


class _LRU(collections.OrderedDict):
    """Host tables are 16 N^2 bytes each (16 MiB at N=1024): keep the few a run alternates between
    (two half steps of a Strang splitting, a couple of sizes), not one per step size ever seen."""

    def __init__(self, maxlen=8):
        super().__init__()
        self.maxlen = maxlen

    def lookup(self, key, build):
        if key in self:
            self.move_to_end(key)
            return self[key]
        val = build()
        self[key] = val
        while len(self) > self.maxlen:
            self.popitem(last=False)
        return val


_table_cache = _LRU()
_plain_table_cache = _LRU(4)


def _shifted_table(N, c0, c1, dtype=np.float64):
    """c0*I - c1*Delta as an (N,N,2) table: the heat/helmholtz/viscdamp operators (cpu.py:765-769), in the
    arithmetic of `dtype` (float32 for complex64 data: the reference's `lap.copy()` is a float32 array then)."""
    dtype = np.dtype(dtype)
    def build():
        lap = _plain_table_cache.lookup((N, dtype.str), lambda: laplacian(N, bc=False, dtype=dtype))
        tab = lap.copy()
        tab[:, :, 0] = c0
        tab[:, :, 1] = 0.0
        tab -= c1 * lap
        return tab
    return _table_cache.lookup((N, float(c0), float(c1), dtype.str), build)


def _real_dtype(W):
    return np.float32 if _is_c64(W) else np.float64


def solve_helmholtz(W, alpha=1.0):
    """(1 - alpha Delta) P = W, quflow/laplacian/cpu.py:784-826."""
    N = np.asarray(W).shape[-1]
    return _solve_with_table(_shifted_table(N, 1.0, alpha, _real_dtype(W)), _table_key("helm", N, float(alpha)), W)


def solve_heat(h_times_nu, W0):
    """(1 - h nu Delta) W = W0, quflow/laplacian/cpu.py:737-781."""
    N = np.asarray(W0).shape[-1]
    return _solve_with_table(_shifted_table(N, 1.0, h_times_nu, _real_dtype(W0)), _table_key("helm", N, float(h_times_nu)), W0)


_globalqg_cache = _LRU(4)


def solve_globalqg(W, gamma=1.0):
    """Delta P + gamma Z P Z = W, quflow/laplacian/cpu.py:829-877: the Laplacian table with
    (gamma/2)(z_i^2 + z_j^2) taken off its diagonal coefficient, z = hbar*(-s..s) the diagonal of
    the third Cartesian generator (geometry.py:132-151,173-194); same device Thomas kernel."""
    N = np.asarray(W).shape[-1]
    def build():
        s = (N - 1) / 2
        zvec = _geometry.hbar(N) * np.arange(-s, s + 1)
        tab = laplacian(N, bc=False, dtype=_real_dtype(W)).copy()
        tab[:, :, 0] -= (gamma / 2.0) * zvec ** 2
        tab[:, :, 0] -= (gamma / 2.0) * zvec[:, np.newaxis] ** 2
        return tab
    return _solve_with_table(_globalqg_cache.lookup((N, float(gamma), np.dtype(_real_dtype(W)).str), build),
                             _table_key("gqg", N, float(gamma)), W)


def solve_viscdamp(h, W0, nu=1e-4, alpha=0.01, force=None, theta=1):
    """Theta scheme for W' - nu Delta W + alpha W = F, quflow/laplacian/cpu.py:880-943."""
    W0 = np.asarray(W0)
    N = W0.shape[-1]
    tab = _shifted_table(N, 1.0 + h * alpha * theta, h * nu * theta, _real_dtype(W0))
    if theta == 1:
        Wrhs = W0.copy()
    else:
        Wrhs = (1.0 - alpha * h * (1 - theta)) * W0
        Wrhs += (nu * h * (1 - theta)) * laplace(W0)
    if force is not None:
        Wrhs += h * force
    return _solve_with_table(tab, _table_key("visc", N, float(h), float(nu), float(alpha), float(theta)), Wrhs)


class ViscDampStep:
    """`strang_splitting=ViscDampStep(nu, alpha)`: the viscous / damped half step
    `lambda h, W: solve_viscdamp(h, W, nu, alpha)` of the reference's forced-turbulence runs
    (cpu.py:880-943, theta = 1, no force) as an object the device stepper recognises: between device
    steps it is applied to the resident state (no PCIe).  Called directly it is that lambda."""

    def __init__(self, nu=1e-4, alpha=0.01):
        self.nu = float(nu)
        self.alpha = float(alpha)

    def __call__(self, h, W):
        return solve_viscdamp(h, W, nu=self.nu, alpha=self.alpha)

    def table_and_key(self, N, h):
        """The (N,N,2) table of 1 + h alpha - h nu Delta and its cache key."""
        return (_shifted_table(N, 1.0 + h * self.alpha, h * self.nu),
                _table_key("visc", N, float(h), self.nu, self.alpha, 1.0))

    def apply_resident(self, ctx, h):
        """W <- (1 + h alpha - h nu Delta)^-1 W on the context's state."""
        tab, key = self.table_and_key(ctx.N, h)
        _lib.check(ctx._lib.qf_solve_tridiagonal(ctx.handle, ptr(np.ascontiguousarray(tab)), ctypes.c_ulonglong(key),
                                                 None, None, int(_SKEW_HERM_)))


class PoissonHIP:
    """Device Poisson operator with the constructor/call shape of the reference's
    DiagTriDiagOp (quflow/experimental/cuda.py:166-189, quflow/simulation.py:554-562):
    `PoissonHIP(N, dtype)(P_out, W_in)`; also usable as `hamiltonian(W) -> P`."""

    __name__ = "quhip"

    def __init__(self, N, dtype=np.complex128, device=None):
        self.N = int(N)
        self.dtype = np.dtype(dtype)
        self.ctx = get_context(self.N, device)

    def __call__(self, *args):
        if len(args) == 1:
            return solve_poisson(args[0])
        P_out, W_in = args
        P_out[...] = solve_poisson(W_in)
        return None

    solve_poisson = staticmethod(solve_poisson)
    laplace = staticmethod(laplace)
    select_skewherm = staticmethod(select_skewherm)
