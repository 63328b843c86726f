"""Isospectral midpoint stepper of quflow on the MI355X.

`isomp` keeps the stepper signature of quflow.integrators.isomp
(= isomp_fixedpoint, quflow/integrators/isospectral.py:338-353,617) so that
`quflow.simulation.solve(..., integrator=quflow_amd.isomp)` (simulation.py:788) and
the reference's notebooks drive it unchanged.  `IsompHIP(N, dtype)` has the
constructor/call shape of the reference's device precedent `IsompCUDA`
(quflow/experimental/isospectral_cuda.py:52-80,120-137) for the runfile selection of
quflow/simulation.py:554-562.

The whole call -- Poisson solves, both complex GEMMs, the fused commutator epilogue,
residual norms and the W update -- runs in hand-written HIP kernels behind one C-ABI
call (qf_isomp).  There is no CPU path: without the library or a GPU this raises.
"""
import ctypes

import numpy as np

from . import _lib
from . import laplacian as _laplacian
from .context import Context, default_device, get_context, ptr
from .geometry import hbar


def _is_native_hamiltonian(h):
    """The built-in Hamiltonian P = Delta^-1 W: ours, a PoissonHIP, or the reference's own
    default `quflow.laplacian.cpu.solve_poisson`, which simulation.solve injects when the
    user gives none (quflow/simulation.py:728-729)."""
    if h is None or h is _laplacian.solve_poisson or isinstance(h, _laplacian.PoissonHIP):
        return True
    mod = getattr(h, "__module__", "") or ""
    return getattr(h, "__name__", "") == "solve_poisson" and (
        mod.startswith("quflow.laplacian") or mod.startswith("quflow_amd.laplacian"))


def isomp_fixedpoint(W,
                     dt,
                     steps=100,
                     hamiltonian=_laplacian.solve_poisson,
                     time=None,
                     forcing=None,
                     strang_splitting=None,
                     stats=None,
                     callback=None,
                     tol='auto',
                     maxit=10,
                     minit=1,
                     verbatim=False,
                     compsum=False,
                     reinitialize=False,
                     device=None):
    """Isospectral midpoint method with fixed-point iterations for skew-Hermitian W
    (quflow/integrators/isospectral.py:338-613).  `W` (host ndarray, complex128 (N,N)) is
    overwritten and returned, like the reference (isospectral.py:361-362,592,613).

    Everything stays on the device for hamiltonian = solve_poisson (the default) with tol, maxit,
    minit, compsum, reinitialize, stats, verbatim, time (autonomous: ignored, as the reference
    does for a Hamiltonian without a `time` argument, isospectral.py:416-423).

    The host hooks of the reference run as host hooks here too:
      * `strang_splitting(dt/2, W)` (isospectral.py:466-467, 598-599) and `callback(W, dW)`
        (:549-550): the device steps one step at a time (qf_isomp, then qf_isomp_continue so that
        the iteration vector carries over as inside one reference call), the state crosses PCIe
        around each hook.  The callback gets host copies: changing them does not change the step.
      * `forcing(P, W[, time])` (:512-520, 591-595) and a foreign `hamiltonian(W[, time])`
        (:488-492) are needed inside every fixed-point iteration: the reference's loop runs on the
        host with the two matrix products (and the built-in Hamiltonian, if that is the one) on the
        device -- a fallback at the reference's own elementwise speed, not the fused device path.
    """
    # Check input (AssertionError like isospectral.py:400-401)
    assert minit >= 1, "minit must be at least 1."
    assert maxit >= minit, "maxit must be at minit."

    native = _is_native_hamiltonian(hamiltonian)
    if forcing is not None or not native:
        if isinstance(W, np.ndarray) and W.ndim != 2:
            raise NotImplementedError("forcing / foreign Hamiltonians with batched (k,N,N) states are not "
                                      "implemented on the HIP path yet.")
        return _isomp_host_loop(W, dt, steps, hamiltonian, native, time, forcing, strang_splitting, stats, callback,
                                tol, maxit, minit, verbatim, compsum, reinitialize, device)
    if strang_splitting is not None or callback is not None:
        if isinstance(W, np.ndarray) and W.ndim != 2:
            raise NotImplementedError("strang_splitting / callback with batched (k,N,N) states are not "
                                      "implemented on the HIP path yet.")
        return _isomp_stepwise(W, dt, steps, strang_splitting, stats, callback, tol, maxit, minit, verbatim,
                               compsum, reinitialize, device)

    if not isinstance(W, np.ndarray):
        raise TypeError("W must be a numpy ndarray")
    if W.ndim == 3 and W.shape[-1] == W.shape[-2]:
        # a stack of states: P from state 0, the exit test on state 0 (isospectral.py:527-532)
        if compsum:
            raise NotImplementedError("compsum with batched (k,N,N) states is not implemented on the HIP path yet.")
        return _isomp_states(W, dt, steps, tol, minit, maxit, reinitialize, False, stats, verbatim, device,
                             tol_key='tol_auto', maxit_key='number_of_maxit')
    if W.ndim != 2 or W.shape[0] != W.shape[1]:
        raise ValueError("W must be a square matrix or a (k,N,N) stack")
    N = W.shape[-1]
    ctx = get_context(N, device)

    if isinstance(tol, str):
        if tol != 'auto':
            raise ValueError("tol must be a float or 'auto'")
        tol_c = -1.0
    else:
        tol_c = float(tol)           # negative => auto (isospectral.py:440)
    auto = tol_c < 0

    Wc = np.ascontiguousarray(W, dtype=np.complex128)
    _lib.check(ctx._lib.qf_upload_W(ctx.handle, ptr(Wc)))
    st = _lib.IsompStats()
    _lib.check(ctx._lib.qf_isomp(ctx.handle, float(dt), int(steps), tol_c, int(minit), int(maxit),
                                 int(bool(compsum)), int(bool(reinitialize)), ctypes.byref(st)))
    _lib.check(ctx._lib.qf_download_W(ctx.handle, ptr(Wc)))
    if Wc is not W:
        W[...] = Wc                  # in-place contract

    if auto:
        if verbatim:
            print("Tolerance set to {}.".format(st.tol_used))
        if stats:
            stats['tol_auto'] = st.tol_used               # isospectral.py:451-452
    if verbatim and steps > 0:
        print("Average number of iterations per step: {:.2f}".format(st.total_iterations / steps))
    if stats and steps > 0:                               # isospectral.py:609-611
        stats["iterations"] = st.total_iterations / steps
        stats["number_of_maxit"] = st.number_of_maxit / steps
    return W


def _auto_tol(W, dt, compsum):
    """isospectral.py:440-452."""
    mach_eps = np.finfo(np.float64).eps
    if not compsum:
        mach_eps = np.sqrt(mach_eps)
    return (mach_eps * dt / hbar(W.shape[-1])) * np.linalg.norm(W, np.inf)


def _isomp_stepwise(W, dt, steps, strang_splitting, stats, callback, tol, maxit, minit, verbatim, compsum,
                    reinitialize, device):
    """strang_splitting / callback around device steps (built-in Hamiltonian): one step per
    qf_isomp / qf_isomp_continue call, the tolerance fixed once from the initial state as in the
    reference (isospectral.py:440-452)."""
    if not isinstance(W, np.ndarray) or W.ndim != 2 or W.shape[0] != W.shape[1]:
        raise ValueError("W must be a square matrix")
    N = W.shape[-1]
    ctx = get_context(N, device)
    Wc = np.ascontiguousarray(W, dtype=np.complex128)
    auto = isinstance(tol, str) or tol < 0
    if isinstance(tol, str) and tol != 'auto':
        raise ValueError("tol must be a float or 'auto'")
    tol_c = _auto_tol(Wc, dt, compsum) if auto else float(tol)
    if auto:
        if verbatim:
            print("Tolerance set to {}.".format(tol_c))
        if stats:
            stats['tol_auto'] = tol_c
    total_iterations = 0.0
    number_of_maxit = 0.0
    st = _lib.IsompStats()
    PW = np.zeros((N, N), dtype=np.complex128) if callback is not None else None
    on_device = False
    resident_strang = isinstance(strang_splitting, _laplacian.ViscDampStep)
    for k in range(steps):
        if resident_strang:                       # the half step on the resident state
            if not on_device:
                _lib.check(ctx._lib.qf_upload_W(ctx.handle, ptr(Wc)))
                on_device = True
            strang_splitting.apply_resident(ctx, dt / 2)
            if callback is not None:
                _lib.check(ctx._lib.qf_download_W(ctx.handle, ptr(Wc)))
        elif strang_splitting:
            if on_device:
                _lib.check(ctx._lib.qf_download_W(ctx.handle, ptr(Wc)))
            Wc = np.ascontiguousarray(strang_splitting(dt / 2, Wc), dtype=np.complex128)
            on_device = False
        elif callback is not None and on_device:
            _lib.check(ctx._lib.qf_download_W(ctx.handle, ptr(Wc)))      # the callback's W: before the update
        if not on_device:
            _lib.check(ctx._lib.qf_upload_W(ctx.handle, ptr(Wc)))
            on_device = True
        step = ctx._lib.qf_isomp if k == 0 else ctx._lib.qf_isomp_continue
        _lib.check(step(ctx.handle, float(dt), 1, float(tol_c), int(minit), int(maxit), int(bool(compsum)),
                        int(bool(reinitialize)), ctypes.byref(st)))
        total_iterations += st.total_iterations
        number_of_maxit += st.number_of_maxit
        if callback is not None:
            # PWcomm of the last iteration, doubled (isospectral.py:547-550), from the device's PW
            _lib.check(ctx._lib.qf_download_buffer(ctx.handle, _lib.BUFFER_IDS["PW"], ptr(PW)))
            callback(Wc.copy(), 2.0 * (PW - PW.conj().T))
        if resident_strang:
            strang_splitting.apply_resident(ctx, dt / 2)
        elif strang_splitting:
            _lib.check(ctx._lib.qf_download_W(ctx.handle, ptr(Wc)))
            Wc = np.ascontiguousarray(strang_splitting(dt / 2, Wc), dtype=np.complex128)
            on_device = False
    if on_device:
        _lib.check(ctx._lib.qf_download_W(ctx.handle, ptr(Wc)))
    W[...] = Wc
    if verbatim and steps > 0:
        print("Average number of iterations per step: {:.2f}".format(total_iterations / steps))
    if stats and steps > 0:
        stats["iterations"] = total_iterations / steps
        stats["number_of_maxit"] = number_of_maxit / steps
    return W


def _isomp_host_loop(W, dt, steps, hamiltonian, native, time, forcing, strang_splitting, stats, callback, tol,
                     maxit, minit, verbatim, compsum, reinitialize, device):
    """The reference's loop (isospectral.py:403-613, 2-D skew-Hermitian branch) on the host, for the
    hooks that act inside an iteration (forcing, a foreign Hamiltonian); the two matrix products
    (:496, :499) and the built-in Hamiltonian run on the device (qf_zgemm, qf_solve_poisson)."""
    if not isinstance(W, np.ndarray) or W.ndim != 2 or W.shape[0] != W.shape[1]:
        raise ValueError("W must be a square matrix")
    if isinstance(tol, str) and tol != 'auto':
        raise ValueError("tol must be a float or 'auto'")
    N = W.shape[-1]
    ctx = get_context(N, device)
    W = np.ascontiguousarray(W, dtype=np.complex128) if W.dtype != np.complex128 or not W.flags.c_contiguous else W
    W_in = W
    if hamiltonian is None or native:
        hamiltonian = _laplacian.solve_poisson

    def matmul(A, B, out):
        _lib.check(ctx._lib.qf_zgemm(ctx.handle, ptr(np.ascontiguousarray(A)), ptr(np.ascontiguousarray(B)), ptr(out)))
        return out

    if forcing is not None:                                        # :404-413
        autonomous_force = True
        if time is not None:
            try:
                FW = forcing(W, W, time=time)
            except TypeError:
                pass
            else:
                autonomous_force = False
        FW = np.zeros_like(W)
    autonomous = True                                              # :416-423
    if time is not None:
        try:
            Phalf = hamiltonian(W, time=time)
        except TypeError:
            pass
        else:
            autonomous = False
    total_iterations = 0
    number_of_maxit = 0
    dW = np.zeros_like(W)                                          # :430-437
    dW_old = np.zeros_like(W)
    Whalf = np.zeros_like(W)
    PWcomm = np.zeros_like(W)
    hb = hbar(N)
    vareps = dt / (2 * hb)
    if isinstance(tol, str) or tol < 0:                            # :440-452
        tol = _auto_tol(W, dt, compsum)
        if verbatim:
            print("Tolerance set to {}.".format(tol))
        if stats:
            stats['tol_auto'] = tol
    if compsum:                                                    # :455-459
        c_compsum = np.zeros_like(W)
    for k in range(steps):                                         # :463
        if strang_splitting:
            W = strang_splitting(dt / 2, W)
        resnorm = np.inf
        if reinitialize:
            dW.fill(0.0)
        for i in range(maxit):                                     # :475
            total_iterations += 1
            np.copyto(Whalf, W)
            Whalf += dW
            np.copyto(dW_old, dW)
            if autonomous:
                Phalf = hamiltonian(Whalf)
            else:
                Phalf = hamiltonian(Whalf, time=time + dt / 2)
            Phalf = np.ascontiguousarray(Phalf, dtype=np.complex128)
            Phalf = Phalf * vareps                                 # (a copy: the device Hamiltonian returns a cached buffer)
            matmul(Phalf, Whalf, PWcomm)                           # :496
            matmul(PWcomm, Phalf, dW)                              # :499
            if _laplacian._SKEW_HERM_:                             # :500-505
                PWcomm -= PWcomm.conj().T
            else:
                PWcomm -= matmul(Whalf, Phalf, np.zeros_like(W))
            dW += PWcomm                                           # :509
            if forcing:                                            # :512-520
                Phalf /= vareps
                if autonomous_force:
                    FW = forcing(Phalf, Whalf)
                else:
                    FW = forcing(Phalf, Whalf, time=time + dt / 2)
                FW = FW * (dt / 2)
                dW += FW
            if i + 1 >= minit:                                     # :523-536
                resnorm_old = resnorm
                dW_old -= dW
                resnorm = np.abs(dW_old).sum(axis=1).max()
                if resnorm <= tol or resnorm >= resnorm_old:
                    break
        else:
            number_of_maxit += 1
            if verbatim:
                print("Max iterations {} reached at step {}.".format(maxit, k))
        PWcomm *= 2                                                # :547
        if callback is not None:
            callback(W, PWcomm)
        if compsum:                                                # :553-589
            y = PWcomm - c_compsum
            t = W + y
            c_compsum = (t - W) - y
            np.copyto(W, t)
            if forcing:
                raise NotImplementedError("Compensated sum with forcing is not yet implemented.")
        else:
            W += PWcomm                                            # :592
            if forcing:
                FW *= 2
                W += FW
        if time is not None:
            time += dt
        if strang_splitting:
            W = strang_splitting(dt / 2, W)
    if verbatim:
        print("Average number of iterations per step: {:.2f}".format(total_iterations / steps))
    if stats:
        stats["iterations"] = total_iterations / steps
        stats["number_of_maxit"] = number_of_maxit / steps
    if W is not W_in:
        W_in[...] = W
    return W_in


def _isomp_states(W, dt, steps, tol, minit, maxit, reinitialize, magnetic, stats, verbatim, device,
                  tol_key, maxit_key):
    """(k,N,N) isomp / magmp through qf_isomp_states; W overwritten and returned."""
    if isinstance(tol, str):
        if tol != 'auto':
            raise ValueError("tol must be a float or 'auto'")
        tol_c = -1.0
    else:
        tol_c = float(tol)
    k, N = W.shape[0], W.shape[-1]
    ctx = get_context(N, device)
    Wc = np.ascontiguousarray(W, dtype=np.complex128)
    st = _lib.IsompStats()
    _lib.check(ctx._lib.qf_isomp_states(ctx.handle, ptr(Wc), int(k), float(dt), int(steps), tol_c, int(minit),
                                        int(maxit), int(bool(reinitialize)), int(bool(magnetic)), ctypes.byref(st)))
    if Wc is not W:
        W[...] = Wc
    if tol_c < 0:
        if verbatim:
            print("Tolerance set to {}.".format(st.tol_used))
        if stats:
            stats[tol_key] = st.tol_used
    if verbatim and steps > 0:
        print("Average number of iterations per step: {:.2f}".format(st.total_iterations / steps))
    if stats and steps > 0:
        stats["iterations"] = st.total_iterations / steps
        stats[maxit_key] = st.number_of_maxit / steps
    return W


# Default isospectral method (isospectral.py:617)
isomp = isomp_fixedpoint


# -------------------------------------------------
# MHD   (quflow/integrators/mhd.py)
# -------------------------------------------------

def solve_mhd(state):
    """Hamiltonian of the standard MHD system, quflow/integrators/mhd.py:10-18:
    state = (W, Theta) -> (P, B) = (Delta^-1 W, Delta Theta), both on the device."""
    W = state[0, :, :]
    Theta = state[1, :, :]
    P = _laplacian.solve_poisson(W)
    B = _laplacian.laplace(Theta)
    return P, B


def magmp_fixedpoint(W, dt, steps=100, hamiltonian=solve_mhd, time=None, forcing=None, stats=None,
                     callback=None, tol='auto', maxit=10, minit=1, verbatim=False, reinitialize=False,
                     device=None):
    """Magnetic isospectral midpoint method for the MHD system
    W' = [P, W] + [B, Theta],  Theta' = [P, Theta]  (quflow/integrators/mhd.py:235-456);
    `W` is the (2,N,N) state (W, Theta), overwritten and returned.  All six products of an
    iteration, the Poisson solve, the Laplacian and the updates run on the device (qf_isomp_states
    with magnetic=1).  stats receives 'tol', 'iterations', 'maxit' like the reference (:341,452-454).
    """
    assert minit >= 1, "minit must be at least 1."
    assert maxit >= minit, "maxit must be at minit."
    if forcing is not None:
        raise NotImplementedError("forcing is not implemented on the HIP path yet.")
    if callback is not None:
        raise NotImplementedError("callback is not implemented on the HIP path yet.")
    if hamiltonian is not solve_mhd and not (getattr(hamiltonian, "__name__", "") == "solve_mhd" and
                                             (getattr(hamiltonian, "__module__", "") or "").startswith("quflow")):
        raise NotImplementedError("only hamiltonian=solve_mhd runs on the HIP path.")
    if not isinstance(W, np.ndarray) or W.ndim != 3 or W.shape[0] != 2 or W.shape[1] != W.shape[2]:
        raise ValueError("the MHD state must be a (2,N,N) ndarray (W, Theta)")
    return _isomp_states(W, dt, steps, tol, minit, maxit, reinitialize, True, stats, verbatim, device,
                         tol_key='tol', maxit_key='maxit')


magmp = magmp_fixedpoint


# -------------------------------------------------
# OTHER ISOSPECTRAL METHODS   (quflow/integrators/isospectral.py:155-335)
# -------------------------------------------------

def _check_device_stepper_args(W, hamiltonian, forcing):
    if forcing is not None:
        raise NotImplementedError("forcing is not implemented on the HIP path yet.")
    if not _is_native_hamiltonian(hamiltonian):
        raise NotImplementedError("only hamiltonian=solve_poisson runs on the HIP path.")
    if not _laplacian._SKEW_HERM_:
        raise NotImplementedError("the HIP path of this stepper is for skew-Hermitian matrices "
                                  "(select_skewherm(True)).")
    if not isinstance(W, np.ndarray):
        raise TypeError("W must be a numpy ndarray")
    if W.ndim != 2 or W.shape[0] != W.shape[1]:
        raise ValueError("W must be a square matrix")


def isomp_quasinewton(W, dt, steps=100, hamiltonian=_laplacian.solve_poisson, forcing=None,
                      tol="auto", maxit=10, verbatim=False, **kwargs):
    """Isospectral midpoint method with the quasi-Newton iteration of
    quflow/integrators/isospectral.py:155-251; W is overwritten and returned.  The two linear
    solves per iteration with A = I - (stepsize/2) Ptilde run on the matrix cores (Newton-Schulz
    inverse, include/quflow_hip.h) instead of LAPACK's LU: same iteration, same result to
    rounding.  `stats` (optional keyword) receives iterations / number_of_maxit / tol."""
    _check_device_stepper_args(W, hamiltonian, forcing)
    if isinstance(tol, str):
        if tol != "auto":
            raise ValueError("tol must be a float or 'auto'")
        tol_c = -1.0
    else:
        tol_c = float(tol)
    ctx = get_context(W.shape[-1], kwargs.get("device"))
    Wc = np.ascontiguousarray(W, dtype=np.complex128)
    st = _lib.IsompStats()
    _lib.check(ctx._lib.qf_upload_W(ctx.handle, ptr(Wc)))
    _lib.check(ctx._lib.qf_isomp_quasinewton(ctx.handle, float(dt), int(steps), tol_c, int(maxit), ctypes.byref(st)))
    _lib.check(ctx._lib.qf_download_W(ctx.handle, ptr(Wc)))
    if Wc is not W:
        W[...] = Wc
    if verbatim and steps > 0:
        print("Average number of iterations per step: {:.2f}".format(st.total_iterations / steps))
    stats = kwargs.get("stats")
    if stats is not None and steps > 0:
        stats["iterations"] = st.total_iterations / steps
        stats["number_of_maxit"] = st.number_of_maxit / steps
        stats["tol"] = st.tol_used
    return W


def isomp_simple(W, dt, steps=100, hamiltonian=_laplacian.solve_poisson, forcing=None, **kwargs):
    """The simplified (explicit) isospectral midpoint method,
    quflow/integrators/isospectral.py:254-335; W is overwritten and returned."""
    _check_device_stepper_args(W, hamiltonian, forcing)
    ctx = get_context(W.shape[-1], kwargs.get("device"))
    Wc = np.ascontiguousarray(W, dtype=np.complex128)
    _lib.check(ctx._lib.qf_upload_W(ctx.handle, ptr(Wc)))
    _lib.check(ctx._lib.qf_isomp_simple(ctx.handle, float(dt), int(steps)))
    _lib.check(ctx._lib.qf_download_W(ctx.handle, ptr(Wc)))
    if Wc is not W:
        W[...] = Wc
    return W


# -------------------------------------------------
# CLASSICAL (EXPLICIT, NON-ISOSPECTRAL) INTEGRATORS   (quflow/integrators/erk.py)
# -------------------------------------------------

def update_stats(stats, **kwargs):
    """quflow/integrators/isospectral.py:85-90."""
    for arg, val in kwargs.items():
        if arg in stats and np.isscalar(val):
            stats[arg] += val
        else:
            stats[arg] = val


def _erk_host_loop(method, W, dt, steps, hamiltonian, forcing, device):
    """euler / heun / rk4 with `forcing` or a foreign Hamiltonian: the reference's loops
    (quflow/integrators/erk.py:47-56, 93-112, 142-160) on the host, the products of the bracket
    (geometry.py:41-49) -- and the built-in Hamiltonian, if that is the one -- on the device."""
    N = W.shape[-1]
    ctx = get_context(N, device)
    if _is_native_hamiltonian(hamiltonian):
        def ham(X):
            return _laplacian.solve_poisson(X).copy()      # (the device Hamiltonian returns a cached buffer)
    else:
        ham = hamiltonian
    hb = hbar(N)

    def bracket(P, X):
        P = np.ascontiguousarray(P, dtype=np.complex128)
        X = np.ascontiguousarray(X, dtype=np.complex128)
        A = np.zeros_like(X)
        B = np.zeros_like(X)
        _lib.check(ctx._lib.qf_zgemm(ctx.handle, ptr(P), ptr(X), ptr(A)))
        _lib.check(ctx._lib.qf_zgemm(ctx.handle, ptr(X), ptr(P), ptr(B)))
        A -= B
        A /= hb
        return A

    if forcing is None:
        rhs = bracket
    else:
        def rhs(P, X):
            return bracket(P, X) + forcing(P, X)
    for k in range(steps):
        if method == "euler":
            P = ham(W)
            W += dt * rhs(P, W)
        elif method == "heun":
            P = ham(W)
            F0 = rhs(P, W)
            Wprime = W + dt * F0
            P = ham(Wprime)
            F = rhs(P, Wprime)
            F += F0
            F *= dt / 2.0
            W += F
        else:
            P = ham(W)
            K1 = rhs(P, W)
            Wprime = W + (dt / 2.0) * K1
            P = ham(Wprime)
            K2 = rhs(P, Wprime)
            Wprime = W + (dt / 2.0) * K2
            P = ham(Wprime)
            K3 = rhs(P, Wprime)
            Wprime = W + dt * K3
            P = ham(Wprime)
            K4 = rhs(P, Wprime)
            W += (dt / 6.0) * (K1 + 2 * K2 + 2 * K3 + K4)
    return W


def _erk(method, W, dt, steps, hamiltonian, forcing, device=None):
    if not isinstance(W, np.ndarray):
        raise TypeError("W must be a numpy ndarray")
    if W.ndim != 2 or W.shape[0] != W.shape[1]:
        if W.ndim == 3:
            raise NotImplementedError("batched (k,N,N) states are not implemented on the HIP path yet.")
        raise ValueError("W must be a square matrix")
    if forcing is not None or not _is_native_hamiltonian(hamiltonian):
        if W.dtype != np.complex128:
            raise NotImplementedError("forcing / foreign Hamiltonians need a complex128 state on the HIP path.")
        return _erk_host_loop(method, W, dt, steps, hamiltonian, forcing, device)
    ctx = get_context(W.shape[-1], device)
    Wc = np.ascontiguousarray(W, dtype=np.complex128)
    _lib.check(ctx._lib.qf_upload_W(ctx.handle, ptr(Wc)))
    _lib.check(ctx._lib.qf_erk(ctx.handle, _lib.ERK_METHODS[method], float(dt), int(steps),
                               int(_laplacian._SKEW_HERM_)))
    _lib.check(ctx._lib.qf_download_W(ctx.handle, ptr(Wc)))
    if Wc is not W:
        W[...] = Wc                  # in-place contract (erk.py:56,110,156)
    return W


def euler(W, dt, steps=100, hamiltonian=_laplacian.solve_poisson, forcing=None, stats=None, **kwargs):
    """Euler's explicit first order method, quflow/integrators/erk.py:19-59; W is overwritten
    and returned.  The whole call (Poisson solves, products, updates) runs on the device; with
    `forcing` or a foreign Hamiltonian the reference's loop runs on the host around device products."""
    W = _erk("euler", W, dt, steps, hamiltonian, forcing, kwargs.get("device"))
    if stats is not None:
        update_stats(stats, steps=steps)          # erk.py:58-59
    return W


def heun(W, dt, steps=100, hamiltonian=_laplacian.solve_poisson, forcing=None, device=None):
    """Heun's second order method, quflow/integrators/erk.py:62-112."""
    return _erk("heun", W, dt, steps, hamiltonian, forcing, device)


def rk4(W, dt, steps=100, hamiltonian=_laplacian.solve_poisson, forcing=None, device=None):
    """The classical Runge-Kutta fourth order method, quflow/integrators/erk.py:115-160."""
    return _erk("rk4", W, dt, steps, hamiltonian, forcing, device)


class IsompHIP:
    """`IsompHIP(N, dtype)` pre-creates the device context (buffers, factor tables, stream)
    like IsompCUDA.__init__ (quflow/experimental/isospectral_cuda.py:52-80); calling it
    has the stepper signature.  Unlike IsompCUDA it updates W in place AND returns it."""

    def __init__(self, N, dtype=np.complex128, device=None):
        self.N = int(N)
        self.dtype = np.dtype(dtype)
        self.device = device
        self.ctx = get_context(self.N, device)

    def __call__(self, W, dt, steps=100, hamiltonian=_laplacian.solve_poisson, time=None, forcing=None,
                 strang_splitting=None, stats=None, callback=None, tol='auto', maxit=10, minit=1,
                 verbatim=False, compsum=False, reinitialize=False):
        if W.shape[-1] != self.N:
            raise ValueError("IsompHIP was built for N=%d, got W of size %d" % (self.N, W.shape[-1]))
        return isomp_fixedpoint(W, dt, steps=steps, hamiltonian=hamiltonian, time=time, forcing=forcing,
                                strang_splitting=strang_splitting, stats=stats, callback=callback, tol=tol,
                                maxit=maxit, minit=minit, verbatim=verbatim, compsum=compsum,
                                reinitialize=reinitialize, device=self.device)


class DeviceTrajectory:
    """Keeps one trajectory resident in HBM across chunks (no PCIe traffic between
    `advance` calls); used by bench.py and the ensemble driver.  Each `advance` has the
    semantics of one `integrator(W, dt, steps=...)` call of simulation.solve
    (quflow/simulation.py:782-798): dW restarts from zero (isospectral.py:430)."""

    def __init__(self, W0, device=None):
        W0 = np.ascontiguousarray(W0, dtype=np.complex128)
        self.N = W0.shape[-1]
        # a private context: the trajectory owns its device state (the shared per-N context of
        # get_context() is scratch for the host-in/host-out entry points)
        self.ctx = Context(self.N, default_device() if device is None else device)
        self._lib = self.ctx._lib
        _lib.check(self._lib.qf_upload_W(self.ctx.handle, ptr(W0)))

    def advance(self, dt, steps, tol='auto', maxit=10, minit=1, compsum=False, reinitialize=False):
        assert minit >= 1, "minit must be at least 1."
        assert maxit >= minit, "maxit must be at minit."
        tol_c = -1.0 if isinstance(tol, str) else float(tol)
        st = _lib.IsompStats()
        _lib.check(self._lib.qf_isomp(self.ctx.handle, float(dt), int(steps), tol_c, int(minit), int(maxit),
                                      int(bool(compsum)), int(bool(reinitialize)), ctypes.byref(st)))
        return {"iterations": st.total_iterations / max(steps, 1),
                "number_of_maxit": st.number_of_maxit / max(steps, 1),
                "total_iterations": st.total_iterations, "tol": st.tol_used,
                "last_resnorm": st.last_resnorm}

    def advance_erk(self, method, dt, steps):
        """`steps` steps of euler / heun / rk4 (quflow/integrators/erk.py) on the resident state."""
        _lib.check(self._lib.qf_erk(self.ctx.handle, _lib.ERK_METHODS[method], float(dt), int(steps),
                                    int(_laplacian._SKEW_HERM_)))
        evals = {"euler": 1, "heun": 2, "rk4": 4}[method]
        return {"iterations": float(evals), "number_of_maxit": 0.0, "total_iterations": evals * int(steps),
                "tol": 0.0, "last_resnorm": 0.0}

    def advance_lu(self, method, dt, steps, tol=-1.0, maxit=10):
        """`steps` steps of isomp_simple / isomp_quasinewton (isospectral.py:155-335) on the resident state."""
        st = _lib.IsompStats()
        if method == "isomp_simple":
            _lib.check(self._lib.qf_isomp_simple(self.ctx.handle, float(dt), int(steps)))
            st.total_iterations = int(steps)
        else:
            _lib.check(self._lib.qf_isomp_quasinewton(self.ctx.handle, float(dt), int(steps), float(tol), int(maxit),
                                                      ctypes.byref(st)))
        return {"iterations": st.total_iterations / max(steps, 1), "number_of_maxit": st.number_of_maxit / max(steps, 1),
                "total_iterations": st.total_iterations, "tol": st.tol_used, "last_resnorm": st.last_resnorm}

    def diagnostics(self):
        """(energy_euler, enstrophy) of the resident state, quflow/physics.py:26-38."""
        e = ctypes.c_double()
        s = ctypes.c_double()
        _lib.check(self._lib.qf_diagnostics(self.ctx.handle, ctypes.byref(e), ctypes.byref(s)))
        return e.value, s.value

    def _need_basis(self):
        if not getattr(self, "_basis_ready", False):
            _lib.check(self._lib.qf_basis_compute(self.ctx.handle))     # quantization.py:68-113, on the device
            self._basis_ready = True

    @classmethod
    def from_shr(cls, omega, N=-1, device=None):
        """Start a trajectory from real spherical-harmonics coefficients: W0 = shr2mat(omega, N)
        (quflow/quantization.py:450-489) is built straight into the resident state."""
        omega = np.ascontiguousarray(omega, dtype=np.float64)
        if N == -1:
            N = round(np.sqrt(omega.shape[0]))
        self = cls.__new__(cls)
        self.N = int(N)
        self.ctx = Context(self.N, default_device() if device is None else device)
        self._lib = self.ctx._lib
        self._need_basis()
        _lib.check(self._lib.qf_shr2mat(self.ctx.handle, ptr(omega), ctypes.c_longlong(omega.shape[0]), None))
        return self

    def shr(self, n_omega=None):
        """mat2shr of the resident state (quflow/quantization.py:492-525): what simulation.py:287-344
        stores for an 'shr' output -- N^2 doubles cross PCIe instead of the N^2 complex state."""
        self._need_basis()
        n = self.N * self.N if n_omega is None else int(n_omega)
        omega = np.zeros(n, dtype=np.float64)
        _lib.check(self._lib.qf_mat2shr(self.ctx.handle, None, ptr(omega), ctypes.c_longlong(n)))
        return omega

    def download(self):
        W = np.zeros((self.N, self.N), dtype=np.complex128)
        _lib.check(self._lib.qf_download_W(self.ctx.handle, ptr(W)))
        return W

    def sync(self):
        _lib.check(self._lib.qf_sync(self.ctx.handle))
