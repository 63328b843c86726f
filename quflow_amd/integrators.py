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
import operator

import numpy as np

from . import _lib
from . import laplacian as _laplacian
from .context import Context, default_device, get_context, get_stepper_context, ptr
from .geometry import hbar


_SKEW_HERM_ = True


def _device_matmul(A, B):
    from .geometry import _device_matmul as mm
    return mm(A, B)


def _device_commutator(W, P, skewherm):
    """qf_commutator: product(s) and the elementwise subtraction on the device, one PCIe round trip."""
    from . import _lib
    from .context import as_c128, get_context, ptr, result_array
    Wc = as_c128(W, "W")
    Pc = as_c128(P, "P")
    if Wc.shape != Pc.shape:
        raise ValueError("operands could not be broadcast together with shapes %s %s" % (Wc.shape, Pc.shape))
    C = result_array(Wc.shape, Wc.dtype, "commutator")
    ctx = get_context(Wc.shape[-1])
    _lib.check(ctx._lib.qf_commutator(ctx.handle, ptr(Wc), ptr(Pc), ptr(C), int(bool(skewherm))))
    return C


def commutator_generic(W, P):
    """W@P - P@W for arbitrary matrices (quflow/integrators/isospectral.py:22-36): both products and the subtraction
    on the device (the bits of the two products followed by numpy's `VF -= ...`)."""
    return _device_commutator(W, P, False)


def commutator_skewherm(W, P):
    """W@P - (W@P)^H: the commutator of skew-Hermitian matrices from ONE product
    (quflow/integrators/isospectral.py:39-54) -- the product and the conjugate subtraction the reference does on the host
    (`VF -= VF.conj().T`) both on the device: x - conj(y) is one exact negation and one rounding either way."""
    return _device_commutator(W, P, True)


# the default commutator (isospectral.py:57); select_skewherm switches it (:109-116)
commutator = commutator_skewherm


def project_skewherm(W):
    """In-place projection onto the skew-Hermitian matrices (isospectral.py:60-63)."""
    W /= 2.0
    W -= W.conj().T


def select_skewherm(flag):
    """quflow/integrators/isospectral.py:96-118: whether the integrators may assume skew-Hermitian
    matrices (commutator as PW - PW^H instead of PW - W@P); also switches the default `commutator`
    and the Laplacian backend."""
    global _SKEW_HERM_, commutator
    _SKEW_HERM_ = bool(flag)
    commutator = commutator_skewherm if flag else commutator_generic
    _laplacian.select_skewherm(flag)


def estimate_stepsize(W, P=None, safety_factor=0.1):
    """safety_factor * pi / lambda_max(P) with P = solve_poisson(W) (the device solve) unless given, and
    lambda_max the spectral norm (quflow/integrators/isospectral.py:121-148, geometry.norm_Linf).  The
    stepsize is dimension-free: delta_time = stepsize * hbar(N)."""
    from .geometry import norm_Linf
    if P is None:
        P = _laplacian.solve_poisson(W)
    lambda_max = norm_Linf(P)
    return safety_factor * np.pi / lambda_max


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

    The host hooks of the reference run as host hooks here too, around a trajectory that stays on the
    device (a context of its own: a hook may call solve_poisson, energy_euler, ... freely):
      * `strang_splitting(dt/2, W)` (isospectral.py:466-467, 598-599) and `callback(W, dW)`
        (:549-550) with the built-in Hamiltonian: the fused device stepper one step at a time (qf_isomp,
        then qf_isomp_continue: the iteration vector carries over as inside one reference call), the
        state crosses PCIe around each hook.  The callback gets host copies.
      * `forcing(P, W[, time])` (:512-520, 591-595), a foreign `hamiltonian(W[, time])` (:488-492), the
        general commutator of select_skewherm(False) (:504-505), and hooks / compsum on (k,N,N) stacks:
        qf_isomp_hooked -- W, dW, Whalf, the products and the Kahan term never leave the device; per
        iteration only what the hook reads goes down and what it returns comes up.
    """
    # Check input (AssertionError like isospectral.py:400-401)
    assert minit >= 1, "minit must be at least 1."
    assert maxit >= minit, "maxit must be at minit."

    native = _is_native_hamiltonian(hamiltonian)
    # `for k in range(steps)` and the in-place complex updates (isospectral.py:463, 481-482, 592): a float step count and a real
    # or integer W are TypeErrors, a negative count an empty loop -- a real W is not silently advanced and truncated here either
    steps = _reference_args(W, steps)
    if steps == 0 and not np.issubdtype(W.dtype, np.complexfloating):
        return W                 # the reference's empty loop never touches a real / integer W
    stacked = W.ndim == 3
    hooks = strang_splitting is not None or callback is not None
    if (forcing is not None or not native or not _SKEW_HERM_ or not _laplacian._SKEW_HERM_
            or (stacked and (hooks or compsum))):
        # hooks that act inside an iteration, the general (not skew-Hermitian) commutator, and hooks /
        # compsum on stacks: device-resident state with host hooks (qf_isomp_hooked)
        return _isomp_hooked(W, dt, steps, hamiltonian, native, time, forcing, strang_splitting, stats, callback,
                             tol, maxit, minit, verbatim, compsum, reinitialize, device)
    if hooks:
        return _isomp_stepwise(W, dt, steps, strang_splitting, stats, callback, tol, maxit, minit, verbatim,
                               compsum, reinitialize, device)

    if not isinstance(W, np.ndarray):
        raise TypeError("W must be a numpy ndarray")
    if W.ndim == 3 and W.shape[-1] == W.shape[-2]:
        # a stack of states: P from state 0, the exit test on state 0 (isospectral.py:527-532)
        return _isomp_states(W, dt, steps, tol, minit, maxit, reinitialize, False, stats, verbatim, device,
                             tol_key='tol_auto', maxit_key='number_of_maxit')
    if W.ndim != 2 or W.shape[0] != W.shape[1]:
        raise ValueError("W must be a square matrix or a (k,N,N) stack")
    N = W.shape[-1]
    ctx = get_context(N, device)

    auto = isinstance(tol, str) or tol < 0      # negative => auto (isospectral.py:440)
    tol_c, tol_report = _device_tol(W, dt, tol, compsum)

    st = _lib.IsompStats()
    if W.dtype == np.complex64 and _laplacian.single_precision_on_device():
        # complex64 data is advanced in single precision, as the reference does it: float32 Poisson solve,
        # complex64 products, float32 Kahan term; the tolerance above is the reference's float32 rule
        Wc = np.ascontiguousarray(W)
        _lib.check(ctx._lib.qf_c64_upload_W(ctx.handle, ptr(Wc)))
        _lib.check(ctx._lib.qf_c64_isomp(ctx.handle, float(dt), int(steps), tol_c, int(minit), int(maxit),
                                         int(bool(compsum)), int(bool(reinitialize)), ctypes.byref(st)))
        _lib.check(ctx._lib.qf_c64_download_W(ctx.handle, ptr(Wc)))
    else:
        Wc = np.ascontiguousarray(W, dtype=np.complex128)
        _lib.check(ctx._lib.qf_upload_W(ctx.handle, ptr(Wc)))
        _lib.check(ctx._lib.qf_isomp(ctx.handle, float(dt), int(steps), tol_c, int(minit), int(maxit),
                                     int(bool(compsum)), int(bool(reinitialize)), ctypes.byref(st)))
        _lib.check(ctx._lib.qf_download_W(ctx.handle, ptr(Wc)))
    if Wc is not W:
        W[...] = Wc                  # in-place contract

    if auto:
        tol_used = st.tol_used if tol_report is None else tol_report
        if verbatim:
            print("Tolerance set to {}.".format(tol_used))
        if stats:
            stats['tol_auto'] = tol_used                  # isospectral.py:451-452
    if verbatim and steps > 0:
        print("Average number of iterations per step: {:.2f}".format(st.total_iterations / steps))
    if stats and steps > 0:                               # isospectral.py:609-611
        stats["iterations"] = st.total_iterations / steps
        stats["number_of_maxit"] = st.number_of_maxit / steps
    return W


def _auto_tol(W, dt, compsum):
    """isospectral.py:440-452: the machine epsilon is that of W.dtype (complex64 input: float32)."""
    W0 = W[(0,) * (W.ndim - 2) + (Ellipsis,)] if W.ndim > 2 else W
    dtype = W.dtype if W.dtype == np.complex64 else np.float64
    mach_eps = np.finfo(dtype).eps
    if not compsum:
        mach_eps = np.sqrt(mach_eps)
    return (mach_eps * dt / hbar(W.shape[-1])) * np.linalg.norm(W0, np.inf)


def _device_tol(W, dt, tol, compsum):
    """(tol for the C entry, tolerance to report or None).  A negative tol asks the device for the
    automatic double-precision tolerance (read back as stats.tol_used); complex64 input gets the
    reference's single-precision tolerance, evaluated here from the input, so that a complex64 run stops
    where the reference's does although the arithmetic on the device is double precision."""
    if isinstance(tol, str):
        if tol != 'auto':
            raise TypeError("tol must be a float or 'auto' (the reference compares it with a number: '<' not supported between instances of 'str' and 'int')")
        tol = -1.0
    tol = float(tol)
    if tol < 0 and W.dtype == np.complex64:
        t = float(_auto_tol(W, dt, compsum))
        return t, t
    return tol, None


def _isomp_stepwise(W, dt, steps, strang_splitting, stats, callback, tol, maxit, minit, verbatim, compsum,
                    reinitialize, device):
    """strang_splitting / callback around device steps (built-in Hamiltonian): one step per
    qf_isomp / qf_isomp_continue call, the tolerance fixed once from the initial state as in the
    reference (isospectral.py:440-452)."""
    if not isinstance(W, np.ndarray) or W.ndim != 2 or W.shape[0] != W.shape[1]:
        raise ValueError("W must be a square matrix")
    N = W.shape[-1]
    if isinstance(tol, str) and tol != 'auto':
        raise TypeError("tol must be a float or 'auto' (the reference compares it with a number: '<' not supported between instances of 'str' and 'int')")
    ctx = get_stepper_context(N, device)      # the hooks may use the shared context (solve_viscdamp, energy_euler, ...)
    Wc = np.ascontiguousarray(W, dtype=np.complex128)
    auto = isinstance(tol, str) or tol < 0
    tol_c = float(_auto_tol(W, dt, compsum)) if auto else float(tol)
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


def _probe_time(fn, args, time):
    """The reference finds out whether a hook is time dependent by calling it once with `time=`
    (isospectral.py:403-423): a TypeError means autonomous.  Returns (takes_time, what that call returned or None)."""
    if time is None:
        return False, None
    try:
        out = fn(*args, time=time)
    except TypeError:
        return False, None
    return True, out


def _takes_time(fn, args, time):
    return _probe_time(fn, args, time)[0]


class _HookTable:
    """qf_isomp_hooks for a set of Python hooks: C callbacks over numpy views of the library's pinned
    staging matrices.  An exception raised by a hook is kept and re-raised once the C call has returned."""

    def __init__(self, N, k, squeeze):
        self.N, self.k, self.squeeze = N, k, squeeze
        self.error = None
        self.c = _lib.IsompHooks()
        self.c.skewh = int(bool(_SKEW_HERM_))
        self.c.solve_skewh = int(bool(_laplacian._SKEW_HERM_))
        self._keep = []

    def _view(self, p, k):
        buf = (ctypes.c_double * (2 * k * self.N * self.N)).from_address(p)
        a = np.frombuffer(buf, dtype=np.complex128).reshape(k, self.N, self.N)
        return a[0] if self.squeeze else a

    def _guard(self, body):
        def run(*args):
            try:
                body(*args)
                return 0
            except BaseException as exc:      # noqa: BLE001 -- carried across the C frame
                self.error = exc
                return 1
        return run

    def set_hamiltonian(self, fn, takes_time, per_state=False):
        """`per_state`: False = one (N,N) stream matrix for all states; True = one per state, a (k,N,N) array
        (np.matmul batches the products); None = a stack whose Hamiltonian has not said yet: qf_isomp_hooks::states_p
        goes out as -1 and the FIRST evaluation the stepper itself asks for settles it (the library reads the field
        back after that call) -- no entry probe, so the user's function is called exactly as often as the reference
        calls it, whatever runs before the first evaluation (Strang half step, a carried increment)."""
        state = {"per_state": per_state}
        def body(user, pW, pP, t):
            W = self._view(pW, self.k)
            P = np.asarray(fn(W, time=t) if takes_time else fn(W))
            if state["per_state"] is None:
                if P.shape not in ((self.N, self.N), (1, self.N, self.N), (self.k, self.N, self.N)):
                    raise ValueError("the Hamiltonian returned a %s array for a (%d,%d,%d) stack: neither one (N,N) stream matrix "
                                     "for all states nor one per state" % (P.shape, self.k, self.N, self.N))
                state["per_state"] = (P.shape == (self.k, self.N, self.N) and self.k > 1)
                self.c.states_p = int(state["per_state"])
            if P.shape == (1, self.N, self.N) and not state["per_state"]:
                P = P.reshape(self.N, self.N)       # (numpy broadcasts a (1,N,N) stream matrix over the stack)
            if state["per_state"]:
                if P.shape != (self.k, self.N, self.N):
                    raise ValueError("the Hamiltonian returned a %s array after a (%d,%d,%d) one" % (P.shape, self.k, self.N, self.N))
                np.frombuffer((ctypes.c_double * (2 * self.k * self.N * self.N)).from_address(pP),
                              dtype=np.complex128).reshape(self.k, self.N, self.N)[...] = P
                return
            if P.shape != (self.N, self.N):
                # (numpy raises ValueError where the reference multiplies or stores it: isospectral.py:496-509)
                raise ValueError("the Hamiltonian returned a %s array for (%d,%d) states: operands could not be broadcast "
                                 "together" % (P.shape, self.N, self.N))
            self._view(pP, 1).reshape(self.N, self.N)[...] = P
        self.c.states_p = -1 if per_state is None else int(bool(per_state))
        cb = _lib.HAMILTONIAN_CB(self._guard(body))
        self._keep.append(cb)
        self.c.hamiltonian = cb
        self.c.hamiltonian_takes_time = int(takes_time)

    def set_hamiltonian_pair(self, fn, takes_time):
        """magmp: `hamiltonian(state[, time])` returns the pair (P, B) (mhd.py:10-18, 371-374)."""
        def body(user, pW, pPB, t):
            state = self._view(pW, 2)
            P, B = fn(state, time=t) if takes_time else fn(state)
            out = self._view(pPB, 2)
            out[0][...] = P
            out[1][...] = B
        cb = _lib.HAMILTONIAN_CB(self._guard(body))
        self._keep.append(cb)
        self.c.hamiltonian = cb
        self.c.hamiltonian_takes_time = int(takes_time)

    def set_forcing(self, fn, takes_time):
        def body(user, pP, pW, pF, t):
            if self.c.states_p:       # the Hamiltonian returns one stream matrix per state: forcing sees them all
                P = np.frombuffer((ctypes.c_double * (2 * self.k * self.N * self.N)).from_address(pP),
                                  dtype=np.complex128).reshape(self.k, self.N, self.N)
            else:
                P = self._view(pP, 1).reshape(self.N, self.N)
            W = self._view(pW, self.k)
            F = fn(P, W, time=t) if takes_time else fn(P, W)
            self._view(pF, self.k)[...] = F
        cb = _lib.FORCING_CB(self._guard(body))
        self._keep.append(cb)
        self.c.forcing = cb
        self.c.forcing_takes_time = int(takes_time)

    def set_strang(self, fn):
        def body(user, h, pW):
            W = self._view(pW, self.k)
            W[...] = fn(h, W)
        cb = _lib.STRANG_CB(self._guard(body))
        self._keep.append(cb)
        self.c.strang = cb

    def set_strang_table(self, table, key):
        table = np.ascontiguousarray(table, dtype=np.float64)
        self._keep.append(table)
        self.c.strang_table = table.ctypes.data
        self.c.strang_key = key

    def set_callback(self, fn):
        def body(user, pW, pD):
            fn(self._view(pW, self.k), self._view(pD, self.k))
        cb = _lib.CALLBACK_CB(self._guard(body))
        self._keep.append(cb)
        self.c.callback = cb

    def check(self, rc):
        if self.error is not None:
            err, self.error = self.error, None
            raise err
        if rc == 6:          # QF_ERR_UNSUPPORTED: what the reference raises NotImplementedError for
            raise NotImplementedError(_lib.load().qf_last_error().decode("utf-8", "replace"))
        _lib.check(rc)


def _isomp_hooked(W, dt, steps, hamiltonian, native, time, forcing, strang_splitting, stats, callback, tol,
                  maxit, minit, verbatim, compsum, reinitialize, device):
    """isomp_fixedpoint with hooks inside the iteration (forcing, foreign Hamiltonian), the general
    commutator, or hooks / compsum on a stack of states: qf_isomp_hooked keeps the trajectory on the
    device and calls back for what only Python can compute."""
    if W.ndim not in (2, 3) or W.shape[-1] != W.shape[-2]:
        raise ValueError("W must be a square matrix or a (k,N,N) stack")
    if isinstance(tol, str) and tol != 'auto':
        raise TypeError("tol must be a float or 'auto' (the reference compares it with a number: '<' not supported between instances of 'str' and 'int')")
    N = W.shape[-1]
    squeeze = W.ndim == 2
    k = 1 if squeeze else W.shape[0]
    Wc = np.ascontiguousarray(W, dtype=np.complex128)
    table = _HookTable(N, k, squeeze)
    if forcing is not None:
        table.set_forcing(forcing, _takes_time(forcing, (Wc, Wc), time))
    if not native:
        # one stream matrix for all states or one per state?  Not asked here: the stepper's first evaluation tells
        # (set_hamiltonian, per_state=None), so the user's function is called as often as the reference calls it --
        # the autonomy probe with `time=` included (isospectral.py:416-423), with or without Strang splitting
        table.set_hamiltonian(hamiltonian, _takes_time(hamiltonian, (Wc,), time), per_state=(False if squeeze or k == 1 else None))
    if isinstance(strang_splitting, _laplacian.ViscDampStep):
        tab, key = strang_splitting.table_and_key(N, dt / 2)
        table.set_strang_table(tab, key)
    elif strang_splitting is not None:
        table.set_strang(strang_splitting)
    if callback is not None:
        table.set_callback(callback)
    table.c.has_time = int(time is not None)
    table.c.time = float(time) if time is not None else 0.0
    auto = isinstance(tol, str) or tol < 0
    ctx = get_stepper_context(N, device)
    st = _lib.IsompStats()
    tol_c, tol_report = _device_tol(W, dt, tol, compsum)
    rc = ctx._lib.qf_isomp_hooked(ctx.handle, ptr(Wc), k, float(dt), int(steps), tol_c, int(minit),
                                  int(maxit), int(bool(compsum)), int(bool(reinitialize)), ctypes.byref(table.c),
                                  ctypes.byref(st))
    table.check(rc)
    if Wc is not W:
        W[...] = Wc
    if auto:
        tol_used = st.tol_used if tol_report is None else tol_report
        if verbatim:
            print("Tolerance set to {}.".format(tol_used))
        if stats:
            stats['tol_auto'] = tol_used
    if verbatim and steps > 0:
        print("Average number of iterations per step: {:.2f}".format(st.total_iterations / steps))
    if stats and steps > 0:
        stats["iterations"] = st.total_iterations / steps
        stats["number_of_maxit"] = st.number_of_maxit / steps
    return W


def _magmp_hooked(W, dt, steps, hamiltonian, native_mhd, time, forcing, stats, callback, tol, maxit, minit, verbatim,
                  reinitialize, device):
    """magmp_fixedpoint with host hooks (quflow/integrators/mhd.py:235-456): qf_isomp_hooked in its magnetic mode --
    the (2,N,N) state, dW, the products and the magnetic terms stay on the device; a foreign Hamiltonian returns the
    pair (P, B), `forcing(P, state)` a (2,N,N) force, `callback(state, 2 PWcomm)` host copies."""
    if isinstance(tol, str) and tol != 'auto':
        raise TypeError("tol must be a float or 'auto' (the reference compares it with a number: '<' not supported between instances of 'str' and 'int')")
    if not (_SKEW_HERM_ and _laplacian._SKEW_HERM_):
        raise NotImplementedError("magmp on the HIP path is for skew-Hermitian matrices (select_skewherm(True)).")
    N = W.shape[-1]
    Wc = np.ascontiguousarray(W, dtype=np.complex128)
    table = _HookTable(N, 2, False)
    table.c.magnetic = 1
    if forcing is not None:
        table.set_forcing(forcing, _takes_time(forcing, (Wc, Wc), time))
    if not native_mhd:
        table.set_hamiltonian_pair(hamiltonian, _takes_time(hamiltonian, (Wc,), time))
    if callback is not None:
        table.set_callback(callback)
    table.c.has_time = int(time is not None)
    table.c.time = float(time) if time is not None else 0.0
    auto = isinstance(tol, str) or tol < 0
    tol_c = -1.0 if auto else float(tol)
    ctx = get_stepper_context(N, device)
    st = _lib.IsompStats()
    rc = ctx._lib.qf_isomp_hooked(ctx.handle, ptr(Wc), 2, float(dt), int(steps), tol_c, int(minit), int(maxit), 0,
                                  int(bool(reinitialize)), ctypes.byref(table.c), ctypes.byref(st))
    table.check(rc)
    if Wc is not W:
        W[...] = Wc
    if auto:
        if verbatim:
            print("Tolerance set to {}.".format(st.tol_used))
        if stats:
            stats['tol'] = st.tol_used                     # mhd.py:344-345
    if verbatim and steps > 0:
        print("Average number of iterations per step: {:.2f}".format(st.total_iterations / steps))
    if stats and steps > 0:                                # mhd.py:451-453
        stats["iterations"] = st.total_iterations / steps
        stats["maxit"] = st.number_of_maxit / steps
    return W


def _isomp_states(W, dt, steps, tol, minit, maxit, reinitialize, magnetic, stats, verbatim, device,
                  tol_key, maxit_key):
    """(k,N,N) isomp / magmp through qf_isomp_states; W overwritten and returned."""
    auto = isinstance(tol, str) or tol < 0
    tol_c, tol_report = _device_tol(W, dt, tol, False)
    k, N = W.shape[0], W.shape[-1]
    ctx = get_context(N, device)
    Wc = np.ascontiguousarray(W, dtype=np.complex128)
    st = _lib.IsompStats()
    _lib.check(ctx._lib.qf_isomp_states(ctx.handle, ptr(Wc), int(k), float(dt), int(steps), tol_c, int(minit),
                                        int(maxit), int(bool(reinitialize)), int(bool(magnetic)), ctypes.byref(st)))
    if Wc is not W:
        W[...] = Wc
    if auto:
        tol_used = st.tol_used if tol_report is None else tol_report
        if verbatim:
            print("Tolerance set to {}.".format(tol_used))
        if stats:
            stats[tol_key] = tol_used
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
    if not isinstance(W, np.ndarray) or W.ndim != 3 or W.shape[0] != 2 or W.shape[1] != W.shape[2]:
        raise ValueError("the MHD state must be a (2,N,N) ndarray (W, Theta)")
    steps = _reference_args(W, steps)         # (mhd.py: `for k in range(steps)`, in-place updates of the state)
    if steps == 0 and not np.issubdtype(W.dtype, np.complexfloating):
        return W                 # the reference's empty loop never touches a real / integer W
    native_mhd = hamiltonian is solve_mhd or (getattr(hamiltonian, "__name__", "") == "solve_mhd" and
                                              (getattr(hamiltonian, "__module__", "") or "").startswith("quflow"))
    if forcing is not None or callback is not None or not native_mhd:
        # forcing(P, state[, time]), callback(state, 2 PWcomm), a foreign hamiltonian(state[, time]) -> (P, B)
        # (mhd.py:296-314, 395-402, 427-428): the hooked device loop in its magnetic mode
        return _magmp_hooked(W, dt, steps, hamiltonian, native_mhd, time, forcing, stats, callback, tol, maxit, minit,
                             verbatim, reinitialize, device)
    return _isomp_states(W, dt, steps, tol, minit, maxit, reinitialize, True, stats, verbatim, device,
                         tol_key='tol', maxit_key='maxit')


magmp = magmp_fixedpoint


# -------------------------------------------------
# OTHER ISOSPECTRAL METHODS   (quflow/integrators/isospectral.py:155-335)
# -------------------------------------------------

def _reference_args(W, steps):
    """What every stepper of the reference does with its state and step count before anything else happens: `range(steps)` (a
    float is a TypeError, a negative count an empty loop) and in-place complex updates of W (numpy refuses them for a real or
    integer array: UFuncTypeError, a TypeError).  Returns the step count to run."""
    if not isinstance(W, np.ndarray):
        raise TypeError("W must be a numpy ndarray")
    steps = operator.index(steps)
    if steps <= 0:
        return 0                 # an empty loop: no in-place update is ever attempted, W comes back as it is (any dtype)
    if not np.issubdtype(W.dtype, np.complexfloating):
        raise TypeError("Cannot cast ufunc 'add' output from dtype('complex128') to dtype('%s') with casting rule 'same_kind'" % W.dtype)
    return steps


def _check_device_stepper_args(W, hamiltonian, forcing):
    if forcing is not None:
        # the reference accepts `forcing` here and never uses it (`assert NotImplementedError(...)` asserts a truthy
        # object: isospectral.py:185-186, 283-284); same result, but say so
        import warnings
        warnings.warn("isomp_simple / isomp_quasinewton ignore `forcing` (as the reference does: "
                      "quflow/integrators/isospectral.py:185-186, 283-284)", stacklevel=3)
    if not isinstance(W, np.ndarray):
        raise TypeError("W must be a numpy ndarray")
    if W.ndim != 2 or W.shape[0] != W.shape[1]:
        raise ValueError("W must be a square matrix")


def _lu_needs_hook_table(hamiltonian):
    """The plain entry points are the default case (built-in Hamiltonian, skew-Hermitian flags on); a foreign
    Hamiltonian or select_skewherm(False) goes through the hooked ones, whose table carries the two flags."""
    return (not _is_native_hamiltonian(hamiltonian)) or not (_SKEW_HERM_ and _laplacian._SKEW_HERM_)


def _lu_hook_table(N, hamiltonian):
    table = _HookTable(N, 1, True)
    if not _is_native_hamiltonian(hamiltonian):
        table.set_hamiltonian(hamiltonian, False)
    return table


def isomp_quasinewton(W, dt, steps=100, hamiltonian=_laplacian.solve_poisson, forcing=None,
                      tol="auto", maxit=10, verbatim=False, **kwargs):
    """Isospectral midpoint method with the quasi-Newton iteration of
    quflow/integrators/isospectral.py:155-251; W is overwritten and returned.  The two linear
    solves per iteration with A = I - (stepsize/2) Ptilde run on the matrix cores (Newton-Schulz
    inverse, include/quflow_hip.h) instead of LAPACK's LU: same iteration, same result to
    rounding.  `stats` (optional keyword) receives iterations / number_of_maxit / tol."""
    _check_device_stepper_args(W, hamiltonian, forcing)
    steps = _reference_args(W, steps)
    if steps == 0 and not np.issubdtype(W.dtype, np.complexfloating):
        return W                 # the reference's empty loop never touches a real / integer W
    if isinstance(tol, str):
        if tol != "auto":
            raise TypeError("tol must be a float or 'auto' (the reference compares it with a number: '<' not supported between instances of 'str' and 'int')")
        tol_c = -1.0
    else:
        tol_c = float(tol)
    if tol_c < 0 and W.dtype == np.complex64:
        # isospectral.py:194-195 with the machine epsilon of the input's precision
        tol_c = float(np.finfo(np.float32).eps * (dt / hbar(W.shape[-1])) * np.linalg.norm(W, np.inf))
    Wc = np.ascontiguousarray(W, dtype=np.complex128)
    st = _lib.IsompStats()
    if not _lu_needs_hook_table(hamiltonian):
        ctx = get_context(W.shape[-1], kwargs.get("device"))
        _lib.check(ctx._lib.qf_upload_W(ctx.handle, ptr(Wc)))
        _lib.check(ctx._lib.qf_isomp_quasinewton(ctx.handle, float(dt), int(steps), tol_c, int(maxit), ctypes.byref(st)))
    else:
        # a foreign Hamiltonian (isospectral.py:207): called back once per pass on host copies; the linear solves and
        # the update stay on the device.  A context of its own: the hook may use the shared one.  With
        # select_skewherm(False) the reference runs the very same formulas (its `assert NotImplementedError(...)`,
        # :188-189, asserts a truthy object) on the general branch of the Poisson solve.
        ctx = get_stepper_context(W.shape[-1], kwargs.get("device"))
        table = _lu_hook_table(W.shape[-1], hamiltonian)
        _lib.check(ctx._lib.qf_upload_W(ctx.handle, ptr(Wc)))
        table.check(ctx._lib.qf_isomp_quasinewton_hooked(ctx.handle, float(dt), int(steps), tol_c, int(maxit),
                                                         ctypes.byref(st), ctypes.byref(table.c)))
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
    steps = _reference_args(W, steps)
    if steps == 0 and not np.issubdtype(W.dtype, np.complexfloating):
        return W                 # the reference's empty loop never touches a real / integer W
    Wc = np.ascontiguousarray(W, dtype=np.complex128)
    if not _lu_needs_hook_table(hamiltonian):
        ctx = get_context(W.shape[-1], kwargs.get("device"))
        _lib.check(ctx._lib.qf_upload_W(ctx.handle, ptr(Wc)))
        _lib.check(ctx._lib.qf_isomp_simple(ctx.handle, float(dt), int(steps)))
    else:
        # a foreign Hamiltonian (isospectral.py:286) and / or select_skewherm(False): the general branch (:303-314)
        ctx = get_stepper_context(W.shape[-1], kwargs.get("device"))
        table = _lu_hook_table(W.shape[-1], hamiltonian)
        _lib.check(ctx._lib.qf_upload_W(ctx.handle, ptr(Wc)))
        table.check(ctx._lib.qf_isomp_simple_hooked(ctx.handle, float(dt), int(steps), ctypes.byref(table.c)))
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


def _erk_hooked(method, W, dt, steps, hamiltonian, forcing, device):
    """euler / heun / rk4 with `forcing(P, W)` or a foreign `hamiltonian(W)` (erk.py:47-56, 93-112,
    142-160): qf_erk_hooked keeps the state and the stage combinations on the device and calls back."""
    N = W.shape[-1]
    Wc = np.ascontiguousarray(W, dtype=np.complex128)
    table = _HookTable(N, 1, True)
    if forcing is not None:
        table.set_forcing(forcing, False)
    if not _is_native_hamiltonian(hamiltonian):
        table.set_hamiltonian(hamiltonian, False)
    ctx = get_stepper_context(N, device)
    rc = ctx._lib.qf_erk_hooked(ctx.handle, ptr(Wc), _lib.ERK_METHODS[method], float(dt), int(steps), ctypes.byref(table.c))
    table.check(rc)
    if Wc is not W:
        W[...] = Wc
    return W


def _erk(method, W, dt, steps, hamiltonian, forcing, device=None):
    steps = _reference_args(W, steps)         # (erk.py: `for k in range(steps)`, in-place updates of W)
    if steps == 0 and not np.issubdtype(W.dtype, np.complexfloating):
        return W                 # the reference's empty loop never touches a real / integer W
    if W.ndim == 3 and W.shape[1] == W.shape[2]:
        # a stack of states: P from state 0, bracket(P, W) broadcast over the stack (erk.py with (k,N,N) input)
        if forcing is not None or not _is_native_hamiltonian(hamiltonian):
            if W.dtype != np.complex128:
                raise NotImplementedError("forcing / foreign Hamiltonians need a complex128 state on the HIP path.")
            # hooks see the whole stack; a foreign Hamiltonian returns ONE (N,N) stream matrix (qf_erk_states_hooked)
            N, k = W.shape[-1], W.shape[0]
            Wc = np.ascontiguousarray(W, dtype=np.complex128)
            table = _HookTable(N, k, False)
            if forcing is not None:
                table.set_forcing(forcing, False)
            if not _is_native_hamiltonian(hamiltonian):
                # one stream matrix for all states or one per state: the first stage's evaluation tells
                table.set_hamiltonian(hamiltonian, False, per_state=(None if k > 1 else False))
            ctx = get_stepper_context(N, device)
            table.check(ctx._lib.qf_erk_states_hooked(ctx.handle, ptr(Wc), int(k), _lib.ERK_METHODS[method], float(dt), int(steps),
                                                      ctypes.byref(table.c)))
            if Wc is not W:
                W[...] = Wc
            return W
        ctx = get_context(W.shape[-1], device)
        Wc = np.ascontiguousarray(W, dtype=np.complex128)
        _lib.check(ctx._lib.qf_erk_states(ctx.handle, ptr(Wc), int(W.shape[0]), _lib.ERK_METHODS[method], float(dt),
                                          int(steps), int(_laplacian._SKEW_HERM_)))
        if Wc is not W:
            W[...] = Wc
        return W
    if W.ndim != 2 or W.shape[0] != W.shape[1]:
        raise ValueError("W must be a square matrix or a (k,N,N) stack")
    if forcing is not None or not _is_native_hamiltonian(hamiltonian):
        if W.dtype != np.complex128:
            raise NotImplementedError("forcing / foreign Hamiltonians need a complex128 state on the HIP path.")
        return _erk_hooked(method, W, dt, steps, hamiltonian, forcing, device)
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
    `forcing` or a foreign Hamiltonian the hooks are called back from the device-resident loop."""
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
        # a complex64 initial state makes a single-precision trajectory (float32 solve, complex64 products:
        # what the reference does with complex64 input); anything else is complex128
        self.c64 = np.asarray(W0).dtype == np.complex64 and _laplacian.single_precision_on_device()
        self.dtype = np.complex64 if self.c64 else np.complex128
        W0 = np.ascontiguousarray(W0, dtype=self.dtype)
        self.N = W0.shape[-1]
        # a private context: the trajectory owns its device state (the shared per-N context of
        # get_context() is scratch for the host-in/host-out entry points)
        self.ctx = Context(self.N, default_device() if device is None else device)
        self._lib = self.ctx._lib
        _lib.check((self._lib.qf_c64_upload_W if self.c64 else self._lib.qf_upload_W)(self.ctx.handle, ptr(W0)))

    def advance(self, dt, steps, tol='auto', maxit=10, minit=1, compsum=False, reinitialize=False, diagnostics=False):
        """`diagnostics=True`: energy_euler and enstrophy of the new state come back with the statistics
        (keys "energy", "enstrophy"), computed behind the last step under the call's one synchronisation
        (qf_isomp_diag) -- what an output chunk of simulation.solve logs (quflow/simulation.py:788-803)."""
        assert minit >= 1, "minit must be at least 1."
        assert maxit >= minit, "maxit must be at minit."
        tol_c = -1.0 if isinstance(tol, str) else float(tol)
        st = _lib.IsompStats()
        args = (self.ctx.handle, float(dt), int(steps), tol_c, int(minit), int(maxit), int(bool(compsum)),
                int(bool(reinitialize)), ctypes.byref(st))
        out = {}
        if self.c64:
            _lib.check(self._lib.qf_c64_isomp(*args))
            if diagnostics:
                e, s = self.diagnostics()
                out = {"energy": e, "enstrophy": s}
        elif diagnostics:
            e = ctypes.c_double()
            s = ctypes.c_double()
            _lib.check(self._lib.qf_isomp_diag(*args, ctypes.byref(e), ctypes.byref(s)))
            out = {"energy": e.value, "enstrophy": s.value}
        else:
            _lib.check(self._lib.qf_isomp(*args))
        out.update({"iterations": st.total_iterations / max(steps, 1),
                    "number_of_maxit": st.number_of_maxit / max(steps, 1),
                    "total_iterations": st.total_iterations, "tol": st.tol_used,
                    "last_resnorm": st.last_resnorm})
        return out

    def advance_erk(self, method, dt, steps):
        """`steps` steps of euler / heun / rk4 (quflow/integrators/erk.py) on the resident state."""
        self._double_only("advance_erk")
        _lib.check(self._lib.qf_erk(self.ctx.handle, _lib.ERK_METHODS[method], float(dt), int(steps),
                                    int(_laplacian._SKEW_HERM_)))
        evals = {"euler": 1, "heun": 2, "rk4": 4}[method]
        return {"iterations": float(evals), "number_of_maxit": 0.0, "total_iterations": evals * int(steps),
                "tol": 0.0, "last_resnorm": 0.0}

    def advance_lu(self, method, dt, steps, tol=-1.0, maxit=10):
        """`steps` steps of isomp_simple / isomp_quasinewton (isospectral.py:155-335) on the resident state."""
        self._double_only("advance_lu")
        st = _lib.IsompStats()
        if method == "isomp_simple":
            _lib.check(self._lib.qf_isomp_simple(self.ctx.handle, float(dt), int(steps)))
            st.total_iterations = int(steps)
        else:
            _lib.check(self._lib.qf_isomp_quasinewton(self.ctx.handle, float(dt), int(steps), float(tol), int(maxit),
                                                      ctypes.byref(st)))
        return {"iterations": st.total_iterations / max(steps, 1), "number_of_maxit": st.number_of_maxit / max(steps, 1),
                "total_iterations": st.total_iterations, "tol": st.tol_used, "last_resnorm": st.last_resnorm}

    def _double_only(self, what):
        if self.c64:
            raise NotImplementedError("%s on a complex64 trajectory: the resident single-precision state has the "
                                      "stepper (advance), diagnostics, upload and download; convert to complex128 "
                                      "for the rest" % what)

    def diagnostics(self):
        """(energy_euler, enstrophy) of the resident state, quflow/physics.py:26-38."""
        e = ctypes.c_double()
        s = ctypes.c_double()
        fn = self._lib.qf_c64_diagnostics if self.c64 else self._lib.qf_diagnostics
        _lib.check(fn(self.ctx.handle, ctypes.byref(e), ctypes.byref(s)))
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
        self.c64 = False
        self.dtype = np.complex128
        self.ctx = Context(self.N, default_device() if device is None else device)
        self._lib = self.ctx._lib
        self._need_basis()
        _lib.check(self._lib.qf_shr2mat(self.ctx.handle, ptr(omega), ctypes.c_longlong(omega.shape[0]), None))
        return self

    def shr(self, n_omega=None):
        """mat2shr of the resident state (quflow/quantization.py:492-525): what simulation.py:287-344
        stores for an 'shr' output -- N^2 doubles cross PCIe instead of the N^2 complex state."""
        self._double_only("shr")
        self._need_basis()
        n = self.N * self.N if n_omega is None else int(n_omega)
        omega = np.zeros(n, dtype=np.float64)
        _lib.check(self._lib.qf_mat2shr(self.ctx.handle, None, ptr(omega), ctypes.c_longlong(n)))
        return omega

    def download(self):
        W = np.zeros((self.N, self.N), dtype=self.dtype)
        _lib.check((self._lib.qf_c64_download_W if self.c64 else self._lib.qf_download_W)(self.ctx.handle, ptr(W)))
        return W

    def upload(self, W):
        """Replace the resident state (e.g. after a call that ended in an error left it undefined)."""
        W = np.ascontiguousarray(W, dtype=self.dtype)
        if W.shape != (self.N, self.N):
            raise ValueError("state must be (%d, %d), got %s" % (self.N, self.N, W.shape))
        _lib.check((self._lib.qf_c64_upload_W if self.c64 else self._lib.qf_upload_W)(self.ctx.handle, ptr(W)))

    def sync(self):
        _lib.check(self._lib.qf_sync(self.ctx.handle))


class DeviceEnsemble:
    """k independent trajectories resident on ONE GPU, advanced together (qf_isomp_multi): each
    member is a DeviceTrajectory of its own -- own Hamiltonian, own exit decisions, own statistics,
    results bit-identical to advancing it alone -- but one host loop feeds all their streams, so the
    GPU overlaps the replicas.  For ensembles with more seeds than GPUs (quflow_amd.ensemble)."""

    def __init__(self, W0s, device=None):
        self.members = [DeviceTrajectory(W0, device=device) for W0 in W0s]
        if not self.members:
            raise ValueError("DeviceEnsemble needs at least one initial condition")
        if len({m.N for m in self.members}) != 1:
            raise ValueError("the members of a DeviceEnsemble share one matrix size")
        if len({m.c64 for m in self.members}) != 1:
            raise ValueError("the members of a DeviceEnsemble share one dtype (complex128 or complex64)")
        self.c64 = self.members[0].c64
        self._lib = self.members[0]._lib

    def __len__(self):
        return len(self.members)

    # Members advanced at the same time: one per hardware pipe.  A stream's hardware queue number mod 4 is its pipe and two
    # replicas on one pipe lose 40 % of their combined rate (profiles/r06_x4_hardware_queues.txt: N = 512
    # sum 18,100 timesteps/s with four replicas, 13,000 with five, 16,100 with eight), so a larger ensemble goes through a
    # call in groups of four neighbours (created back to back: consecutive queues, four pipes).
    CONCURRENT = 4

    def advance(self, dt, steps, tol='auto', maxit=10, minit=1):
        """`steps` steps of every member (the semantics of one `integrator(W, dt, steps=...)` call each);
        returns one stats dict per member."""
        assert minit >= 1, "minit must be at least 1."
        assert maxit >= minit, "maxit must be at minit."
        tol_c = -1.0 if isinstance(tol, str) else float(tol)
        fn = self._lib.qf_c64_isomp_multi if self.c64 else self._lib.qf_isomp_multi
        out = []
        for g0 in range(0, len(self.members), self.CONCURRENT):
            group = self.members[g0:g0 + self.CONCURRENT]
            k = len(group)
            handles = (ctypes.c_void_p * k)(*[m.ctx.handle for m in group])
            st = (_lib.IsompStats * k)()
            _lib.check(fn(handles, k, float(dt), int(steps), tol_c, int(minit), int(maxit), st))
            out += [{"iterations": s.total_iterations / max(steps, 1), "number_of_maxit": s.number_of_maxit / max(steps, 1),
                     "total_iterations": s.total_iterations, "tol": s.tol_used, "last_resnorm": s.last_resnorm} for s in st]
        return out

    def diagnostics(self):
        return [m.diagnostics() for m in self.members]

    def download(self):
        return [m.download() for m in self.members]

    def sync(self):
        for m in self.members:
            m.sync()

    def close(self):
        for m in self.members:
            m.ctx.close()
