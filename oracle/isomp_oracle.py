"""CPU oracle for the hot path: numpy + oracle/libquflow_oracle.so.

TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
bench.py's `cpu_baseline` leg may import this module.  The HIP product path
(quflow_amd/) never imports it and has no CPU fallback.

It restates, function by function, the reference's CPU path:

    hbar              quflow/geometry.py:7-9
    laplacian         quflow/laplacian/cpu.py:55-95, 604-625
    solve_poisson     quflow/laplacian/cpu.py:281-362 (skewh), 200-278 (non-skewh), 681-734
    laplace           quflow/laplacian/cpu.py:98-108, 628-669
    conj_subtract_    quflow/integrators/isospectral.py:66-81
    isomp_fixedpoint  quflow/integrators/isospectral.py:338-613
    isomp_quasinewton, isomp_simple   quflow/integrators/isospectral.py:155-335
    solve_mhd, magmp_fixedpoint       quflow/integrators/mhd.py:10-18, 235-456
    bracket           quflow/geometry.py:41-49
    euler, heun, rk4  quflow/integrators/erk.py:19-160
    energy_euler, enstrophy, inner_L2   quflow/physics.py:26-38, quflow/geometry.py:72-76

The O(N^2) kernels live in quflow_oracle.c (plain C, OpenMP over diagonals = the
analogue of numba `prange`); the two complex GEMMs go through numpy's BLAS
exactly like the reference (`np.matmul(..., out=)`, isospectral.py:496,499);
norms use numpy/scipy like the reference (isospectral.py:448,534).

Parity status: PINNED by tests/test_oracle_vs_golden.py against fixtures
produced by running the reference itself (oracle/gen_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np
import scipy.linalg

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.path.join(_HERE, "libquflow_oracle.so")
_lib = None


def build(force=False):
    """Compile quflow_oracle.c with gcc (no GPU involved)."""
    if force or not os.path.exists(_LIBPATH) or (
            os.path.getmtime(_LIBPATH) < os.path.getmtime(os.path.join(_HERE, "quflow_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libquflow_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIBPATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIBPATH)
        dp = ctypes.POINTER(ctypes.c_double)
        L.qfo_hbar.restype = ctypes.c_double
        L.qfo_hbar.argtypes = [ctypes.c_int]
        L.qfo_laplacian.argtypes = [ctypes.c_int, ctypes.c_int, dp]
        for name in ("qfo_solve_skewh", "qfo_solve_nonskewh"):
            getattr(L, name).argtypes = [ctypes.c_int, dp, dp, dp, dp, dp]
        L.qfo_laplace.argtypes = [ctypes.c_int, dp, dp, dp]
        L.qfo_conj_subtract.argtypes = [ctypes.c_int, dp]
        L.qfo_norm_inf.restype = ctypes.c_double
        L.qfo_norm_inf.argtypes = [ctypes.c_int, dp]
        fp = ctypes.POINTER(ctypes.c_float)
        L.qfo_laplacian_f32.argtypes = [ctypes.c_int, ctypes.c_int, fp]
        L.qfo_solve_skewh_f32.argtypes = [ctypes.c_int, fp, fp, fp, fp, fp]
        L.qfo_laplace_f32.argtypes = [ctypes.c_int, fp, fp, fp]
        L.qfo_max_threads.restype = ctypes.c_int
        L.qfo_set_threads.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def set_threads(n):
    lib().qfo_set_threads(int(n))


def max_threads():
    return lib().qfo_max_threads()


# ---------------------------------------------------------------- geometry
def hbar(N):
    """quflow/geometry.py:7-9"""
    return 2.0 / np.sqrt(N ** 2 - 1)


def inner_L2(P, W):
    """quflow/geometry.py:72-76"""
    N = W.shape[-1]
    return (P * W.conj()).sum().real / N


# ---------------------------------------------------------------- laplacian
_lap_cache = {}
_buf_cache = {}
_SKEWH = True


def select_skewherm(flag):
    """quflow/laplacian/cpu.py:563-591 (returns the previous flag)."""
    global _SKEWH
    old = _SKEWH
    _SKEWH = bool(flag)
    return old


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def laplacian(N, bc=False, dtype=np.float64):
    """quflow/laplacian/cpu.py:604-625 (cached table, shape (N,N,2)); dtype=float32: the table the
    reference builds for complex64 input (cpu.py:725)."""
    if np.dtype(dtype) == np.float32:
        key = (N, bool(bc), "f32")
        if key not in _lap_cache:
            lap = np.zeros((N, N, 2), dtype=np.float32)
            lib().qfo_laplacian_f32(N, int(bool(bc)), _fp(lap))
            _lap_cache[key] = lap
        return _lap_cache[key]
    key = (N, bool(bc))
    if key not in _lap_cache:
        lap = np.zeros((N, N, 2), dtype=np.float64)
        lib().qfo_laplacian(N, int(bool(bc)), _dp(lap))
        _lap_cache[key] = lap
    return _lap_cache[key]


def _bufs(N):
    if N not in _buf_cache:
        _buf_cache[N] = (np.zeros((N, N), dtype=np.complex128),   # P (persistent output)
                         np.zeros((N, N), dtype=np.float64),
                         np.zeros((N, N), dtype=np.complex128))
    return _buf_cache[N]


def solve_poisson(W):
    """quflow/laplacian/cpu.py:681-734.  Returns the SAME buffer on every call
    (cpu.py:726,734); batched input uses state 0 (cpu.py:672-674,696-697)."""
    if W.ndim >= 3:
        W = np.ascontiguousarray(W[(0,) * (W.ndim - 2) + (Ellipsis,)])
    if W.dtype == np.complex64:
        # float32 tables and arithmetic, complex64 result (cpu.py:725-734); skew-Hermitian branch only
        W = np.ascontiguousarray(W)
        N = W.shape[-1]
        lap = laplacian(N, bc=True, dtype=np.float32)
        key = (N, "f32")
        if key not in _buf_cache:
            _buf_cache[key] = (np.zeros((N, N), dtype=np.complex64), np.zeros((N, N), dtype=np.float32),
                               np.zeros((N, N), dtype=np.complex64))
        P, bf, bcx = _buf_cache[key]
        if not _SKEWH:
            raise NotImplementedError("complex64 oracle: skew-Hermitian branch only")
        lib().qfo_solve_skewh_f32(N, _fp(lap), _fp(W), _fp(P), _fp(bf), _fp(bcx))
        return P
    W = np.ascontiguousarray(W, dtype=np.complex128)
    N = W.shape[-1]
    lap = laplacian(N, bc=True)
    P, bf, bcx = _bufs(N)
    fn = lib().qfo_solve_skewh if _SKEWH else lib().qfo_solve_nonskewh
    fn(N, _dp(lap), _dp(W), _dp(P), _dp(bf), _dp(bcx))
    return P


def solve_with_table(lap, W):
    """`_solve_cpu(lap, W, P, ...)` with an arbitrary coefficient table -- the
    heat / helmholtz / viscdamp solves of cpu.py:737-943."""
    W = np.ascontiguousarray(W, dtype=np.complex128)
    N = W.shape[-1]
    P = np.zeros_like(W)
    bf = np.zeros((N, N))
    bcx = np.zeros_like(W)
    fn = lib().qfo_solve_skewh if _SKEWH else lib().qfo_solve_nonskewh
    fn(N, _dp(np.ascontiguousarray(lap)), _dp(W), _dp(P), _dp(bf), _dp(bcx))
    return P


def laplace(P):
    """quflow/laplacian/cpu.py:628-669 (dense branch)."""
    if np.asarray(P).dtype == np.complex64:
        P = np.ascontiguousarray(P)
        N = P.shape[-1]
        W = np.zeros_like(P)
        lib().qfo_laplace_f32(N, _fp(laplacian(N, bc=False, dtype=np.float32)), _fp(P), _fp(W))
        return W
    P = np.ascontiguousarray(P, dtype=np.complex128)
    N = P.shape[-1]
    lap = laplacian(N, bc=False)
    W = np.zeros_like(P)
    lib().qfo_laplace(N, _dp(lap), _dp(P), _dp(W))
    return W


def solve_helmholtz(W, alpha=1.0):
    """quflow/laplacian/cpu.py:784-826: (1 - alpha Delta) P = W."""
    N = W.shape[-1]
    lap = laplacian(N, bc=False)
    tab = lap.copy()
    tab[:, :, 0] = 1.0
    tab[:, :, 1] = 0.0
    tab -= alpha * lap
    return solve_with_table(tab, W)


def solve_heat(h_times_nu, W0):
    """quflow/laplacian/cpu.py:737-781: (1 - h nu Delta) W = W0."""
    return solve_helmholtz(W0, alpha=h_times_nu)


def solve_globalqg(W, gamma=1.0):
    """quflow/laplacian/cpu.py:829-877: Delta P + gamma Z P Z = W (Z = diag of the third
    Cartesian generator, geometry.py:132-151,173-194)."""
    N = W.shape[-1]
    s = (N - 1) / 2
    zvec = hbar(N) * np.arange(-s, s + 1)
    tab = laplacian(N, bc=False).copy()
    tab[:, :, 0] -= (gamma / 2.0) * zvec ** 2
    tab[:, :, 0] -= (gamma / 2.0) * zvec[:, np.newaxis] ** 2
    return solve_with_table(tab, W)


def solve_viscdamp(h, W0, nu=1e-4, alpha=0.01, force=None, theta=1):
    """quflow/laplacian/cpu.py:880-943 (theta scheme)."""
    N = W0.shape[-1]
    lap = laplacian(N, bc=False)
    tab = lap.copy()
    tab[:, :, 0] = 1.0 + h * alpha * theta
    tab[:, :, 1] = 0.0
    tab -= (h * nu * theta) * lap
    if theta == 1:
        Wrhs = W0.copy()
    else:
        Wrhs = (1.0 - alpha * h * (1 - theta)) * W0
        Wrhs += (nu * h * (1 - theta)) * laplace(W0)
    if force is not None:
        Wrhs += h * force
    return solve_with_table(tab, Wrhs)


# ---------------------------------------------------------------- physics
def energy_euler(W):
    """quflow/physics.py:26-32"""
    P = solve_poisson(W)
    return -inner_L2(W, P) / 2.0


def enstrophy(W):
    """quflow/physics.py:34-38"""
    return inner_L2(W, W) / 2.0


# ---------------------------------------------------------------- integrator
def conj_subtract_(a):
    """quflow/integrators/isospectral.py:66-81 with out aliasing a (call site :503)."""
    if a.dtype == np.complex64:
        # one subtraction per entry, the mirror entry its exact negative conjugate: a - a^H elementwise is the
        # reference's loop bit for bit (complex64 arithmetic)
        a[...] = a - np.conj(np.swapaxes(a, -1, -2))
        return
    assert a.flags.c_contiguous and a.dtype == np.complex128
    if a.ndim == 2:
        lib().qfo_conj_subtract(a.shape[-1], _dp(a))
    else:
        for k in range(a.shape[0]):
            lib().qfo_conj_subtract(a.shape[-1], _dp(a[k]))


def isomp_fixedpoint(W, dt, steps=100, hamiltonian=solve_poisson, time=None, forcing=None,
                     strang_splitting=None, stats=None, callback=None, tol='auto', maxit=10,
                     minit=1, verbatim=False, compsum=False, reinitialize=False):
    """quflow/integrators/isospectral.py:338-613, statement by statement
    (skew-Hermitian branch; forcing / strang_splitting hooks included)."""
    assert minit >= 1, "minit must be at least 1."
    assert maxit >= minit, "maxit must be at minit."

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
    hb = hbar(N=W.shape[-1])
    vareps = dt / (2 * hb)

    if (isinstance(tol, str) and tol == 'auto') or (not isinstance(tol, str) and tol < 0):   # :440-452
        mach_eps = np.finfo(W.dtype).eps
        if not compsum:
            mach_eps = np.sqrt(mach_eps)
        if W.ndim > 2:
            zeroind = (0,) * (W.ndim - 2) + (Ellipsis,)
            tol = (mach_eps * dt / hb) * np.linalg.norm(W[zeroind], np.inf)
        else:
            tol = (mach_eps * dt / hb) * np.linalg.norm(W, np.inf)
        if verbatim:
            print("Tolerance set to {}.".format(tol))
        if stats:
            stats['tol_auto'] = tol

    if compsum:                                                    # :455-459
        y_compsum = np.zeros_like(W)
        c_compsum = np.zeros_like(W)
        t_compsum = np.zeros_like(W)
        delta_compsum = np.zeros_like(W)

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
            Phalf *= vareps                                        # :492
            np.matmul(Phalf, Whalf, out=PWcomm)                    # :496
            np.matmul(PWcomm, Phalf, out=dW)                       # :499
            conj_subtract_(PWcomm)                                 # :503
            dW += PWcomm                                           # :509
            if forcing:                                            # :512-520
                Phalf /= vareps
                if autonomous_force:
                    FW = forcing(Phalf, Whalf)
                else:
                    FW = forcing(Phalf, Whalf, time=time + dt / 2)
                FW *= dt / 2
                dW += FW
            if i + 1 >= minit:                                     # :523-536
                resnorm_old = resnorm
                dW_old -= dW
                if dW_old.ndim > 2:
                    resnormvec = scipy.linalg.norm(dW_old, ord=np.inf, axis=(-1, -2))
                    if Phalf.ndim == 2:
                        resnorm = resnormvec[0]
                    else:
                        resnorm = resnormvec.max()
                else:
                    resnorm = scipy.linalg.norm(dW_old, ord=np.inf)
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
            np.copyto(y_compsum, PWcomm)
            y_compsum -= c_compsum
            np.copyto(t_compsum, W)
            t_compsum += y_compsum
            np.copyto(delta_compsum, t_compsum)
            delta_compsum -= W
            np.copyto(c_compsum, delta_compsum)
            c_compsum -= y_compsum
            np.copyto(W, t_compsum)
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
    if stats:                                                      # :609-611
        stats["iterations"] = total_iterations / steps
        stats["number_of_maxit"] = number_of_maxit / steps
    return W


isomp = isomp_fixedpoint


# ---------------------------------------------------------------- synthetic inputs
# -------------------------------------------------------------- MHD (mhd.py)
def solve_mhd(state):
    """quflow/integrators/mhd.py:10-18."""
    P = solve_poisson(state[0, :, :])
    B = laplace(state[1, :, :])
    return P, B


def _conj_subtract_stack(a):
    if a.ndim == 2:
        return conj_subtract_(a)
    for j in range(a.shape[0]):
        conj_subtract_(a[j])
    return a


def magmp_fixedpoint(W, dt, steps=100, hamiltonian=None, time=None, forcing=None, stats=None, callback=None,
                     tol='auto', maxit=10, minit=1, reinitialize=False):
    """quflow/integrators/mhd.py:235-456, statement by statement (hamiltonian=None: solve_mhd)."""
    assert minit >= 1, "minit must be at least 1."
    assert maxit >= minit, "maxit must be at minit."
    if hamiltonian is None:
        hamiltonian = solve_mhd
    if forcing is not None:                                        # :296-304
        autonomous_force = True
        if time is not None:
            try:
                FW = forcing(W, W, time=time)
            except TypeError:
                pass
            else:
                autonomous_force = False
        FW = np.zeros_like(W)
    autonomous = True                                              # :307-314
    if time is not None:
        try:
            Phalf = hamiltonian(W, time=time)
        except TypeError:
            pass
        else:
            autonomous = False
    total_iterations = 0
    number_of_maxit = 0
    dW = np.zeros_like(W)
    dW_old = np.zeros_like(W)
    Whalf = np.zeros_like(W)
    PWcomm = np.zeros_like(W)
    BThetacomm = np.zeros_like(W[0, :, :])
    BThetaPhalf = np.zeros_like(BThetacomm)
    hb = hbar(N=W.shape[-1])
    vareps = dt / (2 * hb)
    if (isinstance(tol, str) and tol == 'auto') or (not isinstance(tol, str) and tol < 0):
        mach_eps = np.sqrt(np.finfo(W.dtype).eps)
        tol = (mach_eps * dt / hb) * np.linalg.norm(W[0], np.inf)
        if stats:
            stats['tol'] = tol
    for k in range(steps):
        resnorm = np.inf
        if reinitialize:
            dW.fill(0.0)
        for i in range(maxit):
            total_iterations += 1
            np.copyto(Whalf, W)
            Whalf += dW
            Thetahalf = Whalf[1, :, :]
            np.copyto(dW_old, dW)
            if autonomous:                                         # :371-376
                Phalf, Bhalf = hamiltonian(Whalf)
            else:
                Phalf, Bhalf = hamiltonian(Whalf, time=time + dt / 2)
            Phalf = Phalf * vareps
            Bhalf = Bhalf * vareps
            np.matmul(Phalf, Whalf, out=PWcomm)
            np.matmul(Bhalf, Thetahalf, out=BThetacomm)
            np.matmul(PWcomm, Phalf, out=dW)
            np.matmul(BThetacomm, Phalf, out=BThetaPhalf)
            _conj_subtract_stack(PWcomm)
            conj_subtract_(BThetacomm)
            dW += PWcomm
            dW[0, :, :] += BThetaPhalf
            dW[0, :, :] -= BThetaPhalf.T.conj()
            dW[0, :, :] += BThetacomm
            if forcing:                                            # :395-402
                Phalf = Phalf / vareps
                if autonomous_force:
                    FW = forcing(Phalf, Whalf)
                else:
                    FW = forcing(Phalf, Whalf, time=time + dt / 2)
                FW = FW * (dt / 2)
                dW += FW
            if i + 1 >= minit:
                resnorm_old = resnorm
                dW_old -= dW
                resnorm = scipy.linalg.norm(dW_old, ord=np.inf, axis=(-1, -2))[0]
                if resnorm <= tol or resnorm >= resnorm_old:
                    break
        else:
            number_of_maxit += 1
        PWcomm *= 2
        BThetacomm *= 2
        if callback is not None:                                   # :427-428
            callback(W, PWcomm)
        W += PWcomm
        W[0, :, :] += BThetacomm
        if forcing:                                                # :433-435
            FW = FW * 2
            W += FW
        if time is not None:
            time += dt
    if stats:
        stats["iterations"] = total_iterations / steps
        stats["maxit"] = number_of_maxit / steps
    return W


# -------------------------------------------------------------- LU-based isospectral steppers
def isomp_quasinewton(W, dt, steps=100, hamiltonian=None, tol="auto", maxit=10, stats=None):
    """quflow/integrators/isospectral.py:155-251 (skew-Hermitian, forcing=None), LAPACK LU as
    in the reference."""
    hamiltonian = hamiltonian or solve_poisson
    stepsize = dt / hbar(N=W.shape[-1])
    if tol == "auto" or tol < 0:
        tol = np.finfo(W.dtype).eps * stepsize * np.linalg.norm(W, np.inf)
    Id = np.eye(W.shape[0])
    Wtilde = W.copy()
    total_iterations = 0
    for k in range(steps):
        for i in range(maxit):
            total_iterations += 1
            Ptilde = hamiltonian(Wtilde)
            A = Id - (stepsize / 2.0) * Ptilde
            luA, piv = scipy.linalg.lu_factor(A)
            B = scipy.linalg.lu_solve((luA, piv), W)
            Wtilde_new = scipy.linalg.lu_solve((luA, piv), -B.conj().T)
            resnorm = scipy.linalg.norm(Wtilde - Wtilde_new, np.inf)
            Wtilde = Wtilde_new
            if resnorm < tol:
                break
        W_new = A.conj().T @ Wtilde @ A
        np.copyto(W, W_new)
    if stats is not None and steps > 0:
        stats["iterations"] = total_iterations / steps
    return W


def isomp_simple(W, dt, steps=100, hamiltonian=None, skewherm=True):
    """quflow/integrators/isospectral.py:254-335: the skew-Hermitian branch, and (skewherm=False) the
    general one of select_skewherm(False) (:303-314)."""
    hamiltonian = hamiltonian or solve_poisson
    Id = np.eye(W.shape[0])
    Wtilde = W.copy()
    stepsize = dt / hbar(W.shape[-1])
    for k in range(steps):
        Ptilde = hamiltonian(Wtilde)
        A = Id - (stepsize / 2.0) * Ptilde
        if skewherm:
            luA, piv = scipy.linalg.lu_factor(A)
            X = scipy.linalg.lu_solve((luA, piv), W)
            Wtilde = scipy.linalg.lu_solve((luA, piv), -X.conj().T)
            W_new = A.conj().T @ Wtilde @ A
        else:
            X = np.linalg.solve(A, W)
            Aalt = Id + (stepsize / 2.0) * Ptilde
            Wtilde = np.linalg.solve(Aalt.conj().T, X.conj().T).conj().T
            W_new = Aalt @ Wtilde @ A
        np.copyto(W, W_new)
    return W


# -------------------------------------------------------------- explicit steppers (erk.py)
def bracket(P, W):
    """quflow/geometry.py:41-49 (dense branch)."""
    A = P @ W
    A -= W @ P
    A /= hbar(P.shape[-1])
    return A


def _erk_rhs(forcing):
    """erk.py:47-52, 87-92, 136-141."""
    if forcing is None:
        return bracket
    return lambda P, W: bracket(P, W) + forcing(P, W)


def euler(W, dt, steps=100, hamiltonian=None, forcing=None):
    """quflow/integrators/erk.py:19-59."""
    hamiltonian = hamiltonian or solve_poisson
    rhs = _erk_rhs(forcing)
    for k in range(steps):
        P = hamiltonian(W)
        VF = rhs(P, W)
        W += dt * VF
    return W


def heun(W, dt, steps=100, hamiltonian=None, forcing=None):
    """quflow/integrators/erk.py:62-112."""
    hamiltonian = hamiltonian or solve_poisson
    rhs = _erk_rhs(forcing)
    for k in range(steps):
        P = hamiltonian(W)
        F0 = rhs(P, W)
        Wprime = W + dt * F0
        P = hamiltonian(Wprime)
        F = rhs(P, Wprime)
        F += F0
        F *= dt / 2.0
        W += F
    return W


def rk4(W, dt, steps=100, hamiltonian=None, forcing=None):
    """quflow/integrators/erk.py:115-160."""
    hamiltonian = hamiltonian or solve_poisson
    rhs = _erk_rhs(forcing)
    for k in range(steps):
        P = hamiltonian(W)
        K1 = rhs(P, W)
        Wprime = W + (dt / 2.0) * K1
        P = hamiltonian(Wprime)
        K2 = rhs(P, Wprime)
        Wprime = W + (dt / 2.0) * K2
        P = hamiltonian(Wprime)
        K3 = rhs(P, Wprime)
        Wprime = W + dt * K3
        P = hamiltonian(Wprime)
        K4 = rhs(P, Wprime)
        W += (dt / 6.0) * (K1 + 2 * K2 + 2 * K3 + K4)
    return W


def make_W0(N, seed):
    """Deterministic synthetic initial condition IC-A."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    W = A - A.conj().T
    W -= np.eye(N) * (np.trace(W) / N)
    W /= np.linalg.norm(W, "fro") / np.sqrt(N)
    return W


def make_W0_smooth(N, seed):
    """IC-B: normalize(solve_poisson(make_W0))."""
    WB = solve_poisson(make_W0(N, seed)).copy()
    WB /= np.linalg.norm(WB, "fro") / np.sqrt(N)
    return WB


def spectrum(W):
    return np.linalg.eigvalsh(1j * W)


def casimirs(W):
    """C_k = tr((iW)^k)/N, k = 2,3,4."""
    H = 1j * W
    N = W.shape[-1]
    H2 = H @ H
    return np.array([np.trace(H2).real / N, np.trace(H2 @ H).real / N,
                     np.trace(H2 @ H2).real / N])
