"""CPU oracle for the spherical-harmonics <-> matrix transforms.

TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/ may import this module.

Restates, in plain numpy (serial forms), the reference's

    basis_break_index        quflow/quantization.py:24-42
    compute_direct_laplacian quflow/laplacian/direct.py:19-62
    adjust_basis_orientation_ quflow/quantization.py:45-65
    compute_basis            quflow/quantization.py:68-113
    shr2mat_ / mat2shr_      quflow/quantization.py:130-183, 251-283 (serial), 188-245, 286-327 (parallel)
    shc2mat_ / mat2shc_      quflow/quantization.py:331-396
    shr2mat / mat2shr        quflow/quantization.py:450-525

Parity status: PINNED by tests/test_oracle_vs_golden.py::test_quantization_* against fixtures
produced by running the reference itself (oracle/gen_golden.py quantization).
"""
import numpy as np
from scipy.linalg import eigh_tridiagonal


def elm2ind(el, m):
    return el * el + el + m


def basis_break_index(absm, N):
    absm = absm - 1
    ind = absm + 2 * absm ** 2 - 6 * absm * N + 6 * N ** 2
    ind *= 1 + absm
    return ind // 6


def compute_direct_laplacian(N, bc=False):
    s = (N - 1) / 2
    mvals = np.linspace(-s, s, N)
    lap = np.zeros((2, N * (N + 1) // 2))
    for m1 in mvals:
        for m2 in mvals:
            coeff1 = 2 * (s * (s + 1) - m1 * m2)
            if abs(coeff1) > 1e-10:
                m = round(m1 - m2)
                if m < 0:
                    continue
                n = N - m
                ind = round(lap.shape[1] - n * (n + 1) // 2 + m2 + s)
                lap[1, ind] = -coeff1
            if m1 < s and m2 < s:
                coeff2 = -np.sqrt(s * (s + 1) - m1 * (m1 + 1)) * np.sqrt(s * (s + 1) - m2 * (m2 + 1))
                if abs(coeff2) > 1e-10:
                    m = round(m1 - m2)
                    if m < 0:
                        continue
                    n = N - m
                    ind = round(lap.shape[1] - n * (n + 1) // 2 + m2 + s + 1)
                    lap[0, ind] = -coeff2
    if bc:
        lap[1, 0] += 0.5
    return lap


def adjust_basis_orientation_(w2, m, tol=1e-16):
    for i in range(w2.shape[1]):
        val = w2[-1, i]
        if val < 0:
            w2[:, i] *= (-1) * (-1 if m % 2 == 1 else 1)
        elif val == 0.0:
            for j in range(2, w2.shape[0]):
                if np.abs(w2[-j, i]) > tol and np.abs(w2[-j - 1, i]) > tol:
                    prev_sign = np.sign(w2[-j - 1, i])
                    this_sign = np.sign(w2[-j, i])
                    if this_sign * prev_sign == -1:
                        w2[:, i] *= this_sign * (-1 if m % 2 == 1 else 1) * (-1 if j % 2 == 0 else 1)
                    else:
                        w2[:, i] *= this_sign * (-1 if m % 2 == 1 else 1)
                    break
        else:
            w2[:, i] *= (-1 if m % 2 == 1 else 1)


def compute_basis(N):
    basis = np.zeros(basis_break_index(N, N))
    lap = compute_direct_laplacian(N, bc=False)
    for m in range(N):
        n = N - m
        start_ind = N * (N + 1) // 2 - n * (n + 1) // 2
        end_ind = start_ind + n
        v2, w2 = eigh_tridiagonal(lap[1, start_ind:end_ind], lap[0, start_ind + 1:end_ind])
        w2 *= np.sqrt(N)
        w2 = w2[:, ::-1]
        adjust_basis_orientation_(w2, m)
        bind0 = basis_break_index(m, N)
        basis[bind0:bind0 + n * n] = w2.ravel()
    return basis


_cache = {}


def get_basis(N):
    if N not in _cache:
        _cache[N] = compute_basis(N)
    return _cache[N]


def _block(basis, m, N):
    bind0 = basis_break_index(m, N)
    return basis[bind0:bind0 + (N - m) ** 2].reshape((N - m, N - m))


def shr2mat_(omega, basis, W_out):
    """quantization.py:188-245 (band limit as in the default, parallel, form: int(sqrt))."""
    N = W_out.shape[-1]
    elmax = N - 1
    if omega.shape[0] < N ** 2:
        elmax = int(np.sqrt(omega.shape[0])) - 1
    Nmax = elmax + 1
    for m in range(Nmax):
        B = _block(basis, m, N)
        if m == 0:
            diag = B[:, :Nmax] @ omega[elm2ind(np.arange(0, Nmax), 0)].astype(complex)
            W_out[np.arange(N), np.arange(N)] = diag
        else:
            els = np.arange(m, Nmax)
            omega_complex = (1. / np.sqrt(2)) * (omega[elm2ind(els, m)] - 1j * omega[elm2ind(els, -m)])
            sgn = 1 if m % 2 == 0 else -1
            diag_m = B[:, :Nmax - m] @ omega_complex
            diag_m *= sgn
            i = np.arange(N - m)
            W_out[i + m, i] = diag_m.conj()
            W_out[i, i + m] = diag_m
    W_out *= 1.0j


def mat2shr_(W, basis, omega_out):
    """quantization.py:286-327."""
    N = W.shape[-1]
    elmax = N - 1
    if omega_out.shape[-1] < N ** 2:
        elmax = int(np.sqrt(omega_out.shape[-1])) - 1
    Nmax = elmax + 1
    sqrt2 = np.sqrt(2.0)
    for m in range(Nmax):
        B = _block(basis, m, N)
        if m == 0:
            tmp = (np.diagonal(W, 0) @ B[:, :Nmax]) / 1.0j
            omega_out[elm2ind(np.arange(0, Nmax), 0)] = tmp.real
        else:
            sgn = 1 if m % 2 == 0 else -1
            els = np.arange(m, Nmax)
            part = np.diagonal(W, -m) @ B[:, :Nmax - m]
            omega_out[elm2ind(els, m)] = sqrt2 * sgn * part.imag
            omega_out[elm2ind(els, -m)] = -sqrt2 * sgn * part.real
    omega_out /= N


def shc2mat_(omega, basis, W_out):
    """quantization.py:331-365."""
    N = W_out.shape[-1]
    for m in range(N):
        B = _block(basis, m, N).astype(W_out.dtype)
        els = np.arange(m, N)
        i = np.arange(N - m)
        W_out[i + m, i] = B @ omega[elm2ind(els, m)]
        if m != 0:
            sgn = 1 if m % 2 == 0 else -1
            W_out[i, i + m] = sgn * B @ omega[elm2ind(els, -m)]
    W_out *= 1.0j


def mat2shc_(W, basis, omega_out):
    """quantization.py:368-396."""
    N = W.shape[0]
    for m in range(N):
        B = _block(basis, m, N).astype(W.dtype)
        els = np.arange(m, N)
        omega_out[elm2ind(els, m)] = np.diagonal(W, -m) @ B
        if m != 0:
            sgn = 1 if m % 2 == 0 else -1
            omega_out[elm2ind(els, -m)] = sgn * np.diagonal(W, m) @ B
    omega_out /= 1.0j * N


def berezin_multipliers(N, el=None):
    """quflow/utils.py:108-135."""
    from scipy.special import gammaln
    if el is None:
        ind = np.arange(N ** 2)
        ells = np.floor(np.sqrt(ind)).astype(np.float64)       # utils.py:73-89
    else:
        ells = np.asarray(el, dtype=np.float64)
    NN = np.float64(N)
    return np.exp(0.5 * (gammaln(NN + 1) + gammaln(NN) - gammaln(NN - ells) - gammaln(NN + ells + 1)))


def shr2mat(omega, N=-1, berezin=False):
    """quantization.py:450-489."""
    if N == -1:
        N = round(np.sqrt(omega.shape[0]))
    W_out = np.zeros((N, N), dtype=complex)
    if berezin:
        bw = berezin_multipliers(N)
        ind = np.nonzero(omega)
        omega = omega.copy()
        omega[ind] /= bw[ind]
    shr2mat_(omega, get_basis(N), W_out)
    return W_out


def mat2shr(W, elmax=-1, berezin=False):
    """quantization.py:492-525."""
    N = W.shape[-1]
    Nmax = N
    if elmax > 0:
        Nmax = (elmax + 1) ** 2
    omega = np.zeros(Nmax ** 2)
    mat2shr_(W, get_basis(N), omega)
    if berezin:
        omega *= berezin_multipliers(N)[:omega.shape[0]]
    return omega


def shc2mat(omega, N=-1, berezin=False):
    """quantization.py:528-566."""
    if N == -1:
        N = round(np.sqrt(omega.shape[0]))
    W_out = np.zeros((N, N), dtype=complex)
    omega = np.asarray(omega, dtype=complex)
    if berezin:
        bw = berezin_multipliers(N)
        ind = np.nonzero(omega)
        omega = omega.copy()
        omega[ind] /= bw[ind]
    shc2mat_(omega, get_basis(N), W_out)
    return W_out


def mat2shc(W, berezin=False):
    """quantization.py:569-592."""
    N = W.shape[0]
    omega = np.zeros(N ** 2, dtype=complex)
    mat2shc_(W, get_basis(N), omega)
    if berezin:
        omega *= berezin_multipliers(N)[:omega.shape[0]]
    return omega
