"""GPU parity tests: the HIP path (through the C ABI, include/quflow_hip.h) against
(1) the CPU oracle on the same seeded inputs and (2) the committed golden vectors that
were produced by running the reference itself (tests/golden/, oracle/gen_golden.py).

Tolerances (fp64):
  * Poisson solve: the reference's own bound 1e-14*N^2 (tests/test_laplacian.py:252) is
    asserted; the observed deviation is also held to 64 eps * max|P| (rounding only).
  * laplace: bit-exact (same operation order, IEEE sqrt).
  * stepper, fixed-iteration mode: max|dW| <= 1e-11 after 100 steps at N=64.
  * stepper, adaptive: 1e-7 is the reference's own acceptance (tests/test_integrators.py:45);
    we assert 1e-11 and identical iteration counts.
"""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

EPS = np.finfo(float).eps


@pytest.fixture(scope="module")
def qfa():
    import quflow_amd
    if quflow_amd.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the MI355X box")
    return quflow_amd


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))))


# ----------------------------------------------------------------------------- Laplacian
@pytest.mark.parametrize("N", [2, 3, 4, 16, 33, 64, 101])
def test_laplacian_table(qfa, N):
    g = load_golden("poisson")
    np.testing.assert_array_equal(qfa.laplacian.laplacian(N, bc=True), g["N%d_lap_bc" % N])


@pytest.mark.parametrize("N", [2, 3, 4, 16, 33, 64, 101])
def test_solve_poisson_golden(qfa, N):
    g = load_golden("poisson")
    for wkey, pkey in (("N%d_W", "N%d_P"), ("N%d_Wtr", "N%d_Ptr")):
        P = qfa.solve_poisson(g[wkey % N])
        ref = g[pkey % N]
        err = maxabs(P, ref)
        assert err <= 1e-14 * N ** 2
        # rounding only: a few ulps of the data scale (the trace removal cancels O(|W|) terms)
        assert err <= 64 * EPS * max(np.abs(ref).max(), np.abs(g[wkey % N]).max()), err
        # skew-Hermitian by construction (cpu.py:334,340)
        np.testing.assert_array_equal(P, -P.conj().T)


@pytest.mark.parametrize("N", [2, 3, 4, 16, 33, 64, 101])
def test_solve_poisson_nonskewh_golden(qfa, N):
    g = load_golden("poisson")
    old = qfa.laplacian.select_skewherm(False)
    try:
        P = qfa.solve_poisson(g["N%d_G" % N]).copy()
    finally:
        assert qfa.laplacian.select_skewherm(old) is False
    ref = g["N%d_PG" % N]
    assert maxabs(P, ref) <= 64 * EPS * np.abs(ref).max()


@pytest.mark.parametrize("N", [2, 3, 4, 16, 33, 64, 101])
def test_laplace_golden(qfa, N):
    g = load_golden("poisson")
    np.testing.assert_array_equal(qfa.laplace(g["N%d_P" % N]), g["N%d_lapP" % N])
    np.testing.assert_array_equal(qfa.laplace(g["N%d_G" % N]), g["N%d_lapG" % N])


def test_solve_poisson_analytic(qfa):
    """The reference's own test (tests/test_laplacian.py:226-252) with its own tolerance."""
    g = load_golden("poisson_analytic")
    for key in g["cases"]:
        key = str(key)
        N = int(key.split("_")[0][1:])
        skewh = key.endswith("sk1")
        old = qfa.laplacian.select_skewherm(skewh)
        try:
            P = qfa.solve_poisson(g[key + "_W"]).copy()
        finally:
            qfa.laplacian.select_skewherm(old)
        np.testing.assert_allclose(P, g[key + "_P"], atol=1e-14 * N ** 2, rtol=0)


def test_solve_poisson_multistate_and_buffer(qfa):
    g = load_golden("poisson")
    P = qfa.solve_poisson(g["multi_W"])          # (2,N,N): state 0 only (cpu.py:696-697)
    assert P.shape == (33, 33)
    assert maxabs(P, g["multi_P"]) <= 64 * EPS * np.abs(g["multi_P"]).max()
    P2 = qfa.solve_poisson(g["N33_W"])           # same persistent buffer (cpu.py:726,734)
    assert P2 is P


@pytest.mark.parametrize("N", [128, 256, 512, 1024, 1000, 2048])
def test_solve_poisson_vs_oracle_large(qfa, oracle, N):
    """Sizes of BASELINE.json's configs, against the oracle on the same seeded input, plus
    the size-independent property laplace(solve(W)) == W."""
    W = oracle.make_W0(N, 11)
    P = qfa.solve_poisson(W).copy()
    Pc = oracle.solve_poisson(W).copy()
    scale = np.abs(Pc).max()
    err = maxabs(P, Pc)
    assert err <= 1e-14 * N ** 2
    assert err <= 256 * EPS * scale, (err, scale)
    back = qfa.laplace(P)
    assert maxabs(back, W) <= 1e-9 * np.abs(W).max()
    old = qfa.laplacian.select_skewherm(False)
    try:
        Pg = qfa.solve_poisson(W).copy()
    finally:
        qfa.laplacian.select_skewherm(old)
    assert maxabs(Pg, Pc) <= 256 * EPS * scale


@pytest.mark.parametrize("N", [256, 333, 500, 1000, 1024, 1025, 1536, 2047, 2048])
def test_solve_poisson_folded_walk_slots(qfa, oracle, monkeypatch, N):
    """k_solve<.., FOLD = 1> (the default above N = 1024, forced here from N = 256 on): walk t and walk N-1-t share a
    slot, the recurrences restart at the junction through the table's zero multiplier.  Odd N (the middle walk has
    no partner), the trace terms of walk 0 (the slot that also carries walk N-1), and against the walk-per-slot
    layout: the same walks cut into different chunks -- rounding-level differences only."""
    W = oracle.make_W0(N, 3)
    Wc = W.copy()
    Wc[np.arange(N), np.arange(N)] += 0.25j           # not trace-free: circulation removed on the way (cpu.py:311-317)
    Pc = oracle.solve_poisson(W).copy()
    scale = np.abs(Pc).max()
    out = {}
    for fold in ("1", "0"):
        monkeypatch.setenv("QUFLOW_HIP_SOLVE_FOLD", fold)
        P = qfa.solve_poisson(W).copy()
        assert maxabs(P, Pc) <= 256 * EPS * scale, (fold, maxabs(P, Pc) / scale)
        np.testing.assert_array_equal(P, -P.conj().T)
        assert abs(np.trace(P)) <= 64 * EPS * N * scale
        out[fold] = P
        # the circulation cancels O(|W|) terms on the main diagonal and the m = 0 system amplifies their rounding
        # (its near-null constant mode, removed again with the mean): same answer to 1e-10 of the scale, bit-equal
        # off the main diagonal
        P2 = qfa.solve_poisson(Wc).copy()
        off = ~np.eye(N, dtype=bool)
        np.testing.assert_array_equal(P2[off], P[off])
        assert maxabs(P2, P) <= 1e-10 * scale
        W32 = W.astype(np.complex64)
        P32 = qfa.solve_poisson(W32).copy()
        assert P32.dtype == np.complex64
        assert maxabs(P32, Pc) <= 8 * N * np.finfo(np.float32).eps * scale
        np.testing.assert_array_equal(P32, -P32.conj().T)
    assert maxabs(out["1"], out["0"]) <= 64 * EPS * scale
    monkeypatch.delenv("QUFLOW_HIP_SOLVE_FOLD")
    back = qfa.laplace(qfa.solve_poisson(W).copy())
    assert maxabs(back, W) <= 1e-9 * np.abs(W).max()


@pytest.mark.parametrize("N", [767, 768, 769, 1150, 1151, 1152, 1153, 2175, 2176, 2177])
def test_solve_poisson_layout_boundaries(qfa, oracle, N):
    """Sizes on either side of every switch in the solve's configuration (walk-per-slot / folded slots at 768, 9- / 17-entry
    chunks at 1152, the end of the folded layout at 2176), both precisions, default settings, against the oracle."""
    W = oracle.make_W0(N, N)
    Pc = oracle.solve_poisson(W).copy()
    scale = np.abs(Pc).max()
    P = qfa.solve_poisson(W).copy()
    assert maxabs(P, Pc) <= 256 * EPS * scale, maxabs(P, Pc) / scale
    np.testing.assert_array_equal(P, -P.conj().T)
    # float32: device (chunked scan order) and oracle (sequential order) are two single-precision evaluations of an
    # ill-conditioned solve -- each against the double-precision solution of the SAME rounded input, as in
    # tests/test_hip_single.py::test_solve_poisson_c64_vs_oracle_large
    W32 = W.astype(np.complex64)
    P32 = qfa.solve_poisson(W32).copy()
    P32_ora = oracle.solve_poisson(W32).copy()
    P32_f64 = oracle.solve_poisson(W32.astype(np.complex128)).copy()
    eps32 = float(np.finfo(np.float32).eps)
    e_dev, e_ora = maxabs(P32, P32_f64) / scale, maxabs(P32_ora, P32_f64) / scale
    assert e_dev <= max(4 * N * eps32, 2 * e_ora), (e_dev / eps32, e_ora / eps32)
    np.testing.assert_array_equal(P32, -P32.conj().T)
    old = qfa.laplacian.select_skewherm(False)
    try:
        Pg = qfa.solve_poisson(W).copy()
    finally:
        qfa.laplacian.select_skewherm(old)
    assert maxabs(Pg, Pc) <= 256 * EPS * scale


@pytest.mark.parametrize("N", [9, 33])
def test_next_solvers(qfa, N):
    """heat / helmholtz / viscdamp share the solve kernel."""
    g = load_golden("next_solvers")
    W = g["N%d_W" % N]
    tol = dict(rtol=0, atol=1e-14)
    np.testing.assert_allclose(qfa.laplacian.solve_helmholtz(W, alpha=0.1), g["N%d_helmholtz_a01" % N], **tol)
    np.testing.assert_allclose(qfa.laplacian.solve_globalqg(W, gamma=2.0), g["N%d_globalqg_g2" % N], **tol)
    np.testing.assert_allclose(qfa.laplacian.solve_heat(1e-3, W), g["N%d_heat_1e3" % N], **tol)
    np.testing.assert_allclose(qfa.laplacian.solve_viscdamp(0.1, W, nu=1e-2, alpha=0.6, theta=0.7),
                               g["N%d_viscdamp" % N], **tol)


def test_f1_reference_test_vectors(qfa):
    """What the reference's own tests hold for heat / helmholtz / viscdamp / laplace
    (tests/test_laplacian.py:255-314, tests/test_geometry.py:81-95), reproduced on the device against
    vectors the reference produced (oracle/gen_golden.py gen_f1_reference_tests)."""
    g = load_golden("f1_reference_tests")
    lap = qfa.laplacian
    # test_solve_viscdamp (:284-314): 100 theta-scheme steps, then mat2shr against the 81-coefficient literal
    Wt = g["smooth_N9_W0"].copy()
    for k in range(100):
        Wt = lap.solve_viscdamp(0.1, Wt, nu=1e-2, alpha=0.6, theta=0.7)
    assert maxabs(Wt, g["viscdamp100_N9_W"]) <= 1e-14
    np.testing.assert_allclose(qfa.mat2shr(Wt), g["viscdamp100_N9_omegatref_literal"], atol=1e-10, rtol=0)
    # test_solve_heat_vs_viscdamp (:270-281)
    for N in (9, 32):
        Wh = g["smooth_N%d_W0" % N].copy()
        Wv = Wh.copy()
        for k in range(100):
            Wh = lap.solve_heat(1e-2 * 0.1, Wh)
            Wv = lap.solve_viscdamp(0.1, Wv, nu=1e-2, alpha=0, theta=1)
        np.testing.assert_allclose(Wh, Wv)
        assert maxabs(Wh, g["heat100_N%d_W" % N]) <= 1e-14
        assert maxabs(Wv, g["viscdamp_alpha0_100_N%d_W" % N]) <= 1e-14
    # test_solve_helmholtz (:255-268): analytic solutions, both branches of select_skewherm
    for N in (33, 65, 128):
        for tag, skewh in (("skewh", True), ("general", False)):
            key = "helm_N%d_%s" % (N, tag)
            old = lap.select_skewherm(skewh)
            try:
                P = lap.solve_helmholtz(g[key + "_W"], alpha=0.1)
            finally:
                lap.select_skewherm(old)
            np.testing.assert_allclose(P, g[key + "_Pexact"])
            assert maxabs(P, g[key + "_P"]) <= 64 * EPS * np.abs(g[key + "_P"]).max()
    # test_hoppe_yau_laplacian (tests/test_geometry.py:81-95)
    for N in (15, 16, 64):
        np.testing.assert_allclose(qfa.laplace(g["hoppe_yau_N%d_P" % N]), g["hoppe_yau_N%d_DeltaP" % N])


@pytest.mark.parametrize("N", [512, 1024])
def test_next_solvers_vs_oracle_large(qfa, oracle, N):
    """heat / helmholtz / viscdamp / globalqg and the resident viscous half step at the bench sizes,
    against the CPU oracle on the same seeded input (rounding only: 64 ulp of the data scale)."""
    lap = qfa.laplacian
    W = oracle.make_W0(N, 5)
    F = oracle.make_W0(N, 6)
    cases = [("helmholtz", lambda m: m.solve_helmholtz(W, alpha=0.1)),
             ("heat", lambda m: m.solve_heat(1e-3, W)),
             ("globalqg", lambda m: m.solve_globalqg(W, gamma=2.0)),
             ("viscdamp theta=1", lambda m: m.solve_viscdamp(0.1, W, nu=1e-3, alpha=0.05)),
             ("viscdamp theta=0.7 + force", lambda m: m.solve_viscdamp(0.1, W, nu=1e-2, alpha=0.3, theta=0.7, force=F))]
    for name, fn in cases:
        ref = np.array(fn(oracle))
        got = np.array(fn(lap))
        assert maxabs(got, ref) <= 64 * EPS * max(np.abs(ref).max(), np.abs(W).max()), name
        np.testing.assert_array_equal(got, -got.conj().T)
    # the half step on the resident state (strang_splitting=ViscDampStep): W <- (1 + h alpha - h nu Delta)^-1 W
    tr = qfa.DeviceTrajectory(W)
    step = qfa.ViscDampStep(nu=1e-3, alpha=0.05)
    step.apply_resident(tr.ctx, 0.1)
    step.apply_resident(tr.ctx, 0.1)
    ref = oracle.solve_viscdamp(0.1, oracle.solve_viscdamp(0.1, W, nu=1e-3, alpha=0.05), nu=1e-3, alpha=0.05)
    assert maxabs(tr.download(), ref) <= 64 * EPS * np.abs(ref).max()
    tr.ctx.close()


def test_factor_cache_is_bounded(qfa, oracle):
    """200 distinct step sizes through solve_viscdamp at N=1024: the device-side factor cache recycles
    its least recently used entries (512 MiB budget = 32 entries of 16 MiB), a recycled key is
    refactored correctly, and a key that returns with another table is not trusted."""
    import ctypes
    from quflow_amd import _lib
    N = 1024
    lap = qfa.laplacian
    W = oracle.make_W0(N, 2)
    ctx = qfa.get_context(N)
    n = ctypes.c_int()
    b = ctypes.c_ulonglong()
    first = None
    for i in range(200):
        h = 0.01 * (1 + i)
        P = lap.solve_viscdamp(h, W, nu=1e-3, alpha=0.05)
        if i == 0:
            first = P.copy()
        _lib.check(ctx._lib.qf_factor_cache_stats(ctx.handle, ctypes.byref(n), ctypes.byref(b)))
        assert b.value <= 512 << 20, (i, n.value, b.value)
    assert n.value <= 32
    assert len(lap._table_cache) <= lap._table_cache.maxlen
    # h = 0.01 was evicted long ago: it comes back bit-identical
    np.testing.assert_array_equal(lap.solve_viscdamp(0.01, W, nu=1e-3, alpha=0.05), first)
    # the same key with a different table: the fingerprint forces a refactorisation
    tabA = np.ascontiguousarray(lap._shifted_table(N, 1.0, 0.3))
    tabB = np.ascontiguousarray(lap._shifted_table(N, 1.0, 0.7))
    PA = lap._solve_with_table(tabA, 12345, W)
    PB = lap._solve_with_table(tabB, 12345, W)
    np.testing.assert_array_equal(PA, lap.solve_helmholtz(W, alpha=0.3))
    np.testing.assert_array_equal(PB, lap.solve_helmholtz(W, alpha=0.7))


def test_factor_cache_budget_switch(qfa, oracle, monkeypatch):
    """QUFLOW_HIP_FACTOR_CACHE_MB (read when a context is created): with a 3 MiB budget a context at N = 256 (1 MiB per
    factor table) never holds more than three, and an evicted step size comes back bit-identical."""
    import ctypes
    from quflow_amd import _lib
    from quflow_amd.context import release_contexts
    N = 256
    lap = qfa.laplacian
    W = oracle.make_W0(N, 2)
    monkeypatch.setenv("QUFLOW_HIP_FACTOR_CACHE_MB", "3")
    release_contexts()
    try:
        ctx = qfa.get_context(N)
        n, b = ctypes.c_int(), ctypes.c_ulonglong()
        first = None
        for i in range(12):
            P = lap.solve_viscdamp(0.01 * (1 + i), W, nu=1e-3, alpha=0.05)
            if i == 0:
                first = P.copy()
            _lib.check(ctx._lib.qf_factor_cache_stats(ctx.handle, ctypes.byref(n), ctypes.byref(b)))
            assert b.value <= 3 << 20 and n.value <= 3, (i, n.value, b.value)
        assert n.value == 3
        np.testing.assert_array_equal(lap.solve_viscdamp(0.01, W, nu=1e-3, alpha=0.05), first)
    finally:
        release_contexts()


# ----------------------------------------------------------------------------- commutator GEMM
@pytest.mark.parametrize("N", [16, 33, 64, 100, 512, 1024])
def test_zgemm_vs_numpy(qfa, N):
    import ctypes
    from quflow_amd import _lib
    from quflow_amd.context import get_context, ptr
    rng = np.random.default_rng(N)
    A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    B = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    # asymmetric operands: catches row/column swaps of the MFMA lane maps
    A[0, 1] += 7.0
    B[2 % N, 0] -= 5.0j
    C = np.zeros_like(A)
    ctx = get_context(N)
    _lib.check(ctx._lib.qf_zgemm(ctx.handle, ptr(A), ptr(B), ptr(C)))
    ref = A @ B
    bound = 8 * EPS * N * (np.abs(A) @ np.abs(B)).max()
    assert maxabs(C, ref) <= bound


def test_zgemm_fresh_context_n1024(qfa):
    """The 64 x 64-tile product kernel on a context of its own (not the shared one) against numpy."""
    from quflow_amd import _lib
    from quflow_amd.context import Context, ptr
    N = 1024
    rng = np.random.default_rng(7)
    A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    B = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    ctx = Context(N)
    try:
        C = np.zeros_like(A)
        _lib.check(ctx._lib.qf_zgemm(ctx.handle, ptr(A), ptr(B), ptr(C)))
    finally:
        ctx.close()
    ref = A @ B
    assert maxabs(C, ref) <= 16 * EPS * N * (np.abs(A) @ np.abs(B)).max()


@pytest.mark.parametrize("digits", [5, 6])
@pytest.mark.parametrize("N", [64, 128, 512, 1024])
def test_zgemm_i8_vs_numpy(qfa, N, digits, monkeypatch):
    """The digit-split product on the INT8 matrix cores (ozaki.hip): general A (graded rows, to
    exercise the per-row scales) times skew-Hermitian B, against numpy.  The only error is the
    truncation of the digit series: 128^-digits of (row scale) x (column scale) per term
    (5 digits: QUFLOW_HIP_GEMM=i8; 6 digits: i8x6)."""
    from quflow_amd import _lib
    from quflow_amd.context import Context, ptr
    monkeypatch.setenv("QUFLOW_HIP_GEMM", "i8" if digits == 5 else "i8x6")
    rng = np.random.default_rng(N)
    A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    A *= (10.0 ** rng.uniform(-6, 2, size=(N, 1)))            # rows spanning 8 orders of magnitude
    A[0, 1] += 7.0
    B = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    B = B - B.conj().T
    B *= np.exp(-0.05 * np.abs(np.subtract.outer(np.arange(N), np.arange(N))))   # decaying off the diagonal
    B = np.ascontiguousarray(B)
    C = np.zeros_like(A)
    ctx = Context(N)          # the digit count is read when the context is created
    try:
        _lib.check(ctx._lib.qf_zgemm_i8(ctx.handle, ptr(A), ptr(B), ptr(C)))
    finally:
        ctx.close()
    ref = A @ B
    rowscale = np.abs(A).max(axis=1, keepdims=True)
    colscale = np.abs(B).max(axis=0, keepdims=True)
    ulp = 2.0 ** (-7 * digits)
    bound = 16 * N * ulp * rowscale * colscale                 # generous: 2x2 scale slack (round 5: 4x4 until then), 3M combination
    assert np.all(np.abs(C - ref) <= bound)
    # and it is far better than that in practice (truncation errors average out over k)
    assert np.all(np.abs(C - ref) <= 4 * np.sqrt(N) * ulp * 4 * rowscale * colscale)


def test_zgemm_identity_asymmetric(qfa):
    from quflow_amd import _lib
    from quflow_amd.context import get_context, ptr
    N = 64
    A = np.eye(N, dtype=complex)
    B = (np.arange(N * N).reshape(N, N) + 1j * np.arange(N * N)[::-1].reshape(N, N)).astype(complex)
    C = np.zeros_like(B)
    ctx = get_context(N)
    _lib.check(ctx._lib.qf_zgemm(ctx.handle, ptr(A), ptr(B), ptr(C)))
    np.testing.assert_array_equal(C, B)
    _lib.check(ctx._lib.qf_zgemm(ctx.handle, ptr(B), ptr(A), ptr(C)))
    np.testing.assert_array_equal(C, B)




def _iteration_operands(N, seed):
    """Skew-Hermitian P (already scaled by eps), W, dW_old of the magnitudes an isomp iteration sees."""
    rng = np.random.default_rng(seed)

    def skew(scale):
        A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
        A = A - A.conj().T
        return A * (scale / np.abs(A).max())
    P = skew(0.05)
    W = skew(1.0)
    dW_old = skew(0.01)
    return P, W, dW_old


@pytest.mark.parametrize("N,min_units,epi_units", [(64, 1, 0), (64, 1, 3), (64, 8, 8), (128, 1, 5), (192, 2, 14),
                                                   (256, 8, 8), (512, 8, 40), (1024, 8, 0), (1024, 8, 14),
                                                   (1088, 8, 30), (2048, 8, 14)])
def test_fixedpoint_products_full_vs_triangle(qfa, N, min_units, epi_units, monkeypatch):
    """One fixed-point iteration's products + fused epilogue (isospectral.py:496-509,481-482,
    526-534): the full second product and the upper-triangle stream-K form, both against numpy.
    min_units=1 forces one K-tile per workgroup, i.e. the most partial-tile exchanges."""
    from quflow_amd import _lib
    from quflow_amd.context import Context, ptr
    P, W, dW_old = _iteration_operands(N, N + min_units)
    Whalf = W + dW_old
    PW = P @ Whalf
    T = PW @ P
    dW_ref = T + (PW - PW.conj().T)
    Whalf_ref = W + dW_ref
    rows_ref = np.abs(dW_old - dW_ref).sum(axis=1)
    bound = 16 * EPS * N * (np.abs(PW) @ np.abs(P)).max() + 4 * EPS * np.abs(PW).max()
    # the partition of the stream-K form goes in with the variant (the stepper's follows rules): least K-tiles per workgroup,
    # weight of a finisher's extra work (0 = plain K-tile split)
    tri = 1 | (min_units << 8) | ((epi_units + 64) << 16)
    ctx = Context(N)
    out = {}
    try:
        for variant in (0, tri):
            dW = np.zeros_like(W)
            Wh = np.zeros_like(W)
            rows = np.zeros(N)
            _lib.check(ctx._lib.qf_fixedpoint_products(ctx.handle, ptr(P), ptr(Whalf), ptr(W), ptr(dW_old), variant,
                                                       ptr(dW), ptr(Wh), ptr(rows)))
            out[variant & 15] = (dW, Wh, rows)
            assert maxabs(dW, dW_ref) <= bound, variant
            assert maxabs(Wh, W + dW) <= 2 * EPS * np.abs(W).max(), variant      # Whalf = W + dW, elementwise
            assert maxabs(rows, rows_ref) <= N * (bound + 4 * EPS * np.abs(dW_old).max()), variant
    finally:
        ctx.close()
    # the triangle form writes an exactly skew-Hermitian dW outside the diagonal tiles
    dW1 = out[1][0]
    blk = np.arange(N) // 64
    off = blk[:, None] != blk[None, :]
    assert np.array_equal(dW1[off], (-dW1.conj().T)[off])
    assert maxabs(out[0][0], out[1][0]) <= bound


@pytest.mark.parametrize("N,split", [(64, "2,1"), (64, "2,2"), (96, "2,1"), (128, "1,1"), (256, "2,2"), (512, "2,1"),
                                     (512, "1,2"), (736, "2,1"), (1024, "2,1"), (512, "4,4"), (512, "4,2"), (256, "4,4"),
                                     (64, "4,4"), (704, "4,1"), (100, "2,1"), (333, "2,2"), (1000, "1,1"), (1001, "4,4"), (72, "1,2")])
def test_fixedpoint_products_tri32(qfa, N, split, monkeypatch):
    """The second product on the upper triangle of 32x32 tiles with the K range of a tile split over two
    workgroups (k_zgemm_tri32, the default below N = 768): against numpy and against the full product, for
    every split of the off-diagonal / diagonal tiles; the result is exactly skew-Hermitian outside the
    diagonal tiles and the same bits on every run (the two-term combine does not depend on which half arrives last)."""
    from quflow_amd import _lib
    from quflow_amd.context import Context, ptr
    P, W, dW_old = _iteration_operands(N, N + 7)
    Whalf = W + dW_old
    PW = P @ Whalf
    dW_ref = PW @ P + (PW - PW.conj().T)
    rows_ref = np.abs(dW_old - dW_ref).sum(axis=1)
    bound = 16 * EPS * N * (np.abs(PW) @ np.abs(P)).max() + 4 * EPS * np.abs(PW).max()
    so, sd = (int(x) for x in split.split(","))
    tri32 = 2 | (so << 8) | (sd << 12)          # (the split goes in with the variant: the stepper's follows a rule)
    ctx = Context(N)
    out = {}
    try:
        for variant in (0, tri32, tri32, tri32):
            dW = np.zeros_like(W)
            Wh = np.zeros_like(W)
            rows = np.zeros(N)
            _lib.check(ctx._lib.qf_fixedpoint_products(ctx.handle, ptr(P), ptr(Whalf), ptr(W), ptr(dW_old), variant,
                                                       ptr(dW), ptr(Wh), ptr(rows)))
            assert maxabs(dW, dW_ref) <= bound, variant
            assert maxabs(Wh, W + dW) <= 2 * EPS * np.abs(W).max(), variant
            assert maxabs(rows, rows_ref) <= N * (bound + 4 * EPS * np.abs(dW_old).max()), variant
            if (variant & 15) in out and (variant & 15) == 2:
                for a, b in zip(out[2], (dW, Wh, rows)):
                    np.testing.assert_array_equal(a, b)           # bit-reproducible
            out[variant & 15] = (dW, Wh, rows)
    finally:
        ctx.close()
    dW2 = out[2][0]
    blk = np.arange(N) // 32
    off = blk[:, None] != blk[None, :]
    assert np.array_equal(dW2[off], (-dW2.conj().T)[off])
    Wh2 = out[2][1]
    assert np.array_equal(Wh2[off], (-Wh2.conj().T)[off])
    assert maxabs(out[0][0], dW2) <= bound


def test_isomp_second_product_variants_agree(qfa, monkeypatch):
    """isomp at N=256 (20 steps) with the full second product and with the upper-triangle form:
    same iteration counts, results equal to rounding; a W that is not skew-Hermitian must take
    the full product (and match numpy's general iteration)."""
    import quflow_amd
    from quflow_amd.integrators import isomp
    from quflow_amd.context import release_contexts
    N = 256
    rng = np.random.default_rng(5)
    A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    W0 = A - A.conj().T
    W0 -= np.eye(N) * np.trace(W0) / N
    W0 /= np.linalg.norm(W0) / np.sqrt(N)
    dt = 0.25 * quflow_amd.hbar(N)
    res = {}
    for mode in ("full", "tri", "tri32"):
        # "tri": the 64x64 stream-K form forced on (default: N >= 960 only); "tri32": what N = 256 takes by default
        monkeypatch.setenv("QUFLOW_HIP_TRI_MIN_N", "960" if mode == "tri32" else "64")
        monkeypatch.setenv("QUFLOW_HIP_GEMM2", "full" if mode == "full" else "tri")
        release_contexts()
        st = {"iterations": 0.0}
        res[mode] = (isomp(W0.copy(), dt, steps=20, stats=st), dict(st))
    release_contexts()
    for mode in ("tri", "tri32"):
        assert res["full"][1]["iterations"] == res[mode][1]["iterations"]
        assert maxabs(res["full"][0], res[mode][0]) <= 1e-13
        Wt = res[mode][0]
        assert np.array_equal(Wt, -Wt.conj().T)


# ----------------------------------------------------------------------------- stepper
STEP_TOL = 1e-11
# the 5-digit int8 products (QUFLOW_HIP_GEMM=i8, a speed demonstration -- config 3 is the 6-digit mode,
# tested without these floors): the series is cut at 2^-35 ~ 3e-11 relative per product, ~1e-10 per
# step in the state (measured: 4e-11 after 100 small steps, 4e-9 .. 6e-9 after 10-40 large ones); the
# spectrum / Casimirs drift at the 1e-11 level instead of fp64's 1e-13
I8_TOL = 2e-8
I8_DRIFT = 2e-11


@pytest.mark.parametrize("tag", ["s010", "s025"])
def test_isomp_n64_golden(qfa, tag):
    g = load_golden("isomp_n64")
    N = 64
    W0 = qfa.ensemble.make_W0(N, 0)
    stats = {"iterations": 0.0}
    dt = float(g[tag + "_stepsize"]) * qfa.hbar(N)
    W = qfa.isomp(W0.copy(), dt, steps=100, stats=stats)
    assert maxabs(W, g[tag + "_W"]) <= STEP_TOL
    assert stats["iterations"] == float(g[tag + "_iterations"])
    assert stats["number_of_maxit"] == float(g[tag + "_number_of_maxit"])
    np.testing.assert_allclose(stats["tol_auto"], float(g[tag + "_tol_auto"]), rtol=1e-13)
    # invariants: drift no worse than the reference's on the same input
    spec0 = g[tag + "_spec0"]
    drift_ref = np.abs(g[tag + "_spec"] - spec0).max()
    drift = np.abs(np.linalg.eigvalsh(1j * W) - spec0).max()
    assert drift <= max(1.05 * drift_ref, 10 * EPS * 100)
    # chunked run + diagnostics from the device
    W = W0.copy()
    E, S = [qfa.energy_euler(W)], [qfa.enstrophy(W)]
    for _ in range(10):
        W = qfa.isomp(W, dt, steps=10)
        E.append(qfa.energy_euler(W))
        S.append(qfa.enstrophy(W))
    assert maxabs(W, g[tag + "_Wchunk"]) <= STEP_TOL
    np.testing.assert_allclose(E, g[tag + "_energy"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(S, g[tag + "_enstrophy"], rtol=0, atol=1e-13)


@pytest.mark.parametrize("mode", ["tri_fused", "tri_unfused", "tri32_fused", "tri32_unfused", "tri32nd_fused", "full_fused",
                                  "full_unfused", "i8_fused", "i8x6_fused"])
def test_isomp_n64_golden_triangle_protocols(qfa, mode, monkeypatch):
    """The N=64 reference fixtures under every combination of second-product kernel (the
    upper-triangle stream-K form forced on -- default: N >= 768 only -- or the full product) and
    step-end protocol (fused: decision + W update inside the product's epilogue / last tile; or the
    separate decide/update kernels): same results, same iteration counts, chunking,
    fixed-iteration and maxit-exhaustion cases."""
    from quflow_amd.context import release_contexts
    if mode.startswith("tri32"):
        # the default below N = 768: upper triangle of 32x32 tiles, K split over two workgroups (k_zgemm_tri32; the other
        # splits: test_fixedpoint_products_tri32)
        if "nd" in mode:
            monkeypatch.setenv("QUFLOW_HIP_DEFER", "0")     # the exit decision back in the product's last finisher
    elif mode.startswith("tri"):
        monkeypatch.setenv("QUFLOW_HIP_TRI_MIN_N", "64")
    elif mode.startswith("i8"):
        # both products on the int8 matrix cores by digit splitting (ozaki.hip).  The 5-digit series
        # is cut at 2^-35 relative to the row scales, i.e. each product carries a ~3e-11 relative
        # error where fp64 carries ~1e-16: after 100 steps the state agrees with the fp64 fixtures
        # to I8_TOL, not STEP_TOL, and a step's iteration count may differ by one near the exit test
        # With a sixth digit (i8x6: 2^-42) the fp64 fixtures are met at the fp64 tolerance STEP_TOL,
        # iteration counts identical.
        monkeypatch.setenv("QUFLOW_HIP_GEMM", "i8x6" if mode.startswith("i8x6") else "i8")
        monkeypatch.setenv("QUFLOW_HIP_I8_MIN_N", "64")
    else:
        monkeypatch.setenv("QUFLOW_HIP_GEMM2", "full")
    monkeypatch.setenv("QUFLOW_HIP_FUSED", "1" if mode.endswith("_fused") else "0")
    release_contexts()
    five = mode.startswith("i8") and not mode.startswith("i8x6")
    tol = I8_TOL if five else STEP_TOL

    def same_count(got, want):
        if five:
            return abs(got - want) <= 0.05 * want
        return got == want
    try:
        g = load_golden("isomp_n64")
        N = 64
        W0 = qfa.ensemble.make_W0(N, 0)
        for tag in ("s010", "s025"):
            stats = {"iterations": 0.0}
            dt = float(g[tag + "_stepsize"]) * qfa.hbar(N)
            W = qfa.isomp(W0.copy(), dt, steps=100, stats=stats)
            assert maxabs(W, g[tag + "_W"]) <= tol
            assert same_count(stats["iterations"], float(g[tag + "_iterations"]))
            assert same_count(stats["number_of_maxit"], float(g[tag + "_number_of_maxit"])) or five
            assert np.array_equal(W, -W.conj().T)
            W = W0.copy()
            for _ in range(10):
                W = qfa.isomp(W, dt, steps=10)
            assert maxabs(W, g[tag + "_Wchunk"]) <= tol
        # fixed iteration counts (minit = maxit) and maxit exhaustion
        stats = {"iterations": 0.0}
        # (the two smoothed-IC cases below amplify a product's truncation ~100x more than the white-noise
        # ones: 5 digits 4e-9 .. 6e-9, 6 digits 3e-11 .. 5e-11)
        tol_b = 1e-10 if mode.startswith("i8x6") else tol
        W = qfa.isomp(g["icb_W0"].copy(), 0.5 * qfa.hbar(N), steps=10, maxit=3, stats=stats)
        assert maxabs(W, g["maxit3_W"]) <= tol_b
        assert same_count(stats["iterations"], float(g["maxit3_iterations"]))
        stats = {"iterations": 0.0}
        W = qfa.isomp(g["icb_W0"].copy(), 0.25 * qfa.hbar(N), steps=40, stats=stats)
        assert maxabs(W, g["icb_W"]) <= tol_b
        assert same_count(stats["iterations"], float(g["icb_iterations"]))
    finally:
        release_contexts()


def test_isomp_fixed_iterations_golden(qfa):
    g = load_golden("isomp_n64")
    N = 64
    W0 = qfa.ensemble.make_W0(N, 0)
    stats = {"iterations": 0.0}
    W = qfa.isomp(W0.copy(), 0.01 * qfa.hbar(N), steps=32, minit=10, maxit=10, stats=stats)
    assert maxabs(W, g["fixed10_W"]) <= STEP_TOL
    assert stats["iterations"] == 10.0
    assert stats["number_of_maxit"] == float(g["fixed10_number_of_maxit"])
    stats = {"iterations": 0.0}
    W = qfa.isomp(W0.copy(), 0.25 * qfa.hbar(N), steps=32, minit=4, maxit=4, stats=stats)
    assert maxabs(W, g["fixed4_W"]) <= STEP_TOL
    assert stats["number_of_maxit"] == float(g["fixed4_number_of_maxit"])


def test_isomp_compsum_golden(qfa):
    g = load_golden("isomp_n64")
    N = 64
    W0 = qfa.ensemble.make_W0(N, 0)
    stats = {"iterations": 0.0}
    W = qfa.isomp(W0.copy(), 0.10 * qfa.hbar(N), steps=100, compsum=True, stats=stats)
    assert maxabs(W, g["compsum_W"]) <= STEP_TOL
    # eps-level tolerance: the exit is decided by rounding noise, allow a small spread
    assert abs(stats["iterations"] - float(g["compsum_iterations"])) <= 0.5
    np.testing.assert_allclose(stats["tol_auto"], float(g["compsum_tol_auto"]), rtol=1e-13)
    drift = np.abs(np.linalg.eigvalsh(1j * W) - g["s010_spec0"]).max()
    drift_ref = np.abs(g["compsum_spec"] - g["s010_spec0"]).max()
    assert drift <= max(2 * drift_ref, 2e-14)


def test_isomp_options_golden(qfa):
    g = load_golden("isomp_n64")
    N = 64
    W0 = qfa.ensemble.make_W0(N, 0)
    stats = {"iterations": 0.0}
    W = qfa.isomp(W0.copy(), 0.25 * qfa.hbar(N), steps=20, tol=1e-10, reinitialize=True, stats=stats)
    assert maxabs(W, g["tol1e10_reinit_W"]) <= STEP_TOL
    assert stats["iterations"] == float(g["tol1e10_reinit_iterations"])
    stats = {"iterations": 0.0}
    W = qfa.isomp(g["icb_W0"].copy(), 0.25 * qfa.hbar(N), steps=40, stats=stats)
    assert maxabs(W, g["icb_W"]) <= STEP_TOL
    assert stats["iterations"] == float(g["icb_iterations"])
    np.testing.assert_allclose(qfa.energy_euler(W), float(g["icb_energy"]), rtol=1e-12)
    stats = {"iterations": 0.0}
    W = qfa.isomp(g["icb_W0"].copy(), 0.5 * qfa.hbar(N), steps=10, maxit=3, stats=stats)
    assert maxabs(W, g["maxit3_W"]) <= STEP_TOL
    assert stats["iterations"] == float(g["maxit3_iterations"])
    assert stats["number_of_maxit"] == float(g["maxit3_number_of_maxit"])


def test_isomp_chunking_semantics(qfa):
    g = load_golden("isomp_chunking")
    N = 32
    W0 = qfa.ensemble.make_W0(N, 3)
    dt = 0.25 * qfa.hbar(N)
    one = qfa.isomp(W0.copy(), dt, steps=40)
    W = W0.copy()
    for _ in range(4):
        W = qfa.isomp(W, dt, steps=10)
    assert maxabs(one, g["one_call"]) <= STEP_TOL
    assert maxabs(W, g["four_calls"]) <= STEP_TOL
    assert maxabs(one, W) > 1e-12
    one_r = qfa.isomp(W0.copy(), dt, steps=40, reinitialize=True)
    W = W0.copy()
    for _ in range(4):
        W = qfa.isomp(W, dt, steps=10, reinitialize=True)
    np.testing.assert_array_equal(one_r, W)      # bit-identical, like the reference
    assert maxabs(one_r, g["one_call_reinit"]) <= STEP_TOL
    # two identically chunked runs are bit-identical (deterministic reductions)
    W2 = W0.copy()
    for _ in range(4):
        W2 = qfa.isomp(W2, dt, steps=10, reinitialize=True)
    np.testing.assert_array_equal(W, W2)


def test_isomp_literal16(qfa):
    g = load_golden("isomp_literal16")
    W0 = g["W0"]
    dt = qfa.hbar(16) * float(g["stepsize"])
    steps = int(g["steps"])
    for tag, kw in (("auto", {}), ("tol1e10", {"tol": 1e-10}),
                    ("auto_compsum", {"compsum": True}), ("tol1e10_compsum", {"compsum": True, "tol": 1e-10})):
        W = qfa.isomp(W0.copy(), dt, steps, **kw)
        np.testing.assert_allclose(W, g["W_" + tag], rtol=0, atol=1e-10)
    W = qfa.isomp(W0.copy(), dt, steps)
    # isospectrality known-answer: spectrum(W_final) == spectrum(W0) == spectrum(stale literal)
    np.testing.assert_allclose(np.linalg.eigvalsh(1j * W), np.linalg.eigvalsh(1j * W0), atol=1e-9)
    np.testing.assert_allclose(np.linalg.eigvalsh(1j * g["Wfinal_stale"]), np.linalg.eigvalsh(1j * W), atol=2e-8)


@pytest.mark.parametrize("N", [5, 16, 61])
def test_isomp_vs_rk4(qfa, N):
    """tests/test_integrators.py:21-34."""
    g = load_golden("isomp_vs_rk4")
    W = qfa.isomp(g["N%d_W0" % N].copy(), 0.02 * qfa.hbar(N), 500)
    assert maxabs(W, g["N%d_isomp" % N]) <= 1e-10
    np.testing.assert_allclose(W, g["N%d_rk4" % N], atol=1e-2, rtol=0)


@pytest.mark.parametrize("method", ["euler", "heun", "rk4"])
@pytest.mark.parametrize("N", [16, 33, 64])
def test_erk_golden(qfa, method, N):
    """euler / heun / rk4 on the device (qf_erk) against the reference's own output
    (quflow/integrators/erk.py:19-160).  Skew-Hermitian data: one product per stage."""
    g = load_golden("erk")
    pre = "N%d_" % N
    W0 = g[pre + "W0"].copy()
    W = getattr(qfa, method)(W0, float(g[pre + "dt"]), int(g[pre + "steps"]))
    assert W is W0                                   # in-place contract (erk.py:56,110,156)
    assert maxabs(W, g[pre + method]) <= 1e-13


def test_erk_general_matrix_golden(qfa):
    """select_skewherm(False): generic Poisson solve, both products of the bracket."""
    g = load_golden("erk")
    old = qfa.laplacian.select_skewherm(False)
    try:
        for method in ("rk4", "heun"):
            W = getattr(qfa, method)(g["G_W0"].copy(), float(g["G_dt"]), int(g["G_steps"]))
            assert maxabs(W, g["G_" + method]) <= 1e-13
    finally:
        qfa.laplacian.select_skewherm(old)


@pytest.mark.parametrize("N", [5, 16, 61])
def test_rk4_vs_reference_and_isomp(qfa, N):
    """tests/test_integrators.py:21-34 with BOTH steppers on the device."""
    g = load_golden("isomp_vs_rk4")
    dt = 0.02 * qfa.hbar(N)
    Wrk4 = qfa.rk4(g["N%d_W0" % N].copy(), dt, 500)
    assert maxabs(Wrk4, g["N%d_rk4" % N]) <= 1e-11
    Wisomp = qfa.isomp(g["N%d_W0" % N].copy(), dt, 500)
    np.testing.assert_allclose(Wrk4, Wisomp, atol=1e-2, rtol=0)


@pytest.mark.parametrize("method,steps", [("euler", 4), ("heun", 3), ("rk4", 2)])
def test_erk_vs_oracle_large(qfa, oracle, method, steps):
    """N=1024 (64x64-tile products, one product per stage) against the CPU oracle; euler also
    fills stats like the reference (erk.py:58-59)."""
    N = 1024
    W0 = oracle.make_W0(N, 2)
    dt = 0.05 * qfa.hbar(N)
    Wc = getattr(oracle, method)(W0.copy(), dt, steps)
    kw = {}
    stats = {"steps": 5}
    if method == "euler":
        kw["stats"] = stats
    Wg = getattr(qfa, method)(W0.copy(), dt, steps, **kw)
    assert maxabs(Wg, Wc) <= 1e-12
    assert np.array_equal(Wg, -Wg.conj().T)
    if method == "euler":
        assert stats["steps"] == 5 + steps


def test_erk_rejects_unsupported(qfa):
    """Hooks on stacks run (test_erk_hooks_on_stacks_golden); a Hamiltonian whose result is neither one stream matrix nor
    one per state is an error, hooks on a complex64 stack are refused."""
    W = np.stack([qfa.ensemble.make_W0(8, 0)] * 2)
    with pytest.raises(ValueError):
        qfa.heun(W.copy(), 0.1, 1, hamiltonian=lambda W: W[:, :4, :4])       # neither (N,N) nor (k,N,N)
    with pytest.raises(NotImplementedError):
        qfa.rk4(W.astype(np.complex64), 0.1, 1, forcing=lambda P, W: W)


@pytest.mark.parametrize("n", [16, 24])
def test_isomp_hamiltonian_per_state_golden(qfa, n):
    """isomp on a (k,N,N) stack with a foreign Hamiltonian that returns one stream matrix PER STATE (np.matmul batches
    the products, isospectral.py:496-499), alone and with a forcing that sees that (k,N,N) P: qf_isomp_hooked with
    qf_isomp_hooks::states_p.  Against the reference's own runs, iteration counts included."""
    g = load_golden("interfaces")
    pre = "isomp_ps_N%d_" % n
    S0, dt, steps = g[pre + "S0"], float(g[pre + "dt"]), int(g[pre + "steps"])

    def perstate(st):
        assert st.shape == S0.shape
        return np.stack([(0.5 + 0.25 * j) * qfa.solve_poisson(st[j]).copy() for j in range(st.shape[0])])

    def force(P, W):
        assert P.shape == S0.shape and W.shape == S0.shape
        return -0.05 * W + 0.01 * P
    st = {"iterations": 0.0}
    S = S0.copy()
    out = qfa.isomp(S, dt, steps=steps, hamiltonian=perstate, stats=st)
    assert out is S
    assert maxabs(S, g[pre + "W"]) <= 1e-12
    assert st["iterations"] == float(g[pre + "iterations"])
    st = {"iterations": 0.0}
    S = qfa.isomp(S0.copy(), dt, steps=steps, hamiltonian=perstate, forcing=force, stats=st)
    assert maxabs(S, g[pre + "W_forcing"]) <= 1e-12
    assert st["iterations"] == float(g[pre + "iterations_forcing"])


@pytest.mark.parametrize("time,strang", [(None, False), (0.5, False), (None, True), (0.5, True)])
def test_foreign_hamiltonian_is_called_as_often_as_the_reference_calls_it(qfa, oracle, time, strang):
    """Round-3 / round-4 advisor: the one-matrix-or-one-per-state question must not cost the user's Hamiltonian an
    extra evaluation.  The reference evaluates it once per fixed-point iteration, plus the autonomy probe when `time`
    is given (isospectral.py:416-423, 488-491); the oracle restates exactly that, so the call counts must agree --
    for isomp on a stack, with a Strang half step in front of the first evaluation (isospectral.py:466-468: the first
    evaluation is then NOT on the input, which an entry probe's cache would miss), for rk4 on a stack, and a (1,N,N)
    stream matrix for a (1,N,N) stack is accepted.  The question is settled by the stepper's first evaluation
    (qf_isomp_hooks::states_p = -1)."""
    n = 24
    S0 = np.stack([oracle.make_W0(n, 3), oracle.make_W0(n, 4)])
    dt = 0.2 * qfa.hbar(n)
    calls = {"dev": 0, "cpu": 0}

    def ham_dev(st, **kw):
        calls["dev"] += 1
        return qfa.solve_poisson(st[0]).copy()

    def ham_cpu(st, **kw):
        calls["cpu"] += 1
        return oracle.solve_poisson(st[0]).copy()
    sd, sc = {"iterations": 0.0}, {"iterations": 0.0}
    kw = {} if time is None else {"time": time}
    if strang:
        kw["strang_splitting"] = lambda h, St: St * (1.0 - 0.01 * h)
    Wd = qfa.isomp(S0.copy(), dt, steps=4, hamiltonian=(lambda st: ham_dev(st)) if time is None else ham_dev, stats=sd, **kw)
    Wc = oracle.isomp(S0.copy(), dt, steps=4, hamiltonian=(lambda st: ham_cpu(st)) if time is None else ham_cpu, stats=sc, **kw)
    assert maxabs(Wd, Wc) <= 1e-12 and sd["iterations"] == sc["iterations"]
    assert calls["dev"] == calls["cpu"], calls
    # ... and one stream matrix per state, settled by the same first evaluation
    calls["dev"] = calls["cpu"] = 0

    def per_dev(st, **kw):
        calls["dev"] += 1
        return np.stack([qfa.solve_poisson(x).copy() for x in st])

    def per_cpu(st, **kw):
        calls["cpu"] += 1
        return np.stack([oracle.solve_poisson(x).copy() for x in st])
    Wd = qfa.isomp(S0.copy(), dt, steps=3, hamiltonian=(lambda st: per_dev(st)) if time is None else per_dev, stats=sd, **kw)
    Wc = oracle.isomp(S0.copy(), dt, steps=3, hamiltonian=(lambda st: per_cpu(st)) if time is None else per_cpu, stats=sc, **kw)
    assert maxabs(Wd, Wc) <= 1e-12 and sd["iterations"] == sc["iterations"]
    assert calls["dev"] == calls["cpu"], calls
    if time is None and not strang:
        calls["dev"] = 0
        qfa.rk4(S0.copy(), dt, steps=3, hamiltonian=lambda st: ham_dev(st))
        assert calls["dev"] == 4 * 3, calls          # four right-hand sides per step (erk.py:115-160), no probe
        # a one-state stack whose Hamiltonian answers with a (1,N,N) array: numpy broadcasts it, so must the hooks
        one = S0[:1].copy()
        Wd1 = qfa.isomp(one.copy(), dt, steps=2, hamiltonian=lambda st: qfa.solve_poisson(st[0]).copy()[None])
        Wc1 = oracle.isomp(one.copy(), dt, steps=2, hamiltonian=lambda st: oracle.solve_poisson(st[0]).copy()[None])
        assert maxabs(Wd1, Wc1) <= 1e-12


def test_compsum_with_tri32_at_n1024_sums_every_row_partial(qfa, oracle, monkeypatch):
    """Round-3 advisor (medium): with QUFLOW_HIP_TRI_MIN_N above N the second product at N = 1024 is k_zgemm_tri32,
    which leaves (N+31)/32 = 32 partial row sums per row; the two-kernel step end (compsum) must add all of them, not
    the N/64 = 16 of the 64-wide kernels -- half the columns would stop the iteration early.  Against the oracle."""
    from quflow_amd.context import release_contexts
    N, steps = 1024, 2
    monkeypatch.setenv("QUFLOW_HIP_TRI_MIN_N", "4096")
    release_contexts()
    try:
        W0 = oracle.make_W0(N, 0)
        dt = 0.25 * qfa.hbar(N)
        sg, sc = {"iterations": 0.0}, {"iterations": 0.0}
        Wg = qfa.isomp(W0.copy(), dt, steps=steps, compsum=True, stats=sg)
        Wc = oracle.isomp(W0.copy(), dt, steps=steps, compsum=True, stats=sc)
        assert sg["iterations"] == sc["iterations"], (sg, sc)
        np.testing.assert_allclose(sg["tol_auto"], sc["tol_auto"], rtol=1e-12)
        assert maxabs(Wg, Wc) <= STEP_TOL
        # and reinitialize (the other user of the two-kernel protocol)
        sg, sc = {"iterations": 0.0}, {"iterations": 0.0}
        Wg = qfa.isomp(W0.copy(), dt, steps=steps, reinitialize=True, stats=sg)
        Wc = oracle.isomp(W0.copy(), dt, steps=steps, reinitialize=True, stats=sc)
        assert sg["iterations"] == sc["iterations"], (sg, sc)
        assert maxabs(Wg, Wc) <= STEP_TOL
    finally:
        release_contexts()


@pytest.mark.parametrize("n", [16, 33])
def test_erk_hooks_on_stacks_golden(qfa, n):
    """euler / heun / rk4 on a (k,N,N) stack WITH hooks (qf_erk_states_hooked): forcing(P, stack) returns a stack, a
    foreign Hamiltonian sees the stack and returns one (N,N) stream matrix; states, stage arguments and accumulators
    stay on the device.  Against the reference's own runs."""
    g = load_golden("interfaces")
    pre = "erk_N%d_" % n
    S0, dt, steps = g[pre + "S0"], float(g[pre + "dt"]), int(g[pre + "steps"])

    def force(P, st):
        assert P.shape == (n, n) and st.shape == S0.shape
        return -0.05 * st

    def foreign(st):
        assert st.shape == S0.shape
        return 0.5 * qfa.solve_poisson(st[0]) + 0.1j * np.eye(n)
    S = S0.copy()
    out = qfa.rk4(S, dt, steps, forcing=force)
    assert out is S
    assert maxabs(S, g[pre + "rk4_forcing"]) <= 1e-12
    assert maxabs(qfa.heun(S0.copy(), dt, steps, hamiltonian=foreign), g[pre + "heun_foreign"]) <= 1e-12
    assert maxabs(qfa.euler(S0.copy(), dt, steps, hamiltonian=foreign, forcing=force), g[pre + "euler_both"]) <= 1e-12
    # a Hamiltonian that returns one stream matrix PER STATE: the bracket becomes a batched product
    def perstate(st):
        return np.stack([(0.5 + 0.25 * j) * qfa.solve_poisson(st[j]).copy() for j in range(st.shape[0])])
    assert maxabs(qfa.rk4(S0.copy(), dt, steps, hamiltonian=perstate), g[pre + "rk4_perstate"]) <= 1e-12
    # an exception raised by a hook comes back as itself
    def bad(P, st):
        raise RuntimeError("forcing failed")
    with pytest.raises(RuntimeError, match="forcing failed"):
        qfa.rk4(S0.copy(), dt, 1, forcing=bad)


@pytest.mark.parametrize("tag", ["s010", "s050"])
@pytest.mark.parametrize("N", [16, 33, 64])
def test_lu_steppers_golden(qfa, N, tag):
    """isomp_simple / isomp_quasinewton on the device (Newton-Schulz inverse on the matrix cores
    in place of the reference's LAPACK LU, isospectral.py:155-335) against the reference's own
    output.  Converged quasi-Newton iterates agree to rounding; the fixed point is the
    isospectral midpoint step, so the result also agrees with isomp to the iteration tolerance."""
    g = load_golden("lu_steppers")
    pre = "N%d_" % N
    W0, steps, dt = g[pre + "W0"], int(g[pre + "steps"]), float(g[pre + tag + "_dt"])
    Ws = qfa.isomp_simple(W0.copy(), dt, steps)
    assert maxabs(Ws, g[pre + tag + "_simple"]) <= 1e-12
    stats = {}
    Wq = qfa.isomp_quasinewton(W0.copy(), dt, steps, stats=stats)
    assert maxabs(Wq, g[pre + tag + "_qn"]) <= 1e-12
    # (tol='auto' is eps*stepsize*|W|: a rounding-level tolerance, the loop may run to maxit)
    assert 1.0 <= stats["iterations"] <= 10.0 and 0.0 <= stats["number_of_maxit"] <= 1.0
    assert maxabs(qfa.isomp_quasinewton(W0.copy(), dt, steps, tol=1e-10), g[pre + tag + "_qn_tol1e10"]) <= 1e-9
    assert maxabs(Wq, g[pre + tag + "_isomp"]) <= 1e-6
    # isospectral to rounding: spectrum drift no worse than the reference's on the same input
    spec = np.linalg.eigvalsh(1j * Wq)
    drift = np.abs(spec - g[pre + tag + "_spec0"]).max()
    drift_ref = np.abs(g[pre + tag + "_spec_qn"] - g[pre + tag + "_spec0"]).max()
    assert drift <= max(2 * drift_ref, 1e-13)


@pytest.mark.parametrize("N", [256, 1024])
def test_lu_steppers_vs_oracle_large(qfa, oracle, N):
    W0 = oracle.make_W0(N, 6)
    dt = 0.25 * qfa.hbar(N)
    steps = 3 if N == 1024 else 6
    assert maxabs(qfa.isomp_simple(W0.copy(), dt, steps), oracle.isomp_simple(W0.copy(), dt, steps)) <= 1e-12
    so, sg = {}, {}
    Wc = oracle.isomp_quasinewton(W0.copy(), dt, steps, stats=so)
    Wg = qfa.isomp_quasinewton(W0.copy(), dt, steps, stats=sg)
    assert maxabs(Wg, Wc) <= 1e-12
    # the exit test compares a rounding-level residual (|Wt - Wt_new|_inf, a few eps |W|) with a
    # rounding-level tolerance (eps * stepsize * |W|_inf, isospectral.py:190-191): how many extra passes it
    # takes until the noise happens to fall below it depends on the last bits of the solve and the
    # products (LU vs Newton-Schulz, scan order of the Thomas carries), not on the method.  What must
    # agree is the state (above) and that both leave the loop by convergence, not by maxit.
    assert abs(sg["iterations"] - so["iterations"]) <= 2.0
    assert sg["number_of_maxit"] == 0.0 and so["iterations"] < 10


def test_lu_steppers_argument_checks(qfa):
    W = qfa.ensemble.make_W0(8, 0)
    with pytest.raises(ValueError):
        qfa.isomp_simple(np.zeros((4, 8), dtype=complex), 0.1, 1)
    with pytest.raises(TypeError):
        qfa.isomp_quasinewton([[0.0]], 0.1, 1)
    with pytest.raises(TypeError):             # the reference compares tol with a number: 'loose' < 0 is a TypeError there too
        qfa.isomp_quasinewton(W.copy(), 0.1, 1, tol="loose")
    with pytest.raises(TypeError):
        qfa.isomp(W.copy(), 0.1, 1, tol="loose")


@pytest.mark.parametrize("N", [16, 33, 64])
def test_states_golden(qfa, N):
    """magmp on a (2,N,N) state (quflow/integrators/mhd.py:235-456) and isomp on a (3,N,N) stack
    (isospectral.py 3-D branches) on the device against the reference's own output: same results,
    same iteration counts, same automatic tolerance."""
    g = load_golden("states")
    pre = "N%d_" % N
    steps, dt = int(g[pre + "steps"]), float(g[pre + "dt"])
    st = {"iterations": 0.0}
    W0 = g[pre + "state0"].copy()
    W = qfa.magmp(W0, dt, steps, stats=st)
    assert W is W0
    # (white-noise Theta: B = Delta Theta is large and the state grows; tolerance relative to its size)
    assert maxabs(W, g[pre + "magmp"]) <= 1e-12 * max(1.0, np.abs(g[pre + "magmp"]).max())
    assert st["iterations"] == float(g[pre + "magmp_iterations"])
    assert st["maxit"] == float(g[pre + "magmp_maxit"])
    np.testing.assert_allclose(st["tol"], float(g[pre + "magmp_tol"]), rtol=1e-14)
    st = {"iterations": 0.0}
    W = qfa.magmp(g[pre + "state0"].copy(), 2 * dt, steps, stats=st, tol=1e-11, minit=2, reinitialize=True)
    assert maxabs(W, g[pre + "magmp_opts"]) <= 1e-12 * max(1.0, np.abs(g[pre + "magmp_opts"]).max())
    assert st["iterations"] == float(g[pre + "magmp_opts_iterations"])
    st = {"iterations": 0.0}
    W = qfa.isomp(g[pre + "stack0"].copy(), 2.5 * dt, steps, stats=st)
    assert maxabs(W, g[pre + "isomp_stack"]) <= 1e-12
    assert st["iterations"] == float(g[pre + "isomp_stack_iterations"])
    np.testing.assert_allclose(st["tol_auto"], float(g[pre + "isomp_stack_tol"]), rtol=1e-14)


def test_states_vs_oracle_large(qfa, oracle):
    """N=1024 (64x64-tile products, upper-triangle second products): magmp and a 2-stack isomp."""
    N = 1024
    state = np.stack([oracle.make_W0(N, 1), oracle.solve_poisson(oracle.make_W0(N, 2)).copy()])
    dt = 0.1 * qfa.hbar(N)
    so, sg = {"iterations": 0.0}, {"iterations": 0.0}
    Wc = oracle.magmp_fixedpoint(state.copy(), dt, 3, stats=so)
    Wg = qfa.magmp(state.copy(), dt, 3, stats=sg)
    assert maxabs(Wg, Wc) <= 1e-12 * max(1.0, np.abs(Wc).max())
    assert sg["iterations"] == so["iterations"]
    assert np.array_equal(Wg[0], -Wg[0].conj().T) and np.array_equal(Wg[1], -Wg[1].conj().T)
    so, sg = {"iterations": 0.0}, {"iterations": 0.0}
    Wc = oracle.isomp(state.copy(), 2.5 * dt, 3, stats=so)
    Wg = qfa.isomp(state.copy(), 2.5 * dt, 3, stats=sg)
    assert maxabs(Wg, Wc) <= 1e-12
    assert sg["iterations"] == so["iterations"]
    # state 0 of the stack evolves exactly like a single trajectory
    W1 = qfa.isomp(state[0].copy(), 2.5 * dt, 3)
    assert maxabs(Wg[0], W1) <= 1e-13


def test_states_reject_unsupported(qfa):
    st = np.stack([qfa.ensemble.make_W0(8, 0), qfa.ensemble.make_W0(8, 1)])
    with pytest.raises(ValueError):
        qfa.magmp(st[0].copy(), 0.1, 1)
    with pytest.raises(AssertionError):
        qfa.magmp(st.copy(), 0.1, 1, minit=0)


def test_isomp_spot_golden(qfa):
    g = load_golden("isomp_spot")
    for N in (128, 256, 512):
        W0 = qfa.ensemble.make_W0(N, 0)
        stats = {"iterations": 0.0}
        W = qfa.isomp(W0.copy(), 0.25 * qfa.hbar(N), steps=int(g["N%d_steps" % N]), stats=stats)
        pre = "N%d_" % N
        assert stats["iterations"] == float(g[pre + "iterations"])
        np.testing.assert_allclose(stats["tol_auto"], float(g[pre + "tol_auto"]), rtol=1e-12)
        np.testing.assert_allclose(np.linalg.norm(W, "fro"), float(g[pre + "fro"]), rtol=1e-13)
        np.testing.assert_allclose(qfa.energy_euler(W), float(g[pre + "energy"]), rtol=1e-10)
        np.testing.assert_allclose(qfa.enstrophy(W), float(g[pre + "enstrophy"]), rtol=1e-12)
        if N == 128:
            assert maxabs(W, g[pre + "W"]) <= STEP_TOL
        else:
            assert maxabs(W[::8, ::8], g[pre + "W_s8"]) <= STEP_TOL
            np.testing.assert_allclose(np.abs(W).sum(axis=1), g[pre + "rowsum"], rtol=1e-11)


@pytest.mark.parametrize("N,steps", [(1024, 1), (1024, 2), (2048, 1)])
def test_isomp_spot_headline_golden(qfa, N, steps):
    """The device stepper at the headline sizes against the REFERENCE's own output (SURVEY.md 8c F7;
    tests/golden/isomp_spot_headline.npz from oracle/gen_golden.py gen_spot_headline: isospectral.py:338-613 run at
    N = 1024 for one step and for a two-step call, at N = 2048 for one step; dt = 0.25 hbar, IC-A)."""
    g = load_golden("isomp_spot_headline")
    pre = "N%d_s%d_" % (N, steps)
    W0 = qfa.ensemble.make_W0(N, 0)
    stats = {"iterations": 0.0}
    W = qfa.isomp(W0.copy(), 0.25 * qfa.hbar(N), steps=steps, stats=stats)
    assert stats["iterations"] == float(g[pre + "iterations"])
    assert stats["number_of_maxit"] == float(g[pre + "number_of_maxit"])
    np.testing.assert_allclose(stats["tol_auto"], float(g[pre + "tol_auto"]), rtol=1e-12)
    assert maxabs(W[::8, ::8], g[pre + "W_s8"]) <= STEP_TOL
    assert maxabs(np.diagonal(W), g[pre + "W_diag"]) <= STEP_TOL
    assert maxabs(W[100], g[pre + "W_row100"]) <= STEP_TOL
    # the increment of the call, where the two products show (W itself is W0 + O(dt))
    assert maxabs((W - W0)[::8, ::8], g[pre + "dW_s8"]) <= 1e-13
    np.testing.assert_allclose(np.abs(W).sum(axis=1), g[pre + "rowsum"], rtol=1e-11)
    np.testing.assert_allclose(np.abs(W - W0).sum(axis=1), g[pre + "dW_rowsum"], rtol=1e-8)
    np.testing.assert_allclose(np.linalg.norm(W, "fro"), float(g[pre + "fro"]), rtol=1e-13)
    np.testing.assert_allclose(qfa.energy_euler(W), float(g[pre + "energy"]), rtol=1e-10)
    np.testing.assert_allclose(qfa.enstrophy(W), float(g[pre + "enstrophy"]), rtol=1e-12)
    assert np.array_equal(W, -W.conj().T)


@pytest.mark.parametrize("N,steps", [(512, 3), (1024, 2), (2048, 1)])
def test_isomp_smooth_initial_data_vs_oracle_large(qfa, oracle, N, steps):
    """The smoothed initial condition IC-B of SURVEY.md 8(d) (normalize(solve_poisson(make_W0)): a larger stream function,
    ~7 fixed-point iterations per step instead of 2 -- the notebook's regime) at the BASELINE sizes against the oracle:
    long iteration sequences inside one step (warm-started increments, the stagnation arm of the exit rule) on every
    product kernel family (32 x 32 + tri32, 64 x 64 + stream-K at one and at two tiles per workgroup)."""
    W0 = oracle.make_W0_smooth(N, 0)
    dt = 0.25 * qfa.hbar(N)
    sg, sc = {"iterations": 0.0}, {"iterations": 0.0}
    Wg = qfa.isomp(W0.copy(), dt, steps=steps, stats=sg)
    Wc = oracle.isomp(W0.copy(), dt, steps=steps, stats=sc)
    assert sg["iterations"] == sc["iterations"] and sg["iterations"] >= 5.0, (sg, sc)
    assert sg["number_of_maxit"] == sc["number_of_maxit"]
    np.testing.assert_allclose(sg["tol_auto"], sc["tol_auto"], rtol=1e-12)
    assert maxabs(Wg, Wc) <= STEP_TOL
    assert np.array_equal(Wg, -Wg.conj().T)
    np.testing.assert_allclose(qfa.energy_euler(Wg), oracle.energy_euler(Wc), rtol=1e-10)


@pytest.mark.parametrize("N,steps", [(512, 6), (1024, 3), (768, 3), (800, 3), (896, 2), (1000, 2), (1056, 2), (1088, 2), (1536, 1), (2048, 1)])
def test_isomp_vs_oracle_large(qfa, oracle, N, steps):
    """BASELINE.json configs 2-3 sizes against the oracle on identical W0 (few steps: the
    oracle costs ~0.1-0.3 s per fixed-point iteration here) -- and the sizes on either side of every switch of the
    product kernels' selection: 32x32 tiles + tri32 (768, 800, 896, 1056: multiples of 32, with and without a 64x64
    alternative), the generic path (1000), 32x32 first product + stream-K second (1088, 1536)."""
    W0 = oracle.make_W0(N, 0)
    dt = 0.25 * qfa.hbar(N)
    sg, sc = {"iterations": 0.0}, {"iterations": 0.0}
    Wg = qfa.isomp(W0.copy(), dt, steps=steps, stats=sg)
    Wc = oracle.isomp(W0.copy(), dt, steps=steps, stats=sc)
    assert maxabs(Wg, Wc) <= STEP_TOL
    assert sg["iterations"] == sc["iterations"]
    np.testing.assert_allclose(sg["tol_auto"], sc["tol_auto"], rtol=1e-12)
    # Casimirs C_k = tr((iW)^k)/N conserved no worse than the CPU path
    c0 = oracle.casimirs(W0)
    dg = np.abs(oracle.casimirs(Wg) - c0).max()
    dc = np.abs(oracle.casimirs(Wc) - c0).max()
    # drift <= the CPU run's on the same input.  5 % slack plus the resolution of the instrument the
    # drifts are read with (matrix powers in fp64: sqrt(N) eps |W|_2, the rule test_config3_... uses) -- no floor.
    res = np.sqrt(N) * EPS * float(np.abs(oracle.spectrum(Wc)).max())
    assert dg <= 1.05 * dc + res, (dg, dc, res)


def _spectrum_resolution(oracle, W):
    """What the eigensolver the spectrum drifts are read with resolves on THIS matrix, measured: the eigenvalues of iW
    against those of the same matrix with rows and columns in reverse order -- the same spectrum through another
    elimination order, i.e. a sample of the solver's own rounding.  (Round 5: 2.7e-14 ... 3.8e-14 at N = 1024, two to three
    times the sqrt(N) eps |W|_2 rule of thumb the round-3 tests used; profiles/r05_drift_statistic_noise_*.jsonl: ONE device
    state read in two processes with different BLAS pools gave 5.64e-13 and 6.02e-13 for the same drift.)"""
    return float(np.abs(oracle.spectrum(W) - oracle.spectrum(np.ascontiguousarray(W[::-1, ::-1]))).max())


def _run_int8_products(qfa, oracle, N, steps, products, monkeypatch):
    from quflow_amd.context import release_contexts
    monkeypatch.setenv("QUFLOW_HIP_GEMM", products)
    monkeypatch.setenv("QUFLOW_HIP_I8_MIN_N", "64")
    release_contexts()
    try:
        W0 = oracle.make_W0(N, 0)
        dt = 0.25 * qfa.hbar(N)
        sg, sc = {"iterations": 0.0}, {"iterations": 0.0}
        Wg = qfa.isomp(W0.copy(), dt, steps=steps, stats=sg)
        Wc = oracle.isomp(W0.copy(), dt, steps=steps, stats=sc)
    finally:
        release_contexts()
    spec0 = oracle.spectrum(W0)
    cas0 = oracle.casimirs(W0)
    return {"Wg": Wg, "Wc": Wc, "its_g": sg["iterations"], "its_c": sc["iterations"],
            "spec_res": _spectrum_resolution(oracle, Wg) + _spectrum_resolution(oracle, Wc),
            "spec_g": np.abs(oracle.spectrum(Wg) - spec0).max(), "spec_c": np.abs(oracle.spectrum(Wc) - spec0).max(),
            "cas_g": np.abs(oracle.casimirs(Wg) - cas0).max(), "cas_c": np.abs(oracle.casimirs(Wc) - cas0).max()}


@pytest.mark.parametrize("N,steps,products", [(N_, s_, p_) for (N_, s_) in ((256, 10), (1024, 40)) for p_ in ("i8x6", "i8x6f", "i8x65")]
                         + [(2048, 1, "i8x65")])      # (the size bench.py reports config 3 at: four tiles per CU)
def test_config3_lowprecision_commutator_vs_oracle(qfa, oracle, N, steps, products, monkeypatch):
    """BASELINE.json config 3 ("N=1024 low-precision-MFMA commutator with fp64 Laplacian, Casimir-drift
    tolerance check vs CPU"): both products on the int8 matrix cores by digit splitting with SIX base-128
    digits (QUFLOW_HIP_GEMM=i8x6; DESIGN.md 3.6 on why int8 digits stand in for bf16 pieces), Laplacian
    inverse in fp64; `i8x6f` (round 4): the FIRST product -- the full one, which the commutator is read from -- on the
    int8 matrix cores, the second on the fp64 upper-triangle kernel; `i8x65`: six digits for the first product, five for
    the second (the O(|Phalf|)-smaller term), both on the int8 matrix cores.  Acceptance = the fp64 path's own bars against the CPU oracle on the same W0:
    identical iteration counts, state within STEP_TOL, and spectrum / Casimir drift no worse than the
    CPU run's (5 % slack, plus the resolution of the eigensolver the drifts are read with) -- no tuned floor."""
    r = _run_int8_products(qfa, oracle, N, steps, products, monkeypatch)
    assert r["its_g"] == r["its_c"]
    assert maxabs(r["Wg"], r["Wc"]) <= STEP_TOL
    assert np.array_equal(r["Wg"], -r["Wg"].conj().T)
    # Both drifts are READ with instruments of finite resolution: a difference below it is not a difference in drift.
    # Spectrum: the eigensolver's, measured on the two states themselves (_spectrum_resolution); Casimirs: matrix powers
    # in fp64, sqrt(N) eps |W|_2 (|W|_2 ~ 2 for make_W0).  Resolutions derived from the data, not tuned floors.
    res = np.sqrt(N) * EPS * float(np.abs(oracle.spectrum(r["Wc"])).max())
    assert r["spec_g"] <= 1.05 * r["spec_c"] + r["spec_res"], (r["spec_g"], r["spec_c"], r["spec_res"])
    assert r["cas_g"] <= 1.05 * r["cas_c"] + res, (r["cas_g"], r["cas_c"], res)


@pytest.mark.parametrize("N,steps,products", [(256, 10, "i8h"), (1024, 40, "i8h"), (256, 10, "i8hx6")])
def test_hybrid_products_meet_the_fp64_bars(qfa, oracle, N, steps, products, monkeypatch):
    """Round 3: the hybrid -- first product on the fp64 matrix cores, only the second one
    (T = PW @ Phalf, O(|Phalf|) smaller than the commutator term it is added to) digit-split -- meets config 3's
    acceptance (the fp64 path's own bars against the CPU oracle) already with FIVE digits: measured at N=1024, 40
    steps: state within 4e-17 of the oracle's (the fp64 products' own distance), drift equal to the CPU run's."""
    r = _run_int8_products(qfa, oracle, N, steps, products, monkeypatch)
    assert r["its_g"] == r["its_c"]
    assert maxabs(r["Wg"], r["Wc"]) <= STEP_TOL
    assert np.array_equal(r["Wg"], -r["Wg"].conj().T)
    res = np.sqrt(N) * EPS * float(np.abs(oracle.spectrum(r["Wc"])).max())
    assert r["spec_g"] <= 1.05 * r["spec_c"] + r["spec_res"], (r["spec_g"], r["spec_c"], r["spec_res"])
    assert r["cas_g"] <= 1.05 * r["cas_c"] + res


@pytest.mark.parametrize("N,steps", [(256, 10), (1024, 4)])
def test_five_digit_int8_products_demonstration(qfa, oracle, N, steps, monkeypatch):
    """NOT config 3's acceptance line: the 5-digit variant (QUFLOW_HIP_GEMM=i8) cuts the digit series
    at 2^-35 ~ 3e-11 relative per product, so its state sits ~1e-10 per step off the fp64 run and its
    spectrum drifts at the 1e-11 level (8x the fp64 run's at N=1024) -- below the stepper's own
    sqrt(eps) fixed-point tolerance, above the reference's drift.  Kept as the speed demonstration
    (1.5x); these bounds only document where it lands."""
    r = _run_int8_products(qfa, oracle, N, steps, "i8", monkeypatch)
    assert abs(r["its_g"] - r["its_c"]) <= 0.05 * r["its_c"]
    assert maxabs(r["Wg"], r["Wc"]) <= I8_TOL
    assert np.array_equal(r["Wg"], -r["Wg"].conj().T)
    assert r["spec_g"] <= max(1.05 * r["spec_c"], I8_DRIFT)
    assert r["cas_g"] <= max(1.05 * r["cas_c"], I8_DRIFT)


@pytest.mark.parametrize("N", [192, 320, 448, 1088])
def test_isomp_i8_odd_tile_counts(qfa, N, monkeypatch):
    """The int8 products at sizes whose tile grid is not a multiple of the XCD blocking (3, 5, 7, 17
    tiles per side): mirrored second product, XCD-contiguous order, three LDS stages with few
    K-steps -- the same trajectory as the fp64 products."""
    from quflow_amd.context import release_contexts
    W0 = qfa.ensemble.make_W0(N, 2)
    dt = 0.25 * qfa.hbar(N)
    sf = {"iterations": 0.0}
    Wf = qfa.isomp(W0.copy(), dt, steps=6, stats=sf)
    for products, tol in (("i8x6", STEP_TOL), ("i8x65", STEP_TOL), ("i8", I8_TOL)):
        monkeypatch.setenv("QUFLOW_HIP_GEMM", products)
        monkeypatch.setenv("QUFLOW_HIP_I8_MIN_N", "64")
        release_contexts()
        try:
            s8 = {"iterations": 0.0}
            W8 = qfa.isomp(W0.copy(), dt, steps=6, stats=s8)
        finally:
            release_contexts()
        assert maxabs(W8, Wf) <= tol, products
        assert np.array_equal(W8, -W8.conj().T)
        if products != "i8":
            assert s8["iterations"] == sf["iterations"]
    monkeypatch.delenv("QUFLOW_HIP_GEMM")
    release_contexts()


@pytest.mark.parametrize("products", ["f64", "i8", "i8x6", "i8x65"])
def test_isomp_full_size_properties(qfa, products, monkeypatch):
    """N=2048 (config 5) through size-independent properties: skew-Hermitian, trace-free,
    enstrophy and spectrum conserved, fixed-iteration count respected -- with the fp64 products and
    with the digit-split int8 products (four 64 x 64 tiles per CU)."""
    from quflow_amd.context import release_contexts
    N = 2048
    W0 = qfa.ensemble.make_W0(N, 0)
    stats = {"iterations": 0.0}
    if products != "f64":
        monkeypatch.setenv("QUFLOW_HIP_GEMM", products)
    release_contexts()
    try:
        W = qfa.isomp(W0.copy(), 0.25 * qfa.hbar(N), steps=4, stats=stats)
    finally:
        release_contexts()
    # (tol='auto' is eps*stepsize*|W|: a rounding-level tolerance, the loop may run to maxit)
    assert 1.0 <= stats["iterations"] <= 10.0 and 0.0 <= stats["number_of_maxit"] <= 1.0
    assert maxabs(W, -W.conj().T) <= 1e-14
    # (round 5: the int8 first product stores Im (Phalf @ Whalf)_ii from an fp64 row product of the slicing launch, so the
    # trace of the commutator is the fp64 products' to rounding -- with any number of digits)
    assert abs(np.trace(W)) <= 1e-14
    assert abs(np.linalg.norm(W, "fro") ** 2 / (2 * N) - 0.5) <= (1e-11 if products == "i8" else 1e-12)
    ev0 = np.linalg.eigvalsh(1j * W0)
    ev = np.linalg.eigvalsh(1j * W)
    assert np.abs(ev - ev0).max() <= 5e-10
    assert maxabs(W, W0) > 1e-6      # it actually moved
    if products != "f64":            # and the same trajectory as the fp64 products give
        Wf = qfa.isomp(W0.copy(), 0.25 * qfa.hbar(N), steps=4)
        assert maxabs(W, Wf) <= (I8_TOL if products == "i8" else STEP_TOL)



def test_modulus_is_the_compilers_square_root(qfa):
    """qf_modulus (qf_internal.h) -- the square root every residual row sum of the stepper is formed with -- is hipcc's own
    sqrt expansion without its range scaling.  The exit test compares sums of it with the tolerance, and the iteration
    counts are asserted equal to the reference's, so the claim "same bits as sqrt()" is pinned here: random arguments over
    the whole normal range the residuals can meet, exact squares, and the edges (0, inf, NaN propagate; below 2^-767 the
    result is allowed to differ -- documented -- but must stay finite and non-negative)."""
    import ctypes
    from quflow_amd import _lib
    from quflow_amd.context import get_context
    ctx = get_context(256)
    rng = np.random.default_rng(5)
    n = 40000
    mag = 10.0 ** rng.uniform(-150, 150, size=n)
    er = rng.standard_normal(n) * mag
    ei = rng.standard_normal(n) * mag * 10.0 ** rng.uniform(-3, 3, size=n)
    er[:8] = [0.0, 3.0, 5.0, 1e-170, 0.0, np.inf, np.nan, 1e154]
    ei[:8] = [0.0, 4.0, 12.0, 0.0, 2.0 ** -400, 1.0, 1.0, 1e154]
    om, os_ = np.empty(n), np.empty(n)
    dp = ctypes.POINTER(ctypes.c_double)
    _lib.check(ctx._lib.qf_debug_modulus(ctx.handle, n, er.ctypes.data_as(dp), ei.ctypes.data_as(dp), om.ctypes.data_as(dp),
                                         os_.ctypes.data_as(dp)))
    with np.errstate(over="ignore", invalid="ignore", under="ignore"):
        x = er * er + ei * ei
    normal = (x > 2.0 ** -700) & np.isfinite(x)
    assert normal.sum() > n // 2
    np.testing.assert_array_equal(om[normal].view(np.uint64), os_[normal].view(np.uint64))      # bit for bit
    assert om[0] == 0.0 and om[1] == 5.0 and om[2] == 13.0 and om[5] == np.inf and np.isnan(om[6])
    tiny = (x <= 2.0 ** -700) & (x > 0)
    assert np.all(np.isfinite(om[tiny])) and np.all(om[tiny] >= 0.0)
    # and the host's square root of the host's argument, to the last place (the device may contract er er + ei ei)
    with np.errstate(over="ignore", invalid="ignore", under="ignore"):
        fine = normal & (x < 1e300) & (x > 1e-290)
        assert np.all(np.abs(om[fine] - np.sqrt(x[fine])) <= 2 * np.spacing(om[fine]))


def test_plan_describe_names_what_was_launched(qfa, monkeypatch):
    """qf_plan_describe (round 5): the launchers record what they launch for each role of the hot path; bench.py labels
    its roofline object with that instead of a copy of the selection rules.  The three BASELINE sizes, config 3's
    products and a complex64 state: kernel names, tile shares and workgroup counts as the kernels' own launch code
    set them."""
    from quflow_amd.context import release_contexts
    cases = [(512, None, np.complex128, "k_zgemm<32,32>", "k_zgemm_tri32<exact>", 136 / 256),
             (1024, None, np.complex128, "k_zgemm<64,64>", "k_zgemm_tri", 136 / 256),
             (2048, None, np.complex128, "k_zgemm<64,64>", "k_zgemm_tri", 528 / 1024),
             (1024, "i8x65", np.complex128, "k_oz_gemm<6,plain,6>", "k_oz_gemm<5,fused,6>", 136 / 256),
             (1024, None, np.complex64, None, "k_cgemm_tri32 (upper triangle, K pieces per tile)", 528 / 1024)]
    for N, products, dtype, first, second, share in cases:
        if products:
            monkeypatch.setenv("QUFLOW_HIP_GEMM", products)
        else:
            monkeypatch.delenv("QUFLOW_HIP_GEMM", raising=False)
        tr = qfa.DeviceTrajectory(qfa.ensemble.make_W0(N, 0).astype(dtype))
        empty = tr.ctx.plan()
        assert empty["N"] == N and empty["first_product"] is None and empty["second_product"] is None
        tr.advance(0.25 * qfa.hbar(N), 2)
        plan = tr.ctx.plan()
        tr.ctx.close()
        assert plan["laplacian_inverse"]["kernel"].startswith("k_solve<%s" % ("float" if dtype == np.complex64 else "double"))
        if first:
            assert plan["first_product"]["kernel"] == first, plan["first_product"]
        assert plan["first_product"]["tile_share"] == 1.0
        assert plan["second_product"]["kernel"] == second, plan["second_product"]
        assert abs(plan["second_product"]["tile_share"] - share) < 1e-6
        assert plan["second_product"]["workgroups"] >= 1 and plan["second_product"]["threads"] in (256, 512)
        assert (plan["slicing"] is not None) == bool(products)
    release_contexts()


def test_randomised_options_against_the_oracle(qfa, oracle):
    """A seeded batch of tests/fuzz_stepper_vs_oracle.py: random sizes (even / odd / around the 32- and 64-tile edges), step
    sizes, step counts and option combinations (tol, minit / maxit, compsum, reinitialize, time, stacks, forcing / Strang /
    callback hooks) -- state within 2e-11, identical iteration statistics, tol_auto and callback counts.  Round 5 ran 140
    cases of the full size list (up to N = 320) with no disagreement (profiles/r05_fuzz_stepper_vs_oracle.txt)."""
    import fuzz_stepper_vs_oracle as fz
    assert fz.main(cases=120, seed=5, sizes=[2, 3, 5, 8, 16, 17, 31, 32, 33, 48, 63, 64, 65, 96, 100, 127, 128, 129], quiet=True) == 0


def test_randomised_backends_and_steppers_against_the_oracle(qfa, oracle):
    """A seeded batch of tests/fuzz_backends_vs_oracle.py: solve_poisson (complex128 / complex64, states with a trace, general
    matrices), laplace, helmholtz / heat / viscdamp / globalqg, euler / heun / rk4 (with forcing), isomp_simple,
    isomp_quasinewton, magmp and isomp on complex64 states at random sizes around the tile and chunk edges.  Round 5 ran 800
    cases up to N = 257 (profiles/r05_fuzz_backends_vs_oracle.txt)."""
    import fuzz_backends_vs_oracle as fz
    assert fz.main(cases=150, seed=11, sizes=[2, 3, 5, 8, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 96, 100, 127, 128, 129], quiet=True) == 0


def test_randomised_chunk_chains_against_the_oracle(qfa, oracle):
    """A seeded batch of tests/fuzz_trajectory_vs_oracle.py: a DeviceTrajectory advanced in random chunks (each chunk a stepper
    call that starts from dW = 0, as the reference's does) against the same chain of oracle calls -- iteration counts per chunk,
    diagnostics after every chunk, the final state -- and bit for bit against the same chain through isomp() on host arrays.
    Round 5 ran 60 cases up to N = 1024 (profiles/r05_fuzz_trajectory_vs_oracle.txt)."""
    import fuzz_trajectory_vs_oracle as fz
    assert fz.main(cases=12, seed=21, sizes=[48, 64, 96, 100, 128, 160], quiet=True) == 0


def test_randomised_config3_products_against_the_oracle(qfa, oracle):
    """A seeded batch of tests/fuzz_config3_vs_oracle.py: the int8 digit-split products (i8x65 / i8x6, let in from N = 64) on white
    AND smooth initial data against the oracle's fp64 run -- identical iteration counts, W exactly skew-Hermitian, state within
    1e-11 (2e-11 on smooth data, 1e-10 where steps end by maxit) and |tr W| at the fp64 run's level.  The smooth cases are the ones
    that exposed round 5's second trace leak (diagonal tiles of the five-digit second product were skew-Hermitian only to the
    truncation: profiles/r05_fuzz_config3_vs_oracle.txt); before the fix they read |tr W| up to 4e-11 after two steps."""
    import fuzz_config3_vs_oracle as fz
    assert fz.main(cases=24, seed=13, sizes=[64, 128, 192, 256, 320, 384, 448], quiet=True) == 0


@pytest.mark.parametrize("N", [64, 512, 1024])
def test_nonfinite_state_raises_as_the_reference_does(qfa, oracle, N):
    """A state with a NaN or an inf in it: the reference's exit test hands the residual to scipy.linalg.norm, which checks its
    argument and raises ValueError("array must not contain infs or NaNs") (isospectral.py:534) -- found by looking at what the
    randomised runs never draw.  The device forms the same residual, flags it where it takes the exit decision (fused, deferred
    and two-kernel step end, the host loops of stacks and hooks, isomp_quasinewton) and the call returns QF_ERR_NONFINITE, which
    the Python mirror raises as the same ValueError; the caller's array is left alone and the context runs a clean state
    afterwards to the same bits as a fresh one."""
    dt = 0.25 * qfa.hbar(N)
    W0 = oracle.make_W0(N, 0)
    for poison in (np.nan, np.inf):
        Wb = W0.copy()
        Wb[3, 5] = poison
        Wb[5, 3] = poison
        for kw in ({}, {"tol": 1e-10}, {"compsum": True}):
            with pytest.raises(ValueError, match="infs or NaNs"):
                oracle.isomp(Wb.copy(), dt, steps=2, **kw)
            Wd = Wb.copy()
            with pytest.raises(ValueError, match="infs or NaNs"):
                qfa.isomp(Wd, dt, steps=2, **kw)
            assert np.array_equal(Wd[0], Wb[0]) and np.array_equal(Wd[7], Wb[7])      # not overwritten with a half-done state
    if N == 64:
        S = np.stack([Wb, W0])
        with pytest.raises(ValueError, match="infs or NaNs"):
            qfa.isomp(S.copy(), dt, steps=2)                                            # stacks: the host loop of qf_isomp_states
        with pytest.raises(ValueError, match="infs or NaNs"):
            qfa.isomp(Wb.copy(), dt, steps=2, forcing=lambda P, W: 0.0 * W)             # hooks: qf_isomp_hooked
        with pytest.raises(ValueError, match="infs or NaNs"):
            qfa.isomp_quasinewton(Wb.copy(), dt, 2)
        with pytest.raises(ValueError, match="infs or NaNs"):
            qfa.isomp(Wb.astype(np.complex64), dt, steps=2)                             # the float32 path shares the step end
    # the context survives: a clean state right afterwards, bit for bit what a fresh process computes (the suite's golden
    # and oracle tests run on the same cached contexts before and after this one)
    sg, sc = {"iterations": 0.0}, {"iterations": 0.0}
    Wg = qfa.isomp(W0.copy(), dt, steps=2, stats=sg)
    Wc = oracle.isomp(W0.copy(), dt, steps=2, stats=sc)
    assert maxabs(Wg, Wc) <= STEP_TOL and sg["iterations"] == sc["iterations"]
    tr = qfa.DeviceTrajectory(Wb)
    with pytest.raises(ValueError, match="infs or NaNs"):
        tr.advance(dt, 2)
    tr.upload(W0)
    tr.advance(dt, 2)
    np.testing.assert_array_equal(tr.download(), Wg)
    tr.ctx.close()


def test_device_info_names_the_bound_device(qfa):
    """qf_device_info (round 5): what a rank of `bench.py --gpus N` prints about the device it bound -- ordinal, PCI bus id in
    the dddd:bb:dd.f form bench.py packs into its all-gathered row, gfx950, the CU count the partitions are built for."""
    import re
    from quflow_amd import _lib
    info = qfa.device_info(0)
    assert info["ordinal"] == 0
    assert re.fullmatch(r"[0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-7]", info["pci_bus_id"]), info
    assert info["gcn_arch"].startswith("gfx9"), info
    assert isinstance(info["compute_units"], int) and info["compute_units"] >= 1 and info["memory_bytes"] > 1 << 30, info
    if info["gcn_arch"].startswith("gfx950"):      # the part this library is written for: an MI355X, whole or partitioned
        assert info["compute_units"] in (256, 128, 64, 32) and info["memory_bytes"] > 30 << 30, info
    assert isinstance(info["name"], str)          # (the marketing name; empty on some boxes of the pool)
    # a short buffer gets a truncated, terminated text and the full length back; a bad ordinal is an error with a message
    import ctypes
    lib = _lib.load()
    buf = ctypes.create_string_buffer(16)
    n = lib.qf_device_info(0, buf, 16)
    assert n > 16 and len(buf.value) == 15 and buf.value.startswith(b'{"ordinal": 0')
    assert lib.qf_device_info(qfa.device_count(), None, 0) < 0
    assert b"out of range" in lib.qf_last_error()


def test_geometry_and_physics_helpers(qfa, oracle):
    """bracket (device products), the L2 / Linf / L1 norms and the Sobolev inner products of
    quflow/geometry.py:41-129 and quflow/physics.py:9-21 against the oracle's solves and numpy."""
    N = 48
    W = qfa.ensemble.make_W0(N, 5)
    P = oracle.solve_poisson(W).copy()
    ref = (P @ W - W @ P) / qfa.hbar(N)
    assert maxabs(qfa.bracket(P, W), ref) <= 1e-13 * np.abs(ref).max() * N
    assert abs(qfa.norm_L2(W) - np.linalg.norm(W) / np.sqrt(N)) <= 1e-15
    assert abs(qfa.inner_L2(P, W) - (P * W.conj()).sum().real / N) <= 1e-16
    assert abs(qfa.norm_Linf(W) - np.linalg.norm(W, 2)) <= 1e-14
    assert abs(qfa.norm_L1(W) - np.abs(np.linalg.eigvals(W)).sum() / N) <= 1e-13
    assert abs(qfa.integral(W + 0.25j * np.eye(N)) - 0.25) <= 1e-15
    # Sobolev products: -<W, Delta^-1 W> = 2 E, and H1 of P = Hm1 of W for P = Delta^-1 W
    assert abs(qfa.inner_Hm1(W, W) - 2 * qfa.energy_euler(W)) <= 1e-15
    assert abs(qfa.norm_Hm1(W) - np.sqrt(-qfa.inner_L2(W, P))) <= 1e-14
    assert abs(qfa.inner_H1(P, P) - qfa.inner_Hm1(W, W)) <= 1e-13
    assert abs(qfa.norm_H1(P) - qfa.norm_Hm1(W)) <= 1e-13


# ----------------------------------------------------------------------------- host hooks
def _hook_forcing(P, W):
    return -0.05 * W + 0.02 * P


def _hook_forcing_t(P, W, time=0.0):
    return (-0.05 * np.cos(time)) * W + 0.02 * P


def test_isomp_hooks_golden(qfa):
    """strang_splitting / callback (device steps with qf_isomp_continue between the hooks) and
    forcing / a foreign Hamiltonian (the reference's loop on the host, products on the device)
    against vectors the reference produced with the same hooks (oracle/gen_golden.py:gen_hooks)."""
    g = load_golden("hooks")
    N = int(g["N"])
    W0 = g["W0"]
    dt = float(g["stepsize"]) * qfa.hbar(N)
    lap = qfa.laplacian

    def strang(h, W):
        return lap.solve_viscdamp(h, W, nu=1e-3, alpha=0.05)

    stats = {"iterations": 0.0}
    W = qfa.isomp(W0.copy(), dt, steps=12, strang_splitting=strang, stats=stats)
    assert maxabs(W, g["strang_W"]) <= STEP_TOL
    assert stats["iterations"] == float(g["strang_iterations"])
    # the same half step as a recognised object: applied to the resident state, no PCIe
    stats = {"iterations": 0.0}
    W = qfa.isomp(W0.copy(), dt, steps=12, strang_splitting=qfa.ViscDampStep(nu=1e-3, alpha=0.05), stats=stats)
    assert maxabs(W, g["strang_W"]) <= STEP_TOL
    assert stats["iterations"] == float(g["strang_iterations"])
    stats = {"iterations": 0.0}
    W = qfa.isomp(W0.copy(), dt, steps=6, strang_splitting=strang, compsum=True, stats=stats)
    assert maxabs(W, g["strang_compsum_W"]) <= STEP_TOL
    assert stats["iterations"] == float(g["strang_compsum_iterations"])

    rec = []

    def cb(W, dW):
        rec.append([np.linalg.norm(W), np.linalg.norm(dW), abs(np.trace(dW @ W))])
    stats = {"iterations": 0.0}
    Win = W0.copy()
    W = qfa.isomp(Win, dt, steps=8, callback=cb, stats=stats)
    assert W is Win
    assert maxabs(W, g["callback_W"]) <= STEP_TOL
    np.testing.assert_allclose(np.array(rec), g["callback_record"], rtol=1e-9, atol=1e-12)
    assert stats["iterations"] == float(g["callback_iterations"])
    # without hooks the same 8 steps give the same state: the stepwise path carries dW over
    assert maxabs(qfa.isomp(W0.copy(), dt, steps=8), W) <= 1e-15

    for tag, kw in (("forcing", {"forcing": _hook_forcing}),
                    ("forcing_t", {"forcing": _hook_forcing_t, "time": 0.3}),
                    ("foreign", {"hamiltonian": lambda W: 0.5 * lap.solve_poisson(W) + 0.1j * np.eye(N)})):
        stats = {"iterations": 0.0}
        W = qfa.isomp(W0.copy(), dt, steps=10, stats=stats, **kw)
        assert maxabs(W, g[tag + "_W"]) <= STEP_TOL, tag
        assert stats["iterations"] == float(g[tag + "_iterations"]), tag

    # the explicit steppers with the same hooks (erk.py:47-56, 93-112, 142-160)
    dte = 0.05 * qfa.hbar(N)
    foreign = lambda W: 0.5 * lap.solve_poisson(W) + 0.1j * np.eye(N)      # noqa: E731
    for name in ("euler", "heun", "rk4"):
        fn = getattr(qfa, name)
        Win = W0.copy()
        W = fn(Win, dte, steps=10, forcing=_hook_forcing)
        assert W is Win
        assert maxabs(W, g["erk_%s_forcing_W" % name]) <= 1e-13, name
        W = fn(W0.copy(), dte, steps=10, hamiltonian=foreign)
        assert maxabs(W, g["erk_%s_foreign_W" % name]) <= 1e-13, name


def test_isomp_hooks_on_stacks_and_general_branch_golden(qfa):
    """qf_isomp_hooked: the hooks, compsum and `reinitialize` on a (2,N,N) stack (P and the exit test from
    state 0, isospectral.py:527-532) and the general commutator of select_skewherm(False) (:504-505),
    against vectors the reference produced (oracle/gen_golden.py:gen_hooks_stack)."""
    g = load_golden("hooks_stack")
    N = int(g["N"])
    W0 = g["W0"]
    dt = float(g["stepsize"]) * qfa.hbar(N)
    lap = qfa.laplacian

    def strang(h, W):
        return np.stack([lap.solve_viscdamp(h, W[j], nu=1e-3, alpha=0.05) for j in range(W.shape[0])])

    def foreign(W):
        return 0.5 * lap.solve_poisson(W) + 0.1j * np.eye(W.shape[-1])

    rec = []

    def cb(W, dW):
        rec.append([np.linalg.norm(W), np.linalg.norm(dW), abs(np.trace(dW[1] @ W[0]))])

    cases = {"plain": {}, "compsum": {"compsum": True}, "forcing": {"forcing": _hook_forcing},
             "forcing_t": {"forcing": _hook_forcing_t, "time": 0.3}, "foreign": {"hamiltonian": foreign},
             "strang": {"strang_splitting": strang}, "strang_compsum": {"strang_splitting": strang, "compsum": True},
             "callback": {"callback": cb}, "reinit": {"reinitialize": True, "forcing": _hook_forcing}}
    for tag, kw in cases.items():
        stats = {"iterations": 0.0}
        Win = W0.copy()
        W = qfa.isomp(Win, dt, steps=6, stats=stats, **kw)
        assert W is Win
        assert maxabs(W, g[tag + "_W"]) <= STEP_TOL, tag
        assert stats["iterations"] == float(g[tag + "_iterations"]), tag
        if tag + "_tol" in g.files:
            np.testing.assert_allclose(stats["tol_auto"], float(g[tag + "_tol"]), rtol=1e-14)
    np.testing.assert_allclose(np.array(rec), g["callback_record"], rtol=1e-9, atol=1e-12)
    # the resident form of the viscous half step on a stack
    stats = {"iterations": 0.0}
    W = qfa.isomp(W0.copy(), dt, steps=6, strang_splitting=qfa.ViscDampStep(nu=1e-3, alpha=0.05), compsum=True, stats=stats)
    assert maxabs(W, g["strang_compsum_W"]) <= STEP_TOL
    assert stats["iterations"] == float(g["strang_compsum_iterations"])
    # compsum + forcing: the reference raises (isospectral.py:588-589)
    with pytest.raises(NotImplementedError):
        qfa.isomp(W0.copy(), dt, steps=2, compsum=True, forcing=_hook_forcing)
    # an exception inside a hook comes back as itself
    def bad(P, W):
        raise KeyError("from the hook")
    with pytest.raises(KeyError):
        qfa.isomp(W0[0].copy(), dt, steps=2, forcing=bad)

    # general branch: the integrators' own flag (it switches the Laplacian backend as well)
    qfa.integrators.select_skewherm(False)
    try:
        for tag, kw in (("general", {}), ("general_forcing", {"forcing": _hook_forcing}), ("general_compsum", {"compsum": True})):
            stats = {"iterations": 0.0}
            W = qfa.isomp(g["general_W0"].copy(), dt, steps=6, stats=stats, **kw)
            assert maxabs(W, g[tag + "_W"]) <= STEP_TOL, tag
            assert stats["iterations"] == float(g[tag + "_iterations"]), tag
        stats = {"iterations": 0.0}
        W = qfa.isomp(g["general_stack_W0"].copy(), dt, steps=4, stats=stats)
        assert maxabs(W, g["general_stack_W"]) <= STEP_TOL
        assert stats["iterations"] == float(g["general_stack_iterations"])
    finally:
        qfa.integrators.select_skewherm(True)
    assert lap._SKEW_HERM_ is True


def test_hooks_may_use_the_shared_context(qfa):
    """A callback / strang hook that calls host-in/host-out entry points at the SAME N (energy_euler,
    solve_poisson stage through the shared per-N context) must not disturb the trajectory the stepper
    keeps resident between the hooks (it lives in a context of its own)."""
    N = 64
    W0 = qfa.ensemble.make_W0(N, 3)
    dt = 0.25 * qfa.hbar(N)
    seen = []

    def cb(W, dW):
        seen.append((qfa.energy_euler(W), qfa.enstrophy(W)))
        qfa.solve_poisson(0.5 * W)

    stats = {"iterations": 0.0}
    Wa = qfa.isomp(W0.copy(), dt, steps=6, callback=cb, stats=stats)
    Wb = qfa.isomp(W0.copy(), dt, steps=6)
    assert len(seen) == 6
    assert maxabs(Wa, Wb) <= 1e-15
    # the same with a resident viscous half step between the device steps
    def cb2(W, dW):
        qfa.energy_euler(W)
    Wc = qfa.isomp(W0.copy(), dt, steps=4, strang_splitting=qfa.ViscDampStep(nu=1e-3, alpha=0.05), callback=cb2)
    Wd = qfa.isomp(W0.copy(), dt, steps=4, strang_splitting=qfa.ViscDampStep(nu=1e-3, alpha=0.05))
    assert maxabs(Wc, Wd) <= 1e-15


@pytest.mark.parametrize("N,k", [(64, 4), (512, 4), (1024, 3), (128, 6), (96, 9)])
def test_device_ensemble_members_are_bit_identical_to_single_runs(qfa, N, k):
    """k replicas advanced together on one GPU (qf_isomp_multi): every member equals its own
    single-trajectory run bit for bit -- state, iteration counts, tolerance -- over chunked calls.  More than four
    members go through a call in groups of four (one per hardware pipe, DeviceEnsemble.CONCURRENT): k = 6 and 9."""
    W0s = [qfa.ensemble.make_W0(N, 40 + r) for r in range(k)]
    dt = 0.25 * qfa.hbar(N)
    steps = 6 if N >= 512 else 20
    ens = qfa.DeviceEnsemble(W0s)
    st_a = ens.advance(dt, steps)
    st_b = ens.advance(dt, steps)              # a second chunk: dW restarts from zero as in qf_isomp
    got = ens.download()
    diag = ens.diagnostics()
    ens.close()
    for r in range(k):
        tr = qfa.DeviceTrajectory(W0s[r])
        s1 = tr.advance(dt, steps)
        s2 = tr.advance(dt, steps)
        np.testing.assert_array_equal(got[r], tr.download())
        assert (st_a[r]["total_iterations"], st_b[r]["total_iterations"]) == (s1["total_iterations"], s2["total_iterations"])
        assert st_a[r]["tol"] == s1["tol"] and st_b[r]["tol"] == s2["tol"]
        assert diag[r] == tr.diagnostics()
        tr.ctx.close()


def test_solve_driver_device_resident_and_restart(qfa, tmp_path):
    """quflow_amd.simulation.solve with the default stepper keeps the trajectory on the device between the
    output chunks: same rows as chunked qfa.isomp calls on host arrays, bit for bit; 50 + 50 steps through a
    re-opened record equal 100 straight (tests/test_simulation.py:130-168); 'shr' rows come from the device
    transform of the resident state."""
    from quflow_amd.simulation import Simulation, solve
    N = 64
    W0 = qfa.ensemble.make_W0(N, 5)
    dt = 0.1 * qfa.hbar(N)
    sim = Simulation(str(tmp_path / "run.qf"), overwrite=True, state=W0, qutypes={'mat': None, 'shr': None},
                     loggers={'enstrophy': qfa.enstrophy})
    solve(W0.copy(), stepsize=0.1, steps=50, steps_out=10, callback=sim)
    sim2 = Simulation(str(tmp_path / "run.qf"))
    Wend = solve(sim2, stepsize=0.1, steps=50, steps_out=10)
    sim3 = Simulation(str(tmp_path / "straight.qf"), overwrite=True, state=W0)
    solve(W0.copy(), stepsize=0.1, steps=100, steps_out=10, callback=sim3)
    np.testing.assert_array_equal(sim['mat'], sim3['mat'])
    np.testing.assert_array_equal(Wend, sim3['mat', -1])
    np.testing.assert_array_equal(sim['step'], 10 * np.arange(11))
    np.testing.assert_allclose(sim['time'], 10 * dt * np.arange(11))
    # the same chunks through the host-array stepper
    W = W0.copy()
    for chunk in range(10):
        stats = {"iterations": 0.0}
        W = qfa.isomp(W, dt, steps=10, stats=stats)
        np.testing.assert_array_equal(sim['mat', chunk + 1], W)
        assert sim['iterations', chunk + 1] == stats["iterations"]
        assert sim['tol_auto', chunk + 1] == stats["tol_auto"]
    np.testing.assert_allclose(sim['shr', -1], qfa.mat2shr(sim['mat', -1]), rtol=0, atol=1e-13)
    assert sim['enstrophy', -1] == qfa.enstrophy(sim['mat', -1])


def test_enstrophy_drift_is_the_oracles(qfa, oracle):
    """Long runs show an enstrophy drift that is LINEAR in the step count (N=2048, 10,000 steps: -1.4e-15
    per step, profiles/r0x_longrun_*): a secular bias of the fixed-point tolerance (the midpoint equation
    is solved to sqrt(eps) only), not a random walk of rounding errors.  It is the method's, not the
    device's: the CPU oracle drifts with the same slope on the same input."""
    N, chunk, chunks = 256, 100, 4
    W0 = oracle.make_W0(N, 0)
    dt = 0.25 * qfa.hbar(N)
    Wg, Wc = W0.copy(), W0.copy()
    s0 = oracle.enstrophy(W0)
    dg, dc = [], []
    for c in range(chunks):
        Wg = qfa.isomp(Wg, dt, steps=chunk)
        Wc = oracle.isomp(Wc, dt, steps=chunk)
        dg.append(oracle.enstrophy(Wg) - s0)
        dc.append(oracle.enstrophy(Wc) - s0)
    steps = chunk * np.arange(1, chunks + 1)
    slope_g = float(np.polyfit(steps, dg, 1)[0])
    slope_c = float(np.polyfit(steps, dc, 1)[0])
    print("enstrophy drift per step: device %.3e, oracle %.3e" % (slope_g, slope_c))
    assert abs(slope_g - slope_c) <= 0.25 * abs(slope_c) + 1e-17
    assert maxabs(dg, dc) <= 0.25 * np.abs(dc).max() + 1e-15


# ----------------------------------------------------------------------------- protocol behaviour
def test_stepper_contract(qfa):
    W0 = qfa.ensemble.make_W0(16, 1)
    W = W0.copy()
    out = qfa.isomp(W, 0.01, steps=2)
    assert out is W                                       # in place AND returned
    with pytest.raises(AssertionError):
        qfa.isomp(W0.copy(), 0.1, steps=1, minit=0)
    with pytest.raises(AssertionError):
        qfa.isomp(W0.copy(), 0.1, steps=1, minit=5, maxit=4)
    stats = {}
    qfa.isomp(W0.copy(), 0.01, steps=2, stats=stats)
    assert stats == {}                                    # empty dict is falsy (isospectral.py:451,609)
    # the host hooks are accepted, also on stacked states (test_isomp_hooks_*_golden), with one stream matrix for all
    # states or one per state (test_isomp_hamiltonian_per_state_golden); any other shape is refused
    Wstack = np.stack([W0, W0])
    with pytest.raises(ValueError):           # numpy's broadcasting error in the reference (isospectral.py:496-509)
        qfa.isomp(Wstack.copy(), 0.01, steps=1, hamiltonian=lambda W: W[:, :4, :4])
    with pytest.raises(ValueError):
        qfa.isomp(W0.copy(), 0.01, steps=1, hamiltonian=lambda W: W[:4, :4])
    # what `range(steps)` and the in-place complex updates of the reference do with odd arguments
    with pytest.raises(TypeError):
        qfa.isomp(W0.copy(), 0.01, steps=2.0)
    with pytest.raises(TypeError):
        qfa.isomp(np.zeros((16, 16)), 0.01, steps=1)               # a real W: numpy refuses the complex in-place update
    np.testing.assert_array_equal(qfa.isomp(W0.copy(), 0.01, steps=-3), W0)
    # hooks that do nothing change nothing
    Wplain = qfa.isomp(Wstack.copy(), 0.01, steps=2)
    for kw in ({"callback": lambda W, dW: None}, {"strang_splitting": lambda h, W: W},
               {"forcing": lambda P, W: np.zeros_like(W)}):
        assert maxabs(qfa.isomp(Wstack.copy(), 0.01, steps=2, **kw), Wplain) <= 1e-15
    # `time` is accepted and ignored for the autonomous built-in Hamiltonian
    Wa = qfa.isomp(W0.copy(), 0.01, steps=3, time=2.0)
    Wb = qfa.isomp(W0.copy(), 0.01, steps=3)
    np.testing.assert_array_equal(Wa, Wb)
    # device-object form (quflow/simulation.py:554-562)
    stepper = qfa.IsompHIP(16, np.complex128)
    ham = qfa.PoissonHIP(16, np.complex128)
    Wc = stepper(W0.copy(), 0.01, steps=3, hamiltonian=ham)
    np.testing.assert_array_equal(Wc, Wb)


def test_driver_loop_restatement(qfa):
    """simulation.solve's chunk loop (quflow/simulation.py:726-798) restated locally: the
    stepper is discovered through inspect.getfullargspec and driven with time/hamiltonian/
    stats kwargs; chunked-and-restarted runs are bit-identical (tests/test_simulation.py:130-168)."""
    import inspect
    N = 32
    W0 = qfa.ensemble.make_W0(N, 9)
    dt = 0.25 * qfa.hbar(N)

    def solve(W, steps, steps_out, integrator, time=0.0):
        kwargs = {"time": time, "hamiltonian": qfa.solve_poisson}
        if 'stats' in inspect.getfullargspec(integrator).args:
            kwargs['stats'] = {'iterations': 0.0}
        log = []
        for k in range(0, steps, steps_out):
            n = min(steps_out, steps - k)
            W = integrator(W, dt, steps=n, **kwargs)
            kwargs['time'] += n * dt
            log.append((kwargs['time'], qfa.energy_euler(W), dict(kwargs.get('stats', {}))))
        return W, log

    assert 'stats' in inspect.getfullargspec(qfa.isomp).args
    Wfull, log = solve(W0.copy(), 40, 10, qfa.isomp)
    Whalf, _ = solve(W0.copy(), 20, 10, qfa.isomp)
    Wrest, _ = solve(Whalf.copy(), 20, 10, qfa.isomp, time=20 * dt)
    np.testing.assert_array_equal(Wfull, Wrest)
    assert all("iterations" in entry[2] for entry in log)
    Wobj, _ = solve(W0.copy(), 40, 10, qfa.IsompHIP(N, np.complex128))
    np.testing.assert_array_equal(Wfull, Wobj)


def test_device_trajectory_and_ensemble_single_rank(qfa):
    N = 64
    dt = 0.25 * qfa.hbar(N)
    hist, trajs = qfa.ensemble.run_ensemble(N, seeds=[0, 1], dt=dt, steps=20, steps_out=10)
    assert len(hist) == 2 and hist[0].shape == (2, 4)
    for seed, tr in trajs:
        W = qfa.ensemble.make_W0(N, seed)
        W = qfa.isomp(W, dt, steps=10)
        W = qfa.isomp(W, dt, steps=10)
        np.testing.assert_array_equal(tr.download(), W)
        row = hist[-1][int(seed)]
        assert row[0] == seed
        np.testing.assert_allclose(row[1], qfa.energy_euler(W), rtol=1e-13)
        np.testing.assert_allclose(row[2], 0.5, rtol=1e-11)


def test_profile_counters(qfa):
    import ctypes
    from quflow_amd import _lib
    N = 128
    tr = qfa.DeviceTrajectory(qfa.ensemble.make_W0(N, 0))
    lib, h = tr.ctx._lib, tr.ctx.handle
    _lib.check(lib.qf_profile_reset(h))
    _lib.check(lib.qf_profile_enable(h, 0x1F))
    st = tr.advance(0.25 * qfa.hbar(N), 5, minit=3, maxit=3)
    _lib.check(lib.qf_profile_enable(h, 0))
    n = ctypes.c_longlong()
    ms = ctypes.c_double()
    for name in ("poisson", "gemm1", "gemm2"):
        _lib.check(lib.qf_profile_read(h, _lib.KERNEL_IDS[name], ctypes.byref(n), ctypes.byref(ms)))
        assert n.value == 15 == st["total_iterations"] and ms.value > 0.0
    # fused step end (default): no decide / update launches at all; the legacy protocol
    # (compsum, reinitialize) has one update per step
    _lib.check(lib.qf_profile_read(h, _lib.KERNEL_IDS["update"], ctypes.byref(n), ctypes.byref(ms)))
    assert n.value == 0
    _lib.check(lib.qf_profile_read(h, _lib.KERNEL_IDS["norm"], ctypes.byref(n), ctypes.byref(ms)))
    assert n.value == 0
    _lib.check(lib.qf_profile_reset(h))
    _lib.check(lib.qf_profile_enable(h, 0x1F))
    st = tr.advance(0.25 * qfa.hbar(N), 5, minit=3, maxit=3, reinitialize=True)
    _lib.check(lib.qf_profile_enable(h, 0))
    _lib.check(lib.qf_profile_read(h, _lib.KERNEL_IDS["update"], ctypes.byref(n), ctypes.byref(ms)))
    assert n.value == 5
    _lib.check(lib.qf_profile_read(h, _lib.KERNEL_IDS["norm"], ctypes.byref(n), ctypes.byref(ms)))
    assert n.value == 15 == st["total_iterations"]


# ----------------------------------------------------------------------------- quantization
@pytest.fixture(scope="module")
def qoracle():
    from oracle import quantization_oracle
    return quantization_oracle


@pytest.mark.parametrize("N", [5, 16, 33])
def test_basis_device_vs_reference(qfa, N):
    """compute_basis on the device (k_basis: twisted factorisations at the known spectrum) against
    the basis the reference computed with LAPACK (quantization.py:68-113)."""
    from quflow_amd import quantization as q
    g = load_golden("quantization")
    b = q.compute_basis(N)
    assert b.shape == g["basis_N%d" % N].shape
    assert maxabs(b, g["basis_N%d" % N]) <= 1e-12


@pytest.mark.parametrize("N", [2, 65, 129, 513])
def test_basis_device_vs_oracle(qfa, qoracle, N):
    """Larger N against the oracle's LAPACK basis; every block orthogonal with norm sqrt(N)."""
    from quflow_amd import quantization as q
    b = q.compute_basis(N)
    ref = qoracle.compute_basis(N)
    assert maxabs(b, ref) <= 1e-12 * N
    for m in (0, 1, N // 2, N - 2):
        n = N - m
        if n < 1:
            continue
        o = q.basis_break_index(m, N)
        B = b[o:o + n * n].reshape(n, n)
        assert maxabs(B.T @ B, N * np.eye(n)) <= 1e-11 * N * N


@pytest.mark.parametrize("N", [5, 16, 33, 64])
def test_sh_transforms_golden(qfa, N):
    """shr2mat / mat2shr / shc2mat / mat2shc on the device (quantization.hip) against the
    reference's own output (quflow/quantization.py:188-396, 450-592)."""
    from quflow_amd import quantization as q
    g = load_golden("quantization")
    pre = "N%d_" % N
    tol = 1e-13 * N
    assert maxabs(q.shr2mat(g[pre + "omega"], N=N), g[pre + "shr2mat"]) <= tol
    assert maxabs(q.mat2shr(g[pre + "W"]), g[pre + "mat2shr"]) <= tol
    assert maxabs(q.mat2shr(g[pre + "G"]), g[pre + "mat2shr_G"]) <= tol
    assert maxabs(q.shc2mat(g[pre + "omega_c"], N=N), g[pre + "shc2mat"]) <= tol
    assert maxabs(q.mat2shc(g[pre + "G"]), g[pre + "mat2shc_G"]) <= tol
    if pre + "berezin_w" in g.files:        # the Berezin-Toeplitz scaling option (utils.py:108-135)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert maxabs(q.berezin_multipliers(N), g[pre + "berezin_w"]) <= 1e-13
            assert maxabs(q.shr2mat(g[pre + "omega"], N=N, berezin=True), g[pre + "shr2mat_berezin"]) <= 1e-12 * np.abs(g[pre + "shr2mat_berezin"]).max()
            assert maxabs(q.mat2shr(g[pre + "W"], berezin=True), g[pre + "mat2shr_berezin"]) <= tol
            assert maxabs(q.shc2mat(g[pre + "omega_c"], N=N, berezin=True), g[pre + "shc2mat_berezin"]) <= 1e-12 * np.abs(g[pre + "shc2mat_berezin"]).max()
            assert maxabs(q.mat2shc(g[pre + "G"], berezin=True), g[pre + "mat2shc_berezin"]) <= tol


@pytest.mark.parametrize("N", [33, 64])
def test_sh_short_omega_golden(qfa, N):
    """band-limited coefficient arrays (tests/test_quantization.py:54-96) and mat2shr's elmax."""
    from quflow_amd import quantization as q
    from quflow_amd import _lib
    from quflow_amd.context import ptr
    import ctypes
    g = load_golden("quantization")
    Ws = q.shr2mat(g["short_omega"], N=N)
    assert maxabs(Ws, g["short_N%d_shr2mat" % N]) <= 1e-13 * N
    # zero outside the band: diagonals |m| >= 3 untouched by 10 coefficients (el <= 2)
    assert np.count_nonzero(np.triu(Ws, 3)) == 0
    om = np.zeros(10)
    ctx = q._resident_context(N)
    Wc = np.ascontiguousarray(g["short_N%d_shr2mat" % N])
    _lib.check(ctx._lib.qf_mat2shr(ctx.handle, ptr(Wc), ptr(om), ctypes.c_longlong(10)))
    assert maxabs(om, g["short_N%d_mat2shr10" % N]) <= 1e-13 * N
    assert maxabs(q.mat2shr(Wc, elmax=2), g["short_N%d_mat2shr_elmax2" % N]) <= 1e-13 * N
    # round trip (test_mat2shr_short_omega): 10 entries carry el < int(sqrt(10)) = 3, i.e. 9 coefficients
    np.testing.assert_allclose(om[:9], g["short_omega"][:9], atol=1e-12)
    assert om[9] == 0.0


@pytest.mark.parametrize("N", [129, 512])
def test_sh_transforms_vs_oracle_large(qfa, qoracle, N):
    """Larger N against the CPU oracle on the same basis, plus the round trips
    mat2shr(shr2mat(omega)) == omega and the state-resident forms (NULL matrix pointer)."""
    from quflow_amd import quantization as q
    from quflow_amd import _lib
    from quflow_amd.context import ptr
    import ctypes
    rng = np.random.default_rng(N)
    basis = q.get_basis(N)
    qoracle._cache[N] = basis          # same basis on both sides: the transforms are what is compared
    omega = rng.standard_normal(N * N)
    W = q.shr2mat(omega, N=N)
    Wc = qoracle.shr2mat(omega, N=N)
    scale = np.abs(Wc).max()
    assert maxabs(W, Wc) <= 1e-13 * N * scale
    assert np.array_equal(W, -W.conj().T)                     # skew-Hermitian by construction
    om2 = q.mat2shr(W)
    assert maxabs(om2, omega) <= 1e-11 * np.abs(omega).max()
    G = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    assert maxabs(q.mat2shr(G), qoracle.mat2shr(G)) <= 1e-13 * N * np.abs(G).max()
    omc = q.mat2shc(G)
    assert maxabs(omc, qoracle.mat2shc(G)) <= 1e-13 * N * np.abs(G).max()
    assert maxabs(q.shc2mat(omc, N=N), G) <= 1e-11 * np.abs(G).max()      # round trip on gl(N, C)
    # state-resident: coefficients -> ctx state -> coefficients, W never crosses PCIe
    ctx = q._resident_context(N)
    _lib.check(ctx._lib.qf_shr2mat(ctx.handle, ptr(omega), ctypes.c_longlong(N * N), None))
    om3 = np.zeros(N * N)
    _lib.check(ctx._lib.qf_mat2shr(ctx.handle, None, ptr(om3), ctypes.c_longlong(N * N)))
    np.testing.assert_array_equal(om3, om2)


def test_trajectory_from_and_to_shr(qfa):
    """Initial data and 'shr' output of a resident trajectory without moving W over PCIe
    (shr2mat -> isomp steps -> mat2shr), against the host-in/host-out forms."""
    from quflow_amd import quantization as q
    N = 128
    rng = np.random.default_rng(3)
    omega = np.zeros(N * N)
    omega[1:200] = rng.standard_normal(199)          # smooth, trace-free (omega[0] = 0)
    tr = qfa.DeviceTrajectory.from_shr(omega, N=N)
    W0 = q.shr2mat(omega, N=N)
    np.testing.assert_array_equal(tr.download(), W0)
    dt = 0.25 * qfa.hbar(N)
    tr.advance(dt, 7)
    W = qfa.isomp(W0.copy(), dt, steps=7)
    np.testing.assert_array_equal(tr.download(), W)
    np.testing.assert_array_equal(tr.shr(), q.mat2shr(W))
    assert maxabs(tr.shr(100), q.mat2shr(W)[:100]) == 0.0


def test_sh_requires_basis(qfa):
    from quflow_amd import _lib
    from quflow_amd.context import Context, ptr
    import ctypes
    ctx = Context(8)
    try:
        om = np.zeros(64)
        W = np.zeros((8, 8), dtype=complex)
        with pytest.raises(_lib.QuflowHipError):
            _lib.check(ctx._lib.qf_shr2mat(ctx.handle, ptr(om), ctypes.c_longlong(64), ptr(W)))
        with pytest.raises(_lib.QuflowHipError):
            _lib.check(ctx._lib.qf_basis_upload(ctx.handle, ptr(om), ctypes.c_longlong(64)))
    finally:
        ctx.close()


# ---------------------------------------------------------------------------------------------
# complex64 input: the reference's dtype contract (quflow/laplacian/cpu.py:721-734, isospectral.py:441)
# ---------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_single_precision_input_contract(qfa):
    """complex64 in -> complex64 out, in place; the automatic tolerance is the reference's single-precision
    one (reported in stats['tol_auto']) so that the run stops where the reference's does.  The device
    evaluates in double precision: its result sits within single-precision rounding of the reference's."""
    g = load_golden("single_precision")
    N = int(g["N"])
    W0 = g["W0"]
    assert W0.dtype == np.complex64
    P = qfa.solve_poisson(W0)
    assert P.dtype == np.complex64 and str(g["P_dtype"]) == "complex64"
    scale = np.abs(g["P"]).max()
    assert np.abs(P - g["P"]).max() <= 64 * np.finfo(np.float32).eps * scale * np.sqrt(N)
    L = qfa.laplace(g["P"])
    assert L.dtype == np.complex64
    assert np.abs(L - g["laplace_P"]).max() <= 64 * np.finfo(np.float32).eps * np.abs(g["laplace_P"]).max() * np.sqrt(N)
    dt = 0.25 * qfa.hbar(N)
    for tag, kw in (("plain", {}), ("compsum", {"compsum": True}), ("tol1e-3", {"tol": 1e-3})):
        W = W0.copy()
        stats = {"iterations": 0.0}
        out = qfa.isomp(W, dt, steps=8, stats=stats, **kw)
        assert out is W and W.dtype == np.complex64
        ref = g[tag + "_W"]
        assert np.abs(W - ref).max() <= 2e-5 * np.abs(ref).max(), tag
        if tag + "_tol" in g.files:
            assert abs(stats["tol_auto"] - float(g[tag + "_tol"])) <= 1e-6 * float(g[tag + "_tol"]), tag
        else:
            assert "tol_auto" not in stats
        # the plain exit threshold sits far above single-precision noise: same count; the compensated
        # one (eps_32 itself) is reached within the reference's noise: a step may differ by an iteration
        slack = 0.0 if tag != "compsum" else 0.5
        assert abs(stats["iterations"] - float(g[tag + "_iterations"])) <= slack, (tag, stats["iterations"])
        assert stats["number_of_maxit"] == float(g[tag + "_maxit"])
    # the double-precision automatic tolerance is untouched by all this
    W = W0.astype(np.complex128)
    stats = {"iterations": 0.0}
    qfa.isomp(W, dt, steps=2, stats=stats)
    assert stats["tol_auto"] < 1e-3 * float(g["plain_tol"])


@pytest.mark.parametrize("N,kw", [(64, {}), (256, {}), (64, {"compsum": True}), (100, {})])
def test_advance_with_diagnostics_equals_two_calls(qfa, N, kw):
    """qf_isomp_diag = qf_isomp + qf_diagnostics under one synchronisation: same state, same statistics,
    bit-identical energy and enstrophy -- on the fused exit (N % 64 == 0), on the two-kernel step end
    (compsum) and on the full-product path (N = 100)."""
    W0 = qfa.ensemble.make_W0(N, 5)
    dt = 0.25 * qfa.hbar(N)
    a = qfa.DeviceTrajectory(W0)
    b = qfa.DeviceTrajectory(W0)
    for chunk in (3, 1, 4):
        sa = a.advance(dt, chunk, diagnostics=True, **kw)
        sb = b.advance(dt, chunk, **kw)
        eb, nb = b.diagnostics()
        assert sa["energy"] == eb and sa["enstrophy"] == nb
        assert sa["total_iterations"] == sb["total_iterations"] and sa["tol"] == sb["tol"]
        np.testing.assert_array_equal(a.download(), b.download())
    a.ctx.close()
    b.ctx.close()


@pytest.mark.parametrize("products", ["f64", "i8x65"])
def test_config5_long_run_in_the_suite(qfa, products, monkeypatch):
    """BASELINE config 5 (N = 2048, long run) at a length the GPU suite can afford: 2,000 steps in ten chunks
    on a resident trajectory (the 10,000-step record is profiles/r04_longrun_n2048_10k_steps.json; tools/longrun.py).  The state
    stays exactly skew-Hermitian, the spectrum and the Casimirs are conserved to rounding accumulated over the
    run, the enstrophy drift stays on its (linear, rounding-bias) line, every chunk closes its steps in two or
    three iterations."""
    N = 2048
    W0 = qfa.ensemble.make_W0(N, 0)
    dt = 0.25 * qfa.hbar(N)
    if products != "f64":       # config 3's products (what QUFLOW_HIP_GEMM=auto selects) under the fp64 run's own bounds
        monkeypatch.setenv("QUFLOW_HIP_GEMM", products)
    tr = qfa.DeviceTrajectory(W0)
    e0, s0 = tr.diagnostics()
    its = []
    for _ in range(10):
        st = tr.advance(dt, 200, diagnostics=True)
        its.append(st["iterations"])
        assert st["number_of_maxit"] == 0.0
    W = tr.download()
    tr.ctx.close()
    assert all(2.0 <= x <= 2.1 for x in its), its
    assert maxabs(W, -W.conj().T) == 0.0
    assert abs(np.trace(W)) <= 1e-13          # (round 5: the int8 products too -- fp64 diagonal of the first product; measured 1.6e-15 after 10,000 steps)
    assert abs(st["enstrophy"] - s0) <= 2000 * 5e-15        # (observed: -1.4e-15 per step)
    assert abs(st["energy"] - e0) <= 1e-9 * abs(e0) + 1e-13
    ev0 = np.linalg.eigvalsh(1j * W0)
    ev = np.linalg.eigvalsh(1j * W)
    assert np.abs(ev - ev0).max() <= 2e-9
    for k in (2, 3, 4):
        assert abs((ev ** k).sum() - (ev0 ** k).sum()) / N <= 1e-9


def test_config4_workload_on_one_rank(qfa):
    """BASELINE config 4's workload (N = 1024, eight independent initial conditions, energy / enstrophy gathered
    per chunk) with the eight replicas owned by ONE rank -- what a rank does when replicas outnumber GPUs
    (DeviceEnsemble, qf_isomp_multi).  The eight-GPU form of it is the driver's to run; here every replica's
    gathered row must be the row its own single-trajectory run produces, bit for bit."""
    N, steps = 1024, 12
    dt = 0.25 * qfa.hbar(N)
    history, _ = qfa.ensemble.run_ensemble(N, list(range(8)), dt, steps, steps_out=6)
    assert len(history) == 2 and all(h.shape == (8, 4) for h in history)
    np.testing.assert_array_equal(history[-1][:, 0], np.arange(8.0))
    for seed in (0, 5):
        tr = qfa.DeviceTrajectory(qfa.ensemble.make_W0(N, seed))
        rows = []
        for _ in range(2):
            st = tr.advance(dt, 6, diagnostics=True)
            rows.append([float(seed), st["energy"], st["enstrophy"], st["iterations"]])
        tr.ctx.close()
        for c in range(2):
            np.testing.assert_array_equal(history[c][seed], np.array(rows[c]))


# ----------------------------------------------------------------------------- round 3: interface gaps
def _foreign_hamiltonian(qfa):
    def h(W):
        W0 = W[(0,) * (W.ndim - 2) + (Ellipsis,)] if W.ndim > 2 else W
        return 0.5 * qfa.solve_poisson(W0) + 0.1j * np.eye(W.shape[-1])
    return h


def test_commutators_and_estimate_stepsize_golden(qfa):
    """quflow.integrators.commutator / commutator_generic / commutator_skewherm (isospectral.py:22-57) and
    estimate_stepsize (:121-148) against values the reference computed; select_skewherm switches the default."""
    g = load_golden("interfaces")
    W, P, G = g["W"], g["P"], g["G"]
    scale = (np.abs(W) @ np.abs(P)).max()
    assert maxabs(qfa.commutator_skewherm(W, P), g["comm_skew"]) <= 16 * EPS * 33 * scale
    assert maxabs(qfa.commutator(W, P), g["comm_default"]) <= 16 * EPS * 33 * scale
    assert maxabs(qfa.commutator_generic(W, G), g["comm_generic"]) <= 32 * EPS * 33 * (np.abs(W) @ np.abs(G)).max()
    C = qfa.commutator_skewherm(W, P)
    assert np.array_equal(C, -C.conj().T)
    np.testing.assert_allclose(qfa.estimate_stepsize(W), float(g["stepsize_default"]), rtol=1e-12)
    np.testing.assert_allclose(qfa.estimate_stepsize(W, P=2.0 * P, safety_factor=0.25), float(g["stepsize_P"]), rtol=1e-12)
    assert qfa.integrators.commutator is qfa.integrators.commutator_skewherm
    qfa.integrators.select_skewherm(False)
    try:
        assert qfa.integrators.commutator is qfa.integrators.commutator_generic
    finally:
        qfa.integrators.select_skewherm(True)
    assert qfa.integrators.commutator is qfa.integrators.commutator_skewherm


@pytest.mark.parametrize("n", [16, 32])
def test_lu_steppers_foreign_hamiltonian_golden(qfa, n):
    """isomp_simple / isomp_quasinewton with a foreign Hamiltonian (isospectral.py:207, 286): called back once per
    pass while the Newton-Schulz inverse, the two solves and the update stay on the device; `forcing` is accepted
    and ignored, as in the reference (:185-186, 283-284)."""
    g = load_golden("interfaces")
    pre = "lu_N%d_" % n
    W0, dt, steps = g[pre + "W0"], float(g[pre + "dt"]), int(g[pre + "steps"])
    h = _foreign_hamiltonian(qfa)
    W = W0.copy()
    out = qfa.isomp_simple(W, dt, steps, hamiltonian=h)
    assert out is W
    assert maxabs(W, g[pre + "simple_foreign"]) <= 1e-11
    st = {}
    W = qfa.isomp_quasinewton(W0.copy(), dt, steps, hamiltonian=h, stats=st)
    assert maxabs(W, g[pre + "qn_foreign"]) <= 1e-11
    assert st["iterations"] >= 1.0
    with pytest.warns(UserWarning, match="ignore `forcing`"):
        W = qfa.isomp_simple(W0.copy(), dt, steps, forcing=lambda P_, W_: W_)
    assert maxabs(W, g[pre + "simple_forcing"]) <= 1e-11

    def broken(W_):
        raise KeyError("from the hook")
    with pytest.raises(KeyError):
        qfa.isomp_simple(W0.copy(), dt, 2, hamiltonian=broken)


@pytest.mark.parametrize("n", [16, 33])
def test_erk_on_stacks_golden(qfa, n):
    """euler / heun / rk4 on a (k,N,N) stack (erk.py with batched input): P from state 0, the bracket broadcast
    over the stack -- in place, against the reference's vectors."""
    g = load_golden("interfaces")
    pre = "erk_N%d_" % n
    S0, dt, steps = g[pre + "S0"], float(g[pre + "dt"]), int(g[pre + "steps"])
    for name in ("euler", "heun", "rk4"):
        S = S0.copy()
        out = getattr(qfa, name)(S, dt, steps)
        assert out is S
        assert maxabs(S, g[pre + name]) <= 1e-12, name
    # state 0 of the stack evolves as it does alone
    W = S0[0].copy()
    qfa.rk4(W, dt, steps)
    assert maxabs(W, g[pre + "rk4"][0]) <= 1e-12


@pytest.mark.parametrize("n", [16, 32])
def test_magmp_hooks_golden(qfa, n):
    """magmp with host hooks (quflow/integrators/mhd.py:235-456): forcing(P, state), a foreign Hamiltonian returning the
    pair (P, B), callback(state, 2 PWcomm), time dependence -- qf_isomp_hooked in its magnetic mode keeps the (2,N,N)
    state, the products and the magnetic terms on the device.  Against the reference's own runs."""
    g = load_golden("interfaces")
    pre = "mhd_N%d_" % n
    s0, dt, steps = g[pre + "state0"], float(g[pre + "dt"]), int(g[pre + "steps"])

    def forcing(P, st):
        return -0.05 * st

    def foreign(st):
        return 0.8 * qfa.solve_poisson(st[0]), 0.9 * qfa.laplace(st[1])

    def forcing_t(P, st, time=0.0):
        return (-0.05 * np.cos(time)) * st

    def foreign_t(st, time=0.0):
        return (0.8 + 0.1 * np.sin(time)) * qfa.solve_poisson(st[0]), 0.9 * qfa.laplace(st[1])
    st = {"iterations": 0.0}
    S = s0.copy()
    out = qfa.magmp(S, dt, steps, forcing=forcing, stats=st)
    assert out is S
    assert maxabs(S, g[pre + "forcing"]) <= STEP_TOL
    assert st["iterations"] == float(g[pre + "forcing_iterations"]) and "tol" in st and "maxit" in st
    st = {"iterations": 0.0}
    assert maxabs(qfa.magmp(s0.copy(), dt, steps, hamiltonian=foreign, stats=st), g[pre + "foreign"]) <= STEP_TOL
    assert st["iterations"] == float(g[pre + "foreign_iterations"])
    seen = []
    W = qfa.magmp(s0.copy(), dt, steps, callback=lambda W_, d_: seen.append((np.linalg.norm(W_), np.linalg.norm(d_))))
    assert maxabs(W, g[pre + "callback"]) <= STEP_TOL
    np.testing.assert_allclose(np.array(seen), g[pre + "callback_seen"], rtol=1e-9)
    st = {"iterations": 0.0}
    W = qfa.magmp(s0.copy(), dt, steps, time=0.5, forcing=forcing_t, hamiltonian=foreign_t, stats=st)
    assert maxabs(W, g[pre + "timed"]) <= STEP_TOL
    assert st["iterations"] == float(g[pre + "timed_iterations"])
    # hooks that change nothing change nothing: the built-in solve_mhd through the hooked loop = the plain magmp
    plain = qfa.magmp(s0.copy(), dt, steps)
    hooked = qfa.magmp(s0.copy(), dt, steps, callback=lambda W_, d_: None)
    assert maxabs(plain, hooked) <= 1e-13


@pytest.mark.parametrize("n", [16, 32])
def test_lu_steppers_general_branch_golden(qfa, n):
    """isomp_simple / isomp_quasinewton with select_skewherm(False) on a general matrix (isospectral.py:303-314): two
    Newton-Schulz inverses (of I - E and of I + E) per step for the general branch of isomp_simple; isomp_quasinewton
    keeps its formulas with A^H = I - E^H formed explicitly.  Against the reference's runs."""
    g = load_golden("interfaces")
    pre = "lug_N%d_" % n
    W0, dt, steps = g[pre + "W0"], float(g[pre + "dt"]), int(g[pre + "steps"])
    qfa.integrators.select_skewherm(False)
    try:
        Ws = qfa.isomp_simple(W0.copy(), dt, steps)
        Wq = qfa.isomp_quasinewton(W0.copy(), dt, steps)
    finally:
        qfa.integrators.select_skewherm(True)
    assert maxabs(Ws, g[pre + "simple"]) <= 1e-11
    assert maxabs(Wq, g[pre + "qn"]) <= 1e-11


@pytest.mark.parametrize("N,steps,kw", [(64, 40, {}), (256, 30, {}), (512, 12, {}), (512, 6, {"minit": 3, "maxit": 3}),
                                        (96, 20, {"maxit": 1}), (512, 8, {"tol": 1e-30, "maxit": 4}),
                                        (100, 20, {}), (333, 10, {"tol": 1e-30, "maxit": 3}), (500, 8, {})])
def test_deferred_step_end_is_bit_identical(qfa, N, steps, kw, monkeypatch):
    """Deferred step end (N <= 512: the exit decision of an iteration is taken by the next solve's workgroups from
    the row sums) against the decision inside the second product's last finisher: the same sums in the
    same order, so the same bits, iteration counts and statistics -- adaptive, fixed-iteration, maxit-exhausting and
    chunked runs (a chunk's last decision is taken by the one-workgroup k_decide launch)."""
    from quflow_amd.context import release_contexts
    W0 = qfa.ensemble.make_W0(N, 6)
    dt = 0.25 * qfa.hbar(N)
    res = {}
    for defer in ("1", "0"):
        monkeypatch.setenv("QUFLOW_HIP_DEFER", defer)
        release_contexts()
        st = {"iterations": 0.0}
        W = qfa.isomp(W0.copy(), dt, steps=steps, stats=st, **kw)
        tr = qfa.DeviceTrajectory(W0)
        sts = [tr.advance(dt, n, **kw) for n in (1, steps - 1)]
        Wc = tr.download()
        tr.ctx.close()
        res[defer] = (W, dict(st), Wc, [(s["total_iterations"], s["number_of_maxit"], s["last_resnorm"], s["tol"]) for s in sts])
    release_contexts()
    np.testing.assert_array_equal(res["1"][0], res["0"][0])
    assert res["1"][1] == res["0"][1]
    np.testing.assert_array_equal(res["1"][2], res["0"][2])
    assert res["1"][3] == res["0"][3]
