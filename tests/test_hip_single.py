"""GPU parity tests of the complex64 path: the device computes complex64 data in float32, as the reference
does (float32 tables and Thomas solve, quflow/laplacian/cpu.py:725; complex64 products,
quflow/integrators/isospectral.py:496,499; float32 automatic tolerance, :440-448).

Checked against (1) vectors the reference produced on complex64 input (tests/golden/single_precision.npz,
oracle/gen_golden.py) and (2) the oracle's float32 restatement, which reproduces those vectors bit for bit
(tests/test_oracle_vs_golden.py::test_single_precision_oracle_reproduces_the_reference).

Tolerances (float32, eps = 1.19e-7):
  * coefficient table, laplace: bit-exact (same operations, one rounding each);
  * Poisson solve: a few eps of the data scale at the fixture sizes; at large N both float32 solves carry the
    conditioning of the tridiagonal systems (kappa ~ N^2/2), so device and oracle are each held to the
    double-precision solution within `SOLVE_GROWTH(N) * eps * max|P|` and to each other within twice that;
  * products: 8 eps N-independent factor times max(|A||B|) (k-ordered fma chains in float32);
  * stepper: <= 1e-5 relative to the state's scale, identical iteration counts, tol_auto to 1e-6.
"""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

EPS32 = float(np.finfo(np.float32).eps)


@pytest.fixture(scope="module")
def qfa():
    import quflow_amd
    if quflow_amd.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the MI355X box")
    return quflow_amd


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.complex128) - np.asarray(b, dtype=np.complex128))))


def make_W0_c64(oracle, N, seed):
    return oracle.make_W0(N, seed).astype(np.complex64)


@pytest.mark.parametrize("N", [64, 101])
def test_float32_table_is_the_references(qfa, N):
    g = load_golden("single_precision")
    lap = qfa.laplacian.laplacian(N, bc=True, dtype=np.float32)
    assert lap.dtype == np.float32
    np.testing.assert_array_equal(lap, g["N%d_lap_bc" % N])


@pytest.mark.parametrize("N", [128, 500, 1024])
def test_float32_table_vs_oracle(qfa, oracle, N):
    for bc in (True, False):
        np.testing.assert_array_equal(qfa.laplacian.laplacian(N, bc=bc, dtype=np.float32), oracle.laplacian(N, bc, np.float32))


def test_laplace_c64_bit_exact(qfa, oracle):
    g = load_golden("single_precision")
    for key_p, key_w in (("P", "laplace_P"), ("N64_P", "N64_laplace_P"), ("N101_P", "N101_laplace_P")):
        L = qfa.laplace(g[key_p])
        assert L.dtype == np.complex64
        np.testing.assert_array_equal(L, g[key_w])
    P = make_W0_c64(oracle, 512, 3)
    np.testing.assert_array_equal(qfa.laplace(P), oracle.laplace(P))


def test_solve_poisson_c64_reference_vectors(qfa):
    """float32 solve against the reference's own complex64 results: a few float32 ulp of the data scale."""
    g = load_golden("single_precision")
    for kw, kp in (("W0", "P"), ("N64_W0", "N64_P"), ("N101_W0", "N101_P")):
        P = qfa.solve_poisson(g[kw])
        assert P.dtype == np.complex64
        ref = g[kp]
        err = maxabs(P, ref)
        assert err <= 16 * EPS32 * np.abs(ref).max(), (kw, err / (EPS32 * np.abs(ref).max()))
        # exactly skew-Hermitian and trace-free to rounding, as the reference's
        assert np.array_equal(P, -P.conj().T)
        assert abs(np.trace(P)) <= 64 * EPS32 * np.abs(ref).max()
    # the same buffer comes back on every call (cpu.py:726,734)
    assert qfa.solve_poisson(g["W0"]) is qfa.solve_poisson(g["W0"])


@pytest.mark.parametrize("N", [128, 256, 512, 1000, 1024, 2048])
def test_solve_poisson_c64_vs_oracle_large(qfa, oracle, N):
    """Device (chunked scan order) and oracle (the reference's sequential order) are two float32 evaluations of
    an ill-conditioned solve: each is held to the double-precision solution, and they to each other."""
    W = make_W0_c64(oracle, N, 11)
    P_dev = qfa.solve_poisson(W).copy()
    P_ora = oracle.solve_poisson(W).copy()
    P_f64 = oracle.solve_poisson(W.astype(np.complex128)).copy()
    scale = np.abs(P_f64).max()
    e_dev, e_ora = maxabs(P_dev, P_f64) / scale, maxabs(P_ora, P_f64) / scale
    # float32 Thomas sweeps along diagonals of up to N entries: rounding errors add up over the sweep
    bound = 4 * N * EPS32
    assert e_ora <= bound, (N, e_ora / EPS32)
    assert e_dev <= max(bound, 2 * e_ora), (N, e_dev / EPS32, e_ora / EPS32)
    assert maxabs(P_dev, P_ora) / scale <= 2 * bound
    assert np.array_equal(P_dev, -P_dev.conj().T)
    # the general (not skew-Hermitian) branch solves the lower diagonals too
    A = (W + 0.5 * np.triu(W, 1)).astype(np.complex64)
    old = qfa.laplacian.select_skewherm(False)
    try:
        Pg = qfa.solve_poisson(A).copy()
    finally:
        qfa.laplacian.select_skewherm(old)
    old = oracle.select_skewherm(False)
    try:
        Pg_ref = oracle.solve_poisson(A.astype(np.complex128)).copy()
    finally:
        oracle.select_skewherm(old)
    assert maxabs(Pg, Pg_ref) <= 2 * bound * np.abs(Pg_ref).max()     # (a matrix with a heavier upper triangle)


@pytest.mark.parametrize("N", [32, 64, 100, 128, 333, 512, 736, 768, 1000, 1024])
def test_cgemm_vs_numpy(qfa, N):
    """C = A @ B on the fp32 matrix cores (3M form) against the double-precision product of the same complex64
    operands: k-ordered float32 fma chains, normwise the error of cgemm."""
    import ctypes
    from quflow_amd import _lib
    from quflow_amd.context import Context, ptr
    rng = np.random.default_rng(N)
    A = (rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))).astype(np.complex64)
    B = (rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))).astype(np.complex64)
    B[:, 0] = 0                       # asymmetric structure: catches a transposed or shifted tile
    A[1, :] *= 3
    C = np.zeros((N, N), dtype=np.complex64)
    ctx = Context(N)
    try:
        _lib.check(ctx._lib.qf_cgemm(ctx.handle, ptr(A), ptr(B), ptr(C)))
    finally:
        ctx.close()
    ref = A.astype(np.complex128) @ B.astype(np.complex128)
    bound = 4 * EPS32 * (np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)).max() * np.sqrt(N)
    assert maxabs(C, ref) <= bound, maxabs(C, ref) / bound
    assert np.all(C[:, 0] == 0)


@pytest.mark.parametrize("N", [64, 96, 100, 256, 512, 1000, 1024])
def test_fixedpoint_products_c64(qfa, N):
    """One iteration's two products with the fused epilogue on complex64 operands (isospectral.py:496-509,
    481-482, 526-534) against numpy in double precision."""
    from quflow_amd import _lib
    from quflow_amd.context import Context, ptr
    rng = np.random.default_rng(N + 1)

    def skew(scale):
        A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
        A = A - A.conj().T
        return (A * (scale / np.abs(A).max())).astype(np.complex64)
    P, W, dW_old = skew(0.05), skew(1.0), skew(0.01)
    Whalf = (W + dW_old).astype(np.complex64)
    P64, W64, d64, Wh64 = (x.astype(np.complex128) for x in (P, W, dW_old, Whalf))
    PW = P64 @ Wh64
    dW_ref = PW @ P64 + (PW - PW.conj().T)
    rows_ref = np.abs(d64 - dW_ref).sum(axis=1)
    dW = np.zeros_like(W)
    Wh = np.zeros_like(W)
    rows = np.zeros(N)
    ctx = Context(N)
    try:
        _lib.check(ctx._lib.qf_c64_fixedpoint_products(ctx.handle, ptr(P), ptr(Whalf), ptr(W), ptr(dW_old), ptr(dW), ptr(Wh),
                                                       ptr(rows)))
    finally:
        ctx.close()
    bound = 16 * EPS32 * np.sqrt(N) * (np.abs(PW) @ np.abs(P64)).max() + 8 * EPS32 * np.abs(PW).max()
    assert maxabs(dW, dW_ref) <= bound
    assert maxabs(Wh, W64 + dW.astype(np.complex128)) <= 2 * EPS32 * np.abs(W).max()
    assert np.abs(rows - rows_ref).max() <= N * (bound + 4 * EPS32 * np.abs(dW_old).max())


@pytest.mark.parametrize("N", [64, 72, 96, 100, 128, 160, 224, 256, 333, 512, 736, 768, 832, 1000, 1001, 1024, 1056, 1536, 2048])
def test_fixedpoint_products_c64_tri(qfa, N):
    """The complex64 second product on the upper triangle of 32 x 32 tiles (k_cgemm_tri32; exact tilings and guarded edge
    tiles), every tile's K range cut into pieces: against numpy in double precision and against the full product; exactly
    skew-Hermitian dW and (outside the diagonal tiles) Whalf; the same bits on every run, whichever piece arrives last."""
    from quflow_amd import _lib
    from quflow_amd.context import Context, ptr
    rng = np.random.default_rng(N + 2)

    def skew(scale):
        A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
        A = A - A.conj().T
        return (A * (scale / np.abs(A).max())).astype(np.complex64)
    P, W, dW_old = skew(0.05), skew(1.0), skew(0.01)
    Whalf = (W + dW_old).astype(np.complex64)
    assert np.array_equal(Whalf, -Whalf.conj().T)
    P64, W64, d64, Wh64 = (x.astype(np.complex128) for x in (P, W, dW_old, Whalf))
    PW = P64 @ Wh64
    dW_ref = PW @ P64 + (PW - PW.conj().T)
    rows_ref = np.abs(d64 - dW_ref).sum(axis=1)
    bound = 16 * EPS32 * np.sqrt(N) * (np.abs(PW) @ np.abs(P64)).max() + 8 * EPS32 * np.abs(PW).max()
    tile = 32
    ctx = Context(N)
    runs = []
    try:
        for variant in ("full", "tri", "tri", "tri"):
            dW = np.zeros_like(W)
            Wh = np.zeros_like(W)
            rows = np.zeros(N)
            fn = ctx._lib.qf_c64_fixedpoint_products if variant == "full" else ctx._lib.qf_c64_fixedpoint_products_tri
            _lib.check(fn(ctx.handle, ptr(P), ptr(Whalf), ptr(W), ptr(dW_old), ptr(dW), ptr(Wh), ptr(rows)))
            assert maxabs(dW, dW_ref) <= bound, variant
            # (below the diagonal inside a diagonal tile Whalf carries the product's own value, the returned dW its
            # mirror image: compared on and above the diagonal, the rest through the skew-symmetry checks below)
            up = np.triu(np.ones((N, N), dtype=bool))
            assert maxabs(Wh[up], (W64 + dW.astype(np.complex128))[up]) <= 2 * EPS32 * np.abs(W).max(), variant
            assert maxabs(Wh, W64 + dW.astype(np.complex128)) <= 2 * EPS32 * np.abs(W).max() + bound, variant
            assert np.abs(rows - rows_ref).max() <= N * (bound + 4 * EPS32 * np.abs(dW_old).max()), variant
            runs.append((dW, Wh, rows))
    finally:
        ctx.close()
    for a, b in zip(runs[1], runs[2]):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(runs[1], runs[3]):
        np.testing.assert_array_equal(a, b)
    dW, Wh, _ = runs[1]
    offd = ~np.eye(N, dtype=bool)         # (a diagonal entry keeps the product's own real part: rounding noise around 0)
    assert np.array_equal(dW[offd], (-dW.conj().T)[offd])
    blk = np.arange(N) // tile
    off = blk[:, None] != blk[None, :]
    assert np.array_equal(Wh[off], (-Wh.conj().T)[off])
    assert maxabs(runs[0][0], dW) <= bound


def test_isomp_c64_reference_vectors(qfa):
    """The stepper on complex64 input against the reference's own complex64 runs: in place, complex64, the float32
    automatic tolerance, identical iteration counts, states within 1e-5 of the state's scale."""
    g = load_golden("single_precision")
    cases = [("plain", g["W0"], int(g["N"]), 8, {}), ("compsum", g["W0"], int(g["N"]), 8, {"compsum": True}),
             ("tol1e-3", g["W0"], int(g["N"]), 8, {"tol": 1e-3}),
             ("N64_plain", g["N64_W0"], 64, 12, {}), ("N64_compsum", g["N64_W0"], 64, 12, {"compsum": True})]
    for tag, W0, N, steps, kw in cases:
        W = W0.copy()
        stats = {"iterations": 0.0}
        out = qfa.isomp(W, 0.25 * qfa.hbar(N), steps=steps, stats=stats, **kw)
        assert out is W and W.dtype == np.complex64
        ref = g[tag + "_W"]
        assert maxabs(W, ref) <= 1e-5 * np.abs(ref).max(), (tag, maxabs(W, ref) / np.abs(ref).max())
        if tag + "_tol" in g.files:
            np.testing.assert_allclose(stats["tol_auto"], float(g[tag + "_tol"]), rtol=1e-6)
        else:
            assert "tol_auto" not in stats
        # the plain exit threshold (sqrt(eps32)) sits far above float32 noise: identical counts; the compensated
        # run's threshold is eps32 itself, reached within the reference's own rounding noise
        slack = 0.0 if "compsum" not in tag else 0.5
        assert abs(stats["iterations"] - float(g[tag + "_iterations"])) <= slack, (tag, stats["iterations"])
        assert stats["number_of_maxit"] == float(g[tag + "_maxit"])
        assert np.array_equal(W, -W.conj().T)


@pytest.mark.parametrize("N,steps", [(128, 10), (500, 4), (512, 4), (1024, 2), (736, 3), (800, 2), (1056, 2), (1088, 2), (1000, 2), (333, 6), (101, 10), (2048, 1)])
def test_isomp_c64_vs_oracle_large(qfa, oracle, N, steps):
    W0 = make_W0_c64(oracle, N, 2)
    dt = 0.25 * qfa.hbar(N)
    sg, so = {"iterations": 0.0}, {"iterations": 0.0}
    Wg = qfa.isomp(W0.copy(), dt, steps=steps, stats=sg)
    Wo = oracle.isomp(W0.copy(), dt, steps=steps, stats=so)
    assert Wg.dtype == np.complex64
    assert sg["iterations"] == so["iterations"] and sg["number_of_maxit"] == so["number_of_maxit"]
    np.testing.assert_allclose(sg["tol_auto"], so["tol_auto"], rtol=1e-6)
    assert maxabs(Wg, Wo) <= 1e-5 * np.abs(Wo).max()
    # and both sit within float32 rounding of the double-precision trajectory
    W64 = oracle.isomp(W0.astype(np.complex128), dt, steps=steps)
    assert maxabs(Wg, W64) <= 2e-5 * np.abs(W64).max()


@pytest.mark.parametrize("N", [64, 100, 512, 768, 1000])
def test_isomp_c64_fused_step_end_is_bit_identical(qfa, oracle, monkeypatch, N):
    """The fused step end (three launches per iteration, the W update and the exit decision in the second product's
    last tile) against the two-kernel protocol: same arithmetic on the same values, so the same bits, counts and
    tolerance; a chunked call (dW restarts) and a continued resident run likewise."""
    from quflow_amd.context import release_contexts
    W0 = make_W0_c64(oracle, N, 5)
    dt = 0.25 * qfa.hbar(N)
    out = {}
    monkeypatch.setenv("QUFLOW_HIP_GEMM2", "full")        # (N = 768: the same second product in both protocols)
    try:
        for fused in ("1", "0"):
            monkeypatch.setenv("QUFLOW_HIP_FUSED", fused)
            release_contexts()
            st = {"iterations": 0.0}
            W = qfa.isomp(W0.copy(), dt, steps=6, stats=st)
            W = qfa.isomp(W, dt, steps=3, maxit=2, stats=st)
            tr = qfa.DeviceTrajectory(W0)
            a = tr.advance(dt, 4)
            b = tr.advance(dt, 3)
            out[fused] = (W, st["iterations"], st["number_of_maxit"], st["tol_auto"], tr.download(), a["total_iterations"],
                          b["total_iterations"])
            tr.ctx.close()
    finally:
        release_contexts()
    for x, y in zip(out["1"], out["0"]):
        if isinstance(x, np.ndarray):
            np.testing.assert_array_equal(x, y)
        else:
            assert x == y
    assert np.array_equal(out["1"][0], -out["1"][0].conj().T)


@pytest.mark.parametrize("N,steps", [(768, 6), (1024, 4), (832, 3), (1088, 3), (1280, 2), (1600, 2), (64, 30), (256, 10),
                                     (512, 6), (736, 4)])
def test_isomp_c64_triangle_product_vs_full(qfa, oracle, monkeypatch, N, steps):
    """The stepper with the upper-triangle second product (default from N = 768) against the full product: equal to
    float32 rounding, same iteration counts, W exactly skew-Hermitian with both triangles in place after the call,
    a continued resident run (carried increment: dW restored by its mirror) likewise."""
    from quflow_amd.context import release_contexts
    W0 = make_W0_c64(oracle, N, 8)
    dt = 0.25 * qfa.hbar(N)
    out = {}
    try:
        for mode in ("tri", "full"):
            monkeypatch.setenv("QUFLOW_HIP_GEMM2", mode)
            release_contexts()
            st = {"iterations": 0.0}
            W = qfa.isomp(W0.copy(), dt, steps=steps, stats=st)
            tr = qfa.DeviceTrajectory(W0)
            a = tr.advance(dt, 2)
            b = tr.advance(dt, steps - 2)
            Wt = tr.download()
            tr.ctx.close()
            assert np.array_equal(W, -W.conj().T) and np.array_equal(Wt, -Wt.conj().T)
            out[mode] = (W, st["iterations"], st["number_of_maxit"], Wt, a["total_iterations"] + b["total_iterations"])
    finally:
        release_contexts()
    scale = np.abs(out["full"][0]).max()
    assert maxabs(out["tri"][0], out["full"][0]) <= 1e-5 * scale
    assert maxabs(out["tri"][3], out["full"][3]) <= 1e-5 * scale
    assert out["tri"][1] == out["full"][1] and out["tri"][2] == out["full"][2] and out["tri"][4] == out["full"][4]
    W64 = oracle.isomp(W0.astype(np.complex128), dt, steps=steps)
    assert maxabs(out["tri"][0], W64) <= 2e-5 * np.abs(W64).max()


@pytest.mark.parametrize("N", [64, 512, 1024])
def test_c64_device_ensemble_members_are_bit_identical_to_single_runs(qfa, oracle, N):
    """k complex64 replicas advanced together on one GPU (qf_c64_isomp_multi): every member equals its own
    single-trajectory run bit for bit over chunked calls; a mixed-dtype ensemble is refused."""
    k = 4 if N < 1024 else 3
    W0s = [make_W0_c64(oracle, N, 60 + r) for r in range(k)]
    dt = 0.25 * qfa.hbar(N)
    steps = 6 if N >= 512 else 20
    ens = qfa.DeviceEnsemble(W0s)
    assert ens.c64
    st_a = ens.advance(dt, steps)
    st_b = ens.advance(dt, steps)
    got = ens.download()
    diag = ens.diagnostics()
    ens.close()
    for r in range(k):
        tr = qfa.DeviceTrajectory(W0s[r])
        s1 = tr.advance(dt, steps)
        s2 = tr.advance(dt, steps)
        W = tr.download()
        assert got[r].dtype == np.complex64
        np.testing.assert_array_equal(got[r], W)
        assert (st_a[r]["total_iterations"], st_b[r]["total_iterations"]) == (s1["total_iterations"], s2["total_iterations"])
        assert st_a[r]["tol"] == s1["tol"] and st_b[r]["tol"] == s2["tol"]
        assert diag[r] == tr.diagnostics()
        tr.ctx.close()
    with pytest.raises(ValueError):
        qfa.DeviceEnsemble([W0s[0], W0s[1].astype(np.complex128)])


def test_c64_trajectory_resident(qfa, oracle):
    """DeviceTrajectory on a complex64 state: single precision on the device, chunked calls restart the iteration
    vector like host-array calls, diagnostics within float32 rounding of the double-precision ones."""
    N = 256
    W0 = make_W0_c64(oracle, N, 9)
    dt = 0.25 * qfa.hbar(N)
    tr = qfa.DeviceTrajectory(W0)
    assert tr.c64 and tr.dtype == np.complex64
    st = tr.advance(dt, 5, diagnostics=True)
    st2 = tr.advance(dt, 5)
    W = tr.download()
    assert W.dtype == np.complex64
    Wh = W0.copy()
    qfa.isomp(Wh, dt, steps=5)
    qfa.isomp(Wh, dt, steps=5)
    np.testing.assert_array_equal(W, Wh)                 # same launches either way
    assert st["total_iterations"] > 0 and st2["total_iterations"] > 0
    e64, s64 = oracle.energy_euler(W.astype(np.complex128)), oracle.enstrophy(W.astype(np.complex128))
    e, s = tr.diagnostics()
    assert abs(e - e64) <= 1e-5 * abs(e64) and abs(s - s64) <= 1e-5 * abs(s64)
    with pytest.raises(NotImplementedError):
        tr.advance_erk("rk4", dt, 1)
    tr.ctx.close()


def test_c64_double_precision_escape(qfa, oracle, monkeypatch):
    """QUFLOW_HIP_C64=f64: complex64 in and out, evaluated in double precision on the device (round 2's behaviour)."""
    g = load_golden("single_precision")
    monkeypatch.setenv("QUFLOW_HIP_C64", "f64")
    P = qfa.solve_poisson(g["W0"])
    assert P.dtype == np.complex64
    P64 = oracle.solve_poisson(g["W0"].astype(np.complex128))
    assert maxabs(P, P64) <= 2 * EPS32 * np.abs(P64).max()


def test_other_tridiagonal_solves_c64(qfa, oracle):
    """solve_helmholtz / solve_heat / solve_viscdamp / solve_globalqg on complex64 input: float32 tables built as the
    reference builds them (cpu.py:760, 809: `laplacian(N, dtype=type(W[0,0].real))`, then `tab -= c * lap` in
    float32) and the float32 solve -- against the reference's own complex64 results and, at a larger size, within
    float32 rounding of the double-precision oracle."""
    g = load_golden("single_precision")
    W = g["N48_W0"]
    for got, ref in ((qfa.solve_helmholtz(W, alpha=0.37), g["N48_helmholtz"]), (qfa.solve_heat(0.013, W), g["N48_heat"])):
        assert got.dtype == np.complex64
        assert maxabs(got, ref) <= 32 * EPS32 * np.abs(ref).max(), maxabs(got, ref) / (EPS32 * np.abs(ref).max())
    N = 256
    W = make_W0_c64(oracle, N, 4)
    W64 = W.astype(np.complex128)
    cases = [(qfa.solve_helmholtz(W, 0.5), oracle.solve_helmholtz(W64, 0.5)),
             (qfa.solve_heat(0.02, W), oracle.solve_heat(0.02, W64)),
             (qfa.solve_viscdamp(0.1, W, nu=1e-3, alpha=0.05), oracle.solve_viscdamp(0.1, W64, nu=1e-3, alpha=0.05)),
             (qfa.solve_viscdamp(0.1, W, nu=1e-3, alpha=0.05, theta=0.5), oracle.solve_viscdamp(0.1, W64, nu=1e-3, alpha=0.05, theta=0.5)),
             (qfa.solve_globalqg(W, gamma=2.0), oracle.solve_globalqg(W64, gamma=2.0))]
    for got, ref in cases:
        assert got.dtype == np.complex64
        assert maxabs(got, ref) <= 4 * N * EPS32 * np.abs(ref).max()
