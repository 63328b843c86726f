"""GPU tests of the ENVELOPE of the hot path (round 6): every size qf_ctx_create accepts above the BASELINE sizes, the
remaining names of the Laplacian backend module, and the threading sentence of include/quflow_hip.h ("independent ctxs
may run concurrently" -- from separate host threads of one process).

Sizes: N = 3072, 4096 and 8192 (the limit, csrc/api_context.hip) against the oracle on the same seeded input; the oracle costs
~0.5 / 1 / 8 s per fixed-point iteration there on the GPU box's 16 cores, so one step each.  N > 2175 is the
`k_solve<double, L=32>` layout of the Laplacian inverse (no folded walk slots), N = 8192 its largest grid.
"""
import hashlib
import threading

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

EPS = np.finfo(float).eps
STEP_TOL = 1e-11


@pytest.fixture(scope="module")
def qfa():
    import quflow_amd
    if quflow_amd.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the MI355X box")
    return quflow_amd


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))))


# ----------------------------------------------------------------------------- reduce= of solve_poisson
@pytest.mark.parametrize("N", [17, 33])
def test_solve_poisson_reduce_golden(qfa, N):
    """solve_poisson(W_stack, reduce=select_first | select_sum) (quflow/laplacian/cpu.py:672-698) against the
    reference's outputs (tests/golden/reduce.npz), and allocate_buffer as the no-result warm-up it is there."""
    g = load_golden("reduce")
    pre = "N%d_" % N
    S = g[pre + "S"]
    lap = qfa.laplacian
    # (solve_poisson hands back ONE persistent array, cpu.py:726: each result is looked at before the next call)
    for call, key in ((lambda: lap.solve_poisson(S), "P_default"), (lambda: lap.solve_poisson(S, reduce=lap.select_first), "P_first"),
                      (lambda: lap.solve_poisson(S, reduce=lap.select_sum), "P_sum"),
                      (lambda: lap.solve_poisson(np.stack([S, 2.0 * S]), reduce=lap.select_sum), "P_sum4")):
        ref = g[pre + key]
        err = maxabs(call(), ref)
        assert err <= 1e-14 * N ** 2 and err <= 64 * EPS * np.abs(ref).max(), (key, err)
    assert lap.allocate_buffer(S[0]) is None
    assert maxabs(lap.solve_poisson(S), g[pre + "P_default"]) <= 64 * EPS * np.abs(g[pre + "P_default"]).max()
    # complex64 stacks reduce the same way and keep the reference's single-precision solve
    P32 = lap.solve_poisson(S.astype(np.complex64), reduce=lap.select_sum)
    assert P32.dtype == np.complex64
    assert maxabs(P32, g[pre + "P_sum"]) <= 2e-5 * np.abs(g[pre + "P_sum"]).max()


# ----------------------------------------------------------------------------- sizes above the BASELINE configs
@pytest.mark.parametrize("N", [2176, 3072, 4096, 8192])
def test_solve_poisson_vs_oracle_beyond_2048(qfa, oracle, N):
    """The Laplacian inverse on the chunk-32 layout (N + 1 > 17 * 128: no folded walk slots) up to the largest N a
    context accepts, against the oracle; laplace(solve(W)) == W; the general (non-skew-Hermitian) solve as well."""
    W = oracle.make_W0(N, 11)
    P = qfa.solve_poisson(W).copy()
    Pc = oracle.solve_poisson(W).copy()
    scale = np.abs(Pc).max()
    err = maxabs(P, Pc)
    assert err <= 1e-14 * N ** 2
    assert err <= 256 * EPS * scale, (err, scale)
    np.testing.assert_array_equal(P, -P.conj().T)
    back = qfa.laplace(P)
    assert maxabs(back, W) <= 1e-9 * np.abs(W).max()
    old = qfa.laplacian.select_skewherm(False)
    try:
        Pg = qfa.solve_poisson(W).copy()
    finally:
        qfa.laplacian.select_skewherm(old)
    assert maxabs(Pg, Pc) <= 256 * EPS * scale
    # (which k_solve layout ran is asserted from the stepper's plan in test_isomp_vs_oracle_beyond_2048)
    from quflow_amd.context import release_contexts
    release_contexts()


@pytest.mark.parametrize("N,steps", [(3072, 2), (4096, 1), (8192, 1)])
def test_isomp_vs_oracle_beyond_2048(qfa, oracle, N, steps):
    """The default stepper at the sizes above BASELINE.json's, up to the limit of qf_ctx_create (N = 3072 with a second,
    warm-started step): state, iteration count and tolerance against the oracle on identical W0; Casimir drift no worse than the CPU run's; energy and enstrophy
    from the device diagnostics against the oracle's."""
    W0 = oracle.make_W0(N, 0)
    dt = 0.25 * qfa.hbar(N)
    sg, sc = {"iterations": 0.0}, {"iterations": 0.0}
    tr = qfa.DeviceTrajectory(W0)
    st = tr.advance(dt, steps, diagnostics=True)
    Wg = tr.download()
    plan = tr.ctx.plan()
    tr.ctx.close()
    Wc = oracle.isomp(W0.copy(), dt, steps=steps, stats=sc)
    assert maxabs(Wg, Wc) <= STEP_TOL
    assert st["iterations"] == sc["iterations"]
    np.testing.assert_allclose(st["tol"], sc["tol_auto"], rtol=1e-12)
    np.testing.assert_array_equal(Wg, -Wg.conj().T)
    np.testing.assert_allclose(st["energy"], oracle.energy_euler(Wc), rtol=1e-10)
    np.testing.assert_allclose(st["enstrophy"], oracle.enstrophy(Wc), rtol=1e-12)
    assert plan["laplacian_inverse"]["kernel"].startswith("k_solve<double, L=32")
    import os
    if not os.environ.get("QUFLOW_HIP_GEMM"):        # (the suite also runs under QUFLOW_HIP_GEMM=auto: config 3's kernels up to N = 4096)
        assert plan["first_product"]["kernel"] == "k_zgemm<64,64>" and plan["second_product"]["kernel"] == "k_zgemm_tri"
    if N <= 4096:
        # (three N^3 products per state in numpy: seconds at 4096, a minute at 8192 -- the state bound above covers that size)
        c0 = oracle.casimirs(W0)
        dg = np.abs(oracle.casimirs(Wg) - c0).max()
        dc = np.abs(oracle.casimirs(Wc) - c0).max()
        res = np.sqrt(N) * EPS * 2.0
        assert dg <= 1.05 * dc + res, (dg, dc, res)
    # the host-array entry point at the same size: the same call, bit for bit
    if N <= 4096:
        Wh = qfa.isomp(W0.copy(), dt, steps=steps, stats=sg)
        np.testing.assert_array_equal(Wh, Wg)
        assert sg["iterations"] == sc["iterations"]
        from quflow_amd.context import release_contexts
        release_contexts()


@pytest.mark.parametrize("N", [3072, 4096])
def test_config3_products_at_their_largest_sizes(qfa, oracle, N, monkeypatch):
    """Config 3's int8 digit-split products (i8x65) up to N = 4096, the largest size their slicing kernel takes
    (csrc/ozaki.hip): one step against the oracle -- identical iteration count, state within the fp64 path's bar."""
    from quflow_amd.context import release_contexts
    monkeypatch.setenv("QUFLOW_HIP_GEMM", "i8x65")
    release_contexts()
    try:
        W0 = oracle.make_W0(N, 0)
        dt = 0.25 * qfa.hbar(N)
        tr = qfa.DeviceTrajectory(W0)
        st = tr.advance(dt, 1)
        Wg = tr.download()
        plan = tr.ctx.plan()
        tr.ctx.close()
    finally:
        release_contexts()
    sc = {"iterations": 0.0}
    Wc = oracle.isomp(W0.copy(), dt, steps=1, stats=sc)
    assert plan["first_product"]["kernel"].startswith("k_oz_gemm") and plan["second_product"]["kernel"].startswith("k_oz_gemm"), plan
    assert st["iterations"] == sc["iterations"]
    assert maxabs(Wg, Wc) <= STEP_TOL
    np.testing.assert_array_equal(Wg, -Wg.conj().T)


def test_sizes_outside_the_envelope_are_refused(qfa):
    """What qf_ctx_create does not accept is an error return with a message, not a fault: N < 2, N > 8192; the int8
    products above N = 4096 fall back to the fp64 kernels by the selection rule (never a silent wrong size)."""
    import ctypes
    from quflow_amd import _lib
    lib = _lib.load()
    for N in (0, 1, 8193, 1 << 20):
        h = ctypes.c_void_p()
        rc = lib.qf_ctx_create(N, 0, ctypes.byref(h))
        assert rc == 1 and b"out of range" in lib.qf_last_error(), (N, rc, lib.qf_last_error())
        assert not h.value


# ----------------------------------------------------------------------------- host threads
def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _chunk_chain(qfa, N, seed, chunks, steps, barrier=None):
    """One DeviceTrajectory advanced `chunks` times; per chunk: sha256 of the state, the statistics, the diagnostics."""
    W0 = qfa.ensemble.make_W0(N, seed)
    dt = 0.25 * qfa.hbar(N)
    tr = qfa.DeviceTrajectory(W0)
    rows = []
    if barrier is not None:
        barrier.wait()
    for c in range(chunks):
        st = tr.advance(dt, steps, diagnostics=True)
        rows.append((_digest(tr.download()), st["total_iterations"], st["number_of_maxit"], st["tol"], st["last_resnorm"],
                     st["energy"], st["enstrophy"]))
    tr.ctx.close()
    return rows


def test_two_host_threads_drive_independent_contexts(qfa):
    """include/quflow_hip.h: "one ctx is used by one host thread at a time; independent ctxs may run concurrently".
    Two Python threads (ctypes releases the GIL inside every call), each with its own DeviceTrajectory on device 0 --
    N = 512 and N = 1024, so the two drive DIFFERENT kernels and raise the LDS limit of the SAME k_solve template
    concurrently -- advance 50 chunks at the same time.  Every chunk's state (sha256 of the downloaded bytes),
    iteration statistics, tolerance, residual and diagnostics equal the sequential runs bit for bit."""
    cases = [(512, 71, 50, 4), (1024, 72, 50, 3)]
    sequential = [_chunk_chain(qfa, *c) for c in cases]
    barrier = threading.Barrier(len(cases))
    results, errors = [None] * len(cases), []

    def work(i):
        try:
            results[i] = _chunk_chain(qfa, *cases[i], barrier=barrier)
        except BaseException as e:      # noqa: BLE001  (reported by the assert below)
            errors.append((i, repr(e)))
            try:
                barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(cases))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads)
    for i, c in enumerate(cases):
        assert len(results[i]) == c[2]
        assert results[i] == sequential[i], "N=%d: the concurrent run differs from the sequential one at chunk %d" % (
            c[0], next(k for k, (a, b) in enumerate(zip(results[i], sequential[i])) if a != b))


def test_same_size_trajectories_from_four_threads(qfa):
    """Four threads, four trajectories of ONE size (N = 256: launch-bound, the host loops interleave most): each
    equals its sequential run; the shared function-local records of the launchers are not a channel between them."""
    cases = [(256, 80 + r, 25, 8) for r in range(4)]
    sequential = [_chunk_chain(qfa, *c) for c in cases]
    barrier = threading.Barrier(len(cases))
    results, errors = [None] * len(cases), []

    def work(i):
        try:
            results[i] = _chunk_chain(qfa, *cases[i], barrier=barrier)
        except BaseException as e:      # noqa: BLE001
            errors.append((i, repr(e)))
            try:
                barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(cases))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors, errors
    assert results == sequential


def test_last_error_is_per_thread(qfa):
    """qf_last_error() is thread-local: an error return in one thread leaves the other thread's message untouched,
    and a failing call in one thread does not disturb a trajectory another thread is advancing."""
    import ctypes
    from quflow_amd import _lib
    lib = _lib.load()
    step = threading.Barrier(2)
    seen, errors = {}, []

    def a():
        try:
            h = ctypes.c_void_p()
            assert lib.qf_ctx_create(9999, 0, ctypes.byref(h)) == 1
            seen["a_first"] = lib.qf_last_error()
            step.wait(60)                 # (1) both have failed once
            step.wait(60)                 # (2) b failed again in between
            seen["a_after_b"] = lib.qf_last_error()
        except BaseException as e:        # noqa: BLE001
            errors.append(repr(e))
            step.abort()

    def b():
        try:
            N = 128
            W0 = qfa.ensemble.make_W0(N, 3)
            tr = qfa.DeviceTrajectory(W0)
            s1 = tr.advance(0.25 * qfa.hbar(N), 5)
            step.wait(60)                 # (1)
            h = ctypes.c_void_p()
            assert lib.qf_ctx_create(1, 0, ctypes.byref(h)) == 1
            seen["b"] = lib.qf_last_error()
            s2 = tr.advance(0.25 * qfa.hbar(N), 5)        # a failure elsewhere does not touch this context
            seen["b_state"] = _digest(tr.download())
            seen["b_stats"] = (s1["total_iterations"], s2["total_iterations"])
            tr.ctx.close()
            step.wait(60)                 # (2)
        except BaseException as e:        # noqa: BLE001
            errors.append(repr(e))
            step.abort()

    ta, tb = threading.Thread(target=a), threading.Thread(target=b)
    ta.start(); tb.start()
    ta.join(300); tb.join(300)
    assert not errors, errors
    assert b"N=9999" in seen["a_first"] and seen["a_after_b"] == seen["a_first"]
    assert b"N=1 " in seen["b"] and b"9999" not in seen["b"]
    N = 128
    tr = qfa.DeviceTrajectory(qfa.ensemble.make_W0(N, 3))
    r1 = tr.advance(0.25 * qfa.hbar(N), 5)
    r2 = tr.advance(0.25 * qfa.hbar(N), 5)
    assert seen["b_state"] == _digest(tr.download())
    assert seen["b_stats"] == (r1["total_iterations"], r2["total_iterations"])
    tr.ctx.close()


# ----------------------------------------------------------------------------- a residual that stops being finite mid-call
@pytest.mark.parametrize("N,kw,aligned", [(64, {}, True), (512, {}, True), (1024, {}, True), (64, {"compsum": True}, True),
                                          (256, {"reinitialize": True}, False), (1024, {"products": "i8x65"}, True)])
def test_nonfinite_residual_mid_call_keeps_the_last_completed_step(qfa, oracle, N, kw, aligned, monkeypatch):
    """include/quflow_hip.h, QF_ERR_NONFINITE: "the state is left as it was after the last completed step" -- what the
    reference leaves when scipy.linalg.norm raises inside its exit test (isospectral.py:534: W is updated in place at the END
    of a step, :592).  A finite state that blows up a few steps into ONE call (dt = 1e7 hbar, one unconverged iteration per
    step: the exponent of |W| triples every step): the device closes the call at that iteration -- nothing queued behind it
    runs, no NaN is written into W -- and the resident state is the oracle's state after m >= 1 completed steps, on the fused
    (N = 1024), deferred (N <= 512) and two-kernel (compsum / reinitialize) step ends and with the int8 products.
    `aligned`: the residual goes from below 1e154 to inf within one step, so the device (whose residual entries overflow
    where their SQUARES do, qf_modulus: above 1.3e154) and the oracle (numpy's scaled abs: above 1.8e308) stop in the SAME
    step and m is the oracle's own count; otherwise the device may stop a step earlier, still on a completed step."""
    import warnings
    from quflow_amd.context import release_contexts
    kw = dict(kw)
    products = kw.pop("products", None)
    if products:
        monkeypatch.setenv("QUFLOW_HIP_GEMM", products)
        release_contexts()
    W0 = oracle.make_W0(N, 0)
    dt = 1e7 * qfa.hbar(N)
    opts = dict(minit=1, maxit=1, **kw)
    states = []                                  # the oracle's state after m completed steps of ONE call, m = 1, 2, ...
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for m in range(1, 12):
            Wc = W0.copy()
            try:
                oracle.isomp(Wc, dt, steps=m, **opts)
            except ValueError as e:
                assert "infs or NaNs" in str(e)
                break                            # (in place: Wc is the state after the last completed step = states[-1])
            states.append(Wc)
    assert 2 <= len(states) <= 9 and np.isfinite(states[-1]).all()
    tr = qfa.DeviceTrajectory(W0)
    try:
        with pytest.raises(ValueError, match="infs or NaNs"):
            tr.advance(dt, 12, **opts)
        Wg = tr.download()
        assert np.isfinite(Wg).all()
        np.testing.assert_array_equal(Wg, -Wg.conj().T)
        import os
        # (int8 digit-split products -- named here or through QUFLOW_HIP_GEMM=auto, under which the suite also runs -- carry 2^-35
        # per product, and the unconverged map triples relative differences every step)
        rtol = 1e-9 if not (products or os.environ.get("QUFLOW_HIP_GEMM")) else 1e-6
        rel = [maxabs(Wg, Wm) / np.abs(Wm).max() for Wm in states]
        m = int(np.argmin(rel)) + 1
        assert rel[m - 1] <= rtol, (rel, [float(np.abs(Wm).max()) for Wm in states], float(np.abs(Wg).max()))
        assert m >= 2, "the call was closed before a step had completed"
        if aligned:
            assert m == len(states), (m, len(states))
        # the context keeps working: a fresh state, two ordinary steps against the oracle
        tr.upload(W0)
        s = tr.advance(0.25 * qfa.hbar(N), 2)
        sc = {"iterations": 0.0}
        Wo = oracle.isomp(W0.copy(), 0.25 * qfa.hbar(N), steps=2, stats=sc)
        assert maxabs(tr.download(), Wo) <= STEP_TOL and s["iterations"] == sc["iterations"]
    finally:
        tr.ctx.close()
        if products:
            release_contexts()
    # the host-array entry point: the caller's array is not overwritten with a half-done or NaN state
    if not products:
        Wh = W0.copy()
        with pytest.raises(ValueError, match="infs or NaNs"):
            qfa.isomp(Wh, dt, steps=12, **opts)
        np.testing.assert_array_equal(Wh, W0)


def test_nonfinite_entry_in_a_passive_state_of_a_stack_raises(qfa, oracle):
    """The reference's exit test hands the WHOLE (k,N,N) residual to scipy.linalg.norm(..., axis=(-1,-2)), whose check_finite
    raises for a NaN / inf in ANY state (isospectral.py:528) although only state 0's norm decides (shared stream matrix).  Both
    host loops of the device path (qf_isomp_states; qf_isomp_hooked with a forcing) form every state's residual norm for that
    check; the oracle raises on the same input."""
    N = 48
    dt = 0.25 * qfa.hbar(N)
    W0 = oracle.make_W0(N, 0)
    for poison in (np.nan, np.inf):
        Wb = oracle.make_W0(N, 1)
        Wb[3, 5] = poison
        Wb[5, 3] = poison
        for S in (np.stack([W0, Wb]), np.stack([W0, oracle.make_W0(N, 2), Wb])):
            with pytest.raises(ValueError, match="infs or NaNs"):
                oracle.isomp(S.copy(), dt, steps=2)
            with pytest.raises(ValueError, match="infs or NaNs"):
                qfa.isomp(S.copy(), dt, steps=2)                                          # qf_isomp_states
            with pytest.raises(ValueError, match="infs or NaNs"):
                qfa.isomp(S.copy(), dt, steps=2, forcing=lambda P, W: 0.0 * np.nan_to_num(W))   # qf_isomp_hooked
    # magmp: the second state (theta) of the hooked loop is checked too
    St = np.stack([W0, Wb])
    with pytest.raises(ValueError, match="infs or NaNs"):
        oracle.magmp_fixedpoint(St.copy(), dt, steps=2)
    with pytest.raises(ValueError, match="infs or NaNs"):
        qfa.magmp(St.copy(), dt, steps=2)                                                  # qf_isomp_states, magnetic
    with pytest.raises(ValueError, match="infs or NaNs"):
        qfa.magmp(St.copy(), dt, steps=2, forcing=lambda P, state: 0.0 * np.nan_to_num(state))   # qf_isomp_hooked, magnetic
    # and a clean stack still runs to the oracle's bits-level neighbourhood on the same contexts
    S = np.stack([W0, oracle.make_W0(N, 1)])
    sg, sc = {"iterations": 0.0}, {"iterations": 0.0}
    Wg = qfa.isomp(S.copy(), dt, steps=3, stats=sg)
    Wc = oracle.isomp(S.copy(), dt, steps=3, stats=sc)
    assert maxabs(Wg, Wc) <= STEP_TOL and sg["iterations"] == sc["iterations"]


def test_commutators_combine_on_the_device_to_the_hosts_bits(qfa):
    """qf_commutator (round 6): commutator_skewherm / commutator_generic (isospectral.py:22-57) form X - X^H / W@P - P@W on the
    device -- one PCIe round trip instead of a host pass over N^2 entries -- with the bits of the device product followed by
    numpy's elementwise subtraction (exact negation, one rounding)."""
    from quflow_amd.geometry import _device_matmul
    for N in (33, 64, 500, 1024):
        W = qfa.ensemble.make_W0(N, 3)
        P = qfa.solve_poisson(W).copy()
        X = _device_matmul(W, P)
        np.testing.assert_array_equal(qfa.commutator_skewherm(W, P), X - X.conj().T)
        G = np.random.default_rng(N).standard_normal((N, N)) + 1j * np.random.default_rng(N + 1).standard_normal((N, N))
        np.testing.assert_array_equal(qfa.commutator_generic(W, G), _device_matmul(W, G) - _device_matmul(G, W))
        C = qfa.commutator(W, P)
        assert np.array_equal(C, -C.conj().T) and np.all(C.diagonal().real == 0.0)
    with pytest.raises(ValueError):
        qfa.commutator(np.zeros((8, 8), complex), np.zeros((9, 9), complex))
    from quflow_amd.context import release_contexts
    release_contexts()


def test_host_results_stay_distinct_while_held(qfa, oracle):
    """laplace / commutator / products hand back a new ndarray per call as the reference's do; the allocation of a result the
    caller has DROPPED is reused (quflow_amd.context.result_array).  Held results are never overwritten, nested calls never
    alias their input."""
    N = 96
    W = oracle.make_W0(N, 5)
    P = qfa.solve_poisson(W).copy()
    A = qfa.laplace(P)
    A0 = A.copy()
    B = qfa.laplace(2.0 * P)
    assert B is not A and not np.shares_memory(A, B)
    np.testing.assert_array_equal(A, A0)
    np.testing.assert_array_equal(B, 2.0 * A0)
    L2 = qfa.laplace(qfa.laplace(P))                       # the inner result is the outer call's input
    np.testing.assert_array_equal(L2, qfa.laplace(A0.copy()))
    C1 = qfa.commutator(W, P)
    C1c = C1.copy()
    C2 = qfa.commutator(C1, P)                             # a held result as an operand
    assert C2 is not C1
    np.testing.assert_array_equal(C1, C1c)
    ids = set()
    for _ in range(5):
        ids.add(id(qfa.laplace(P)))                        # dropped at once: one allocation serves them all
    assert len(ids) <= 2
    np.testing.assert_array_equal(qfa.bracket(P, W), (qfa.geometry._device_matmul(P, W).copy() - qfa.geometry._device_matmul(W, P)) / qfa.hbar(N))


def test_generated_runfile_continues_the_record_on_the_device(qfa, tmp_path):
    """quflow_amd.create_runfile: the script it writes, run as its own process, re-opens the record and appends the stored
    number of steps with the device-resident stepper; the rows equal the same chunks through qfa.isomp on host arrays."""
    import os
    import subprocess
    import sys
    N = 48
    W0 = qfa.ensemble.make_W0(N, 9)
    rec = str(tmp_path / "run.qf")
    sim = qfa.Simulation(rec, overwrite=True, state=W0, loggers={'enstrophy': qfa.enstrophy})
    sim['stepsize'] = 0.2
    sim['steps'] = 6
    sim['steps_out'] = 3
    path = qfa.create_runfile(sim)
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, path], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "step 6" in r.stdout
    r2 = subprocess.run([sys.executable, path, "--steps", "3", "--tol", "1e-12"], capture_output=True, text=True, env=env, timeout=300)
    assert r2.returncode == 0, (r2.stdout + r2.stderr)[-2000:]
    back = qfa.Simulation(rec)
    np.testing.assert_array_equal(back['step'], [0, 3, 6, 9])
    dt = 0.2 * qfa.hbar(N)
    Wc = W0.copy()
    for row, kw in ((1, {}), (2, {}), (3, {"tol": 1e-12})):
        Wc = qfa.isomp(Wc, dt, steps=3, **kw)
        np.testing.assert_array_equal(back['mat', row], Wc)
    assert back['enstrophy', -1] == qfa.enstrophy(Wc)


# ----------------------------------------------------------------------------- guard zones around device allocations
def test_guard_zones_report_a_stray_store(qfa):
    """QUFLOW_HIP_DEBUG_GUARD (csrc/guard.hip): every device allocation of the library fenced by two 64 KiB pattern zones.
    In a process of its own: a solve, a stepper call on a size with guarded edge tiles and one with the stream-K exchange
    leave every zone intact, and the self-test's two stray bytes (one on either side of a scratch allocation) are found."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, ctypes, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import quflow_amd as qfa\n"
        "from quflow_amd import _lib\n"
        "lib = _lib.load()\n"
        "def report():\n"
        "    a, d = ctypes.c_longlong(0), ctypes.c_longlong(0)\n"
        "    t = ctypes.create_string_buffer(512)\n"
        "    assert lib.qf_debug_guard_check(ctypes.byref(a), ctypes.byref(d), t, 512) == 0\n"
        "    return a.value, d.value, t.value.decode()\n"
        "for N in (50, 1024):\n"
        "    W = qfa.ensemble.make_W0(N, 3)\n"
        "    qfa.solve_poisson(W)\n"
        "    qfa.isomp(W.copy(), 0.25 * qfa.hbar(N), steps=2)\n"
        "qfa.release_contexts()\n"
        "print(report())\n" % repo)
    out = {}
    for mode in ("1", "selftest"):
        env = dict(os.environ, QUFLOW_HIP_DEBUG_GUARD=mode)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
        out[mode] = eval(r.stdout.strip().splitlines()[-1])
    allocs, damaged, first = out["1"]
    assert allocs >= 20 and damaged == 0 and first == "", out["1"]
    allocs, damaged, first = out["selftest"]
    assert damaged == 2 and "allocation of 1000 bytes" in first, out["selftest"]
    # and without the variable the entry point answers zeros (nothing is fenced, nothing is slower)
    import ctypes
    from quflow_amd import _lib
    a, d = ctypes.c_longlong(7), ctypes.c_longlong(7)
    assert _lib.load().qf_debug_guard_check(ctypes.byref(a), ctypes.byref(d), None, 0) == 0
    if os.environ.get("QUFLOW_HIP_DEBUG_GUARD", "0") in ("", "0"):
        assert (a.value, d.value) == (0, 0)


def test_idle_contexts_make_room_on_a_full_device(qfa):
    """quflow_amd/context.py on the device: the Python layer caches one context per (device, N); when a new one does not fit
    (here: QUFLOW_HIP_DEBUG_GUARD_LIMIT_MB makes the device "full" at 700 MiB of library allocations, csrc/guard.hip) the
    library reports HIP's out-of-memory error with the half-built context cleaned up, the cached contexts nobody holds are
    closed and the creation succeeds on the second try -- a sweep over many sizes keeps running; results stay right; a held
    context survives.  In a process of its own (the variables are read once)."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import quflow_amd as qfa\n"
        "from quflow_amd import context\n"
        "held = qfa.PoissonHIP(384, np.complex128)\n"             # holds its context for the whole sweep
        "Wh = qfa.ensemble.make_W0(384, 1)\n"
        "Ph = held(Wh).copy()\n"
        "closed = 0\n"
        "orig = context.release_idle_contexts\n"
        "def counting():\n"
        "    global closed\n"
        "    n = orig(); closed += n; return n\n"
        "context.release_idle_contexts = counting\n"
        "for N in range(500, 524):\n"
        "    W = qfa.ensemble.make_W0(N, N)\n"
        "    P = qfa.solve_poisson(W).copy()\n"
        "    back = qfa.laplace(P)\n"
        "    assert np.abs(back - W).max() <= 1e-9 * np.abs(W).max(), N\n"
        "assert closed > 0, 'the limit was never reached'\n"
        "assert np.array_equal(held(Wh), Ph)\n"
        "print(closed, len(context._contexts), qfa.guard_report()[1])\n" % repo)
    env = dict(os.environ, QUFLOW_HIP_DEBUG_GUARD="1", QUFLOW_HIP_DEBUG_GUARD_LIMIT_MB="700")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    closed, cached, damaged = (int(x) for x in r.stdout.strip().splitlines()[-1].split())
    assert closed >= 1 and cached < 24 and damaged == 0, r.stdout
    # a context that cannot fit even on an empty device is still the error it was
    code2 = ("import sys\nsys.path.insert(0, %r)\nimport quflow_amd as qfa\n"
             "try:\n    qfa.solve_poisson(qfa.ensemble.make_W0(4096, 0))\nexcept qfa.QuflowHipError as e:\n    print('raised', 'out of memory' in str(e).lower())\n" % repo)
    r2 = subprocess.run([sys.executable, "-c", code2], env=env, capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0 and r2.stdout.strip().endswith("raised True"), (r2.stdout + r2.stderr)[-2000:]
