"""Performance guard of the hot path (round 6; runs last in the GPU suite by its file name).

Not a benchmark: a tripwire.  It advances the headline workload (N = 1024, IC-A, dt = 0.25 hbar, adaptive) and config 2's
(N = 512) with HIP events around EVERY launch of the three kernels of an iteration (qf_profile_*, the instrument of
bench.py's instrumented_pass) and holds the mean duration per executed launch to the round's event-measured figures plus
slack, and qf_plan_describe to the kernels those figures belong to -- so a refactor of the host code, a toolchain bump or a
changed default cannot cost several percent unnoticed between two driver benches.

Budgets (us per executed launch; event figures of this protocol -- events around EVERY launch -- + ~8 %):
    N = 1024: first product <= 107, second product (k_zgemm_tri) <= 89, Laplacian inverse <= 17      (measured 97.9 / 76.8 / 15.3)
    N = 512 : first product <= 23, second product (k_zgemm_tri32) <= 26.5, Laplacian inverse <= 15.8  (measured on four boxes 20.7-21.3 /
              23.8-23.9 / 13.8-14.3: ~10 % -- a microsecond is 5 % at this size;
              the solve of the deferred protocol also takes the previous iteration's exit decision)
The best of three passes counts (a cold clock or a neighbour's burst on a shared host slows single passes; a regression
slows all three).  QUFLOW_PERF_GUARD=0 skips the timing asserts (plan checks stay) on hardware that is not an MI355X.
"""
import ctypes
import os

import pytest

pytestmark = pytest.mark.gpu

BUDGET_US = {
    1024: {"gemm1": 107.0, "gemm2": 89.0, "poisson": 17.0},
    512: {"gemm1": 23.0, "gemm2": 26.5, "poisson": 15.8},
}
KERNELS = {
    1024: {"first_product": "k_zgemm<64,64>", "second_product": "k_zgemm_tri", "laplacian_inverse": "k_solve<double, L=9, skew-Hermitian, folded walk slots>"},
    512: {"first_product": "k_zgemm<32,32>", "second_product": "k_zgemm_tri32<exact>", "laplacian_inverse": "k_solve<double, L=8, skew-Hermitian>"},
}


@pytest.fixture(scope="module")
def qfa():
    import quflow_amd
    if quflow_amd.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the MI355X box")
    return quflow_amd


def _pass(qfa, _lib, tr, dt, steps):
    lib, h = tr.ctx._lib, tr.ctx.handle
    _lib.check(lib.qf_profile_reset(h))
    _lib.check(lib.qf_profile_stride(h, 1))
    _lib.check(lib.qf_profile_enable(h, sum(1 << _lib.KERNEL_IDS[k] for k in ("poisson", "gemm1", "gemm2"))))
    st = tr.advance(dt, steps)
    tr.sync()
    _lib.check(lib.qf_profile_enable(h, 0))
    executed = max(int(st["total_iterations"]), 1)
    out = {}
    n, ms, seen = ctypes.c_longlong(), ctypes.c_double(), ctypes.c_longlong()
    for name in ("poisson", "gemm1", "gemm2"):
        _lib.check(lib.qf_profile_read(h, _lib.KERNEL_IDS[name], ctypes.byref(n), ctypes.byref(ms)))
        _lib.check(lib.qf_profile_seen(h, _lib.KERNEL_IDS[name], ctypes.byref(seen)))
        total_ms = ms.value * (seen.value / n.value if n.value else 0.0)
        out[name] = 1e3 * total_ms / executed          # us per executed launch
    return out, st


@pytest.mark.parametrize("N", [1024, 512])
def test_kernel_budgets_and_plan(qfa, N):
    from quflow_amd import _lib
    from quflow_amd.context import release_contexts
    for v in ("QUFLOW_HIP_GEMM", "QUFLOW_HIP_SOLVE_FOLD", "QUFLOW_HIP_DEFER"):
        assert v not in os.environ, "%s is set: the guard measures the defaults" % v
    release_contexts()
    W0 = qfa.ensemble.make_W0(N, 0)
    dt = 0.25 * qfa.hbar(N)
    tr = qfa.DeviceTrajectory(W0)
    tr.advance(dt, 300 if N >= 1024 else 1200)        # clocks up (~150 ms of work)
    tr.sync()
    passes = [_pass(qfa, _lib, tr, dt, 60 if N >= 1024 else 150)[0] for _ in range(3)]
    plan = tr.ctx.plan()
    tr.ctx.close()
    for role, kernel in KERNELS[N].items():
        got = plan[role]["kernel"]
        assert got == kernel, (role, got)
    best = {k: min(p[k] for p in passes) for k in passes[0]}
    print("perf guard N=%d: us per executed launch (best of 3) %s; passes %s" % (N, best, passes))
    if os.environ.get("QUFLOW_PERF_GUARD", "1") == "0":
        pytest.skip("QUFLOW_PERF_GUARD=0: timing asserts skipped (%s)" % best)
    for k, budget in BUDGET_US[N].items():
        assert best[k] <= budget, "N=%d %s: %.1f us per executed launch > budget %.1f (passes: %s)" % (N, k, best[k], budget, passes)


def _ensemble_rate(qfa, N, k, steps):
    import time
    dt = 0.25 * qfa.hbar(N)
    ens = qfa.DeviceEnsemble([qfa.ensemble.make_W0(N, s) for s in range(k)])
    t_end = time.perf_counter() + 0.15
    while time.perf_counter() < t_end:
        ens.advance(dt, 10)
    ens.advance(dt, 20)
    ens.sync()
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        ens.advance(dt, steps)
        ens.sync()
        best = max(best, k * steps / (time.perf_counter() - t0))
    ens.close()
    return best


def test_four_replicas_share_the_gpu(qfa):
    """four replicas per GPU at N = 512 reach 1.83 - 1.92 x one trajectory -- IF their streams' hardware queues
    sit on four different pipes (queue number mod 4; two replicas on one pipe: 1.1 - 1.2 x, profiles/r06_x4_hardware_queues.txt).
    The members of a DeviceEnsemble are created back to back and the library never uses the NULL stream, so their queues are
    consecutive whatever the process did before (this suite created hundreds of contexts before this test).  Floor 1.6 x."""
    from quflow_amd.context import release_contexts
    release_contexts()
    one = _ensemble_rate(qfa, 512, 1, 300)
    four = _ensemble_rate(qfa, 512, 4, 300)
    print("perf guard: N=512 one trajectory %.0f, four replicas %.0f timesteps/s (%.2f x)" % (one, four, four / one))
    if os.environ.get("QUFLOW_PERF_GUARD", "1") == "0":
        pytest.skip("QUFLOW_PERF_GUARD=0: timing asserts skipped (%.2f x)" % (four / one))
    assert four >= 1.6 * one, (one, four)
