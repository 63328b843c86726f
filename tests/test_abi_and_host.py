"""CPU-only checks: the C-ABI library loads and exports every symbol that
include/quflow_hip.h declares, the host-side wrappers validate arguments, and the
product path fails loudly (no CPU fallback) when no GPU is present."""
import os
import re
import sys

import numpy as np
import pytest

from conftest import REPO, have_gpu


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from quflow_amd import _lib
    return _lib


def header_symbols():
    text = open(os.path.join(REPO, "include", "quflow_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qf_[a-z_A-Z0-9]+)\s*\(", text)))


def test_header_symbols_exported(built):
    lib = built.load()
    names = header_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), "libquflow_hip.so does not export %s" % name
    # and the Python binding declares a signature for each of them
    assert sorted(built.SIGNATURES) == names


def test_version_and_hbar(built):
    lib = built.load()
    assert lib.qf_version() == 100
    for N in (2, 64, 1024):
        assert lib.qf_hbar(N) == 2.0 / np.sqrt(N ** 2 - 1)


def test_no_silent_cpu_fallback(built):
    """Without a HIP device every compute entry point must raise, not fall back."""
    if have_gpu():
        pytest.skip("a GPU is present")
    import quflow_amd as qfa
    assert qfa.device_count() == 0
    W = np.zeros((8, 8), dtype=complex)
    with pytest.raises(qfa.QuflowHipError, match="NO_DEVICE"):
        qfa.solve_poisson(W)
    with pytest.raises(qfa.QuflowHipError, match="NO_DEVICE"):
        qfa.isomp(W, 0.1, steps=1)
    with pytest.raises(qfa.QuflowHipError, match="NO_DEVICE"):
        qfa.energy_euler(W)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under quflow_amd/ may reference it."""
    pkg = os.path.join(REPO, "quflow_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("test oracle", ""), os.path.join(root, f)


def test_argument_validation_before_device(built):
    import quflow_amd as qfa
    W = np.zeros((8, 8), dtype=complex)
    with pytest.raises(AssertionError):
        qfa.isomp(W, 0.1, steps=1, minit=0)
    with pytest.raises(AssertionError):
        qfa.isomp(W, 0.1, steps=1, minit=3, maxit=2)
    # the host hooks (also on (k,N,N) stacks, also with compsum) run around a device-resident
    # trajectory: there is no CPU fallback, without a device they fail loudly
    stack = np.zeros((2, 8, 8), dtype=complex)
    if qfa.device_count() < 1:
        for kw in (dict(forcing=lambda P, W: W), dict(hamiltonian=lambda W: W[0]),
                   dict(strang_splitting=lambda h, W: W), dict(compsum=True)):
            with pytest.raises(qfa.QuflowHipError):
                qfa.isomp(stack.copy(), 0.1, steps=1, **kw)
        with pytest.raises(qfa.QuflowHipError):
            qfa.isomp(W.copy(), 0.1, steps=1, forcing=lambda P, W: W)
    with pytest.raises(ValueError):
        qfa.isomp(np.zeros((4, 8), dtype=complex), 0.1, steps=1)
    # the other steppers validate before touching the device, too
    with pytest.raises(NotImplementedError):          # hooks on a complex64 stack: refused before any device call
        qfa.rk4(np.zeros((2, 8, 8), dtype=np.complex64), 0.1, 1, forcing=lambda P, W: W)
    if qfa.device_count() < 1:
        # (a foreign Hamiltonian runs since round 3 -- around a device-resident state: loud without a device)
        with pytest.raises(qfa.QuflowHipError):
            qfa.isomp_quasinewton(W, 0.1, 1, hamiltonian=lambda W: W)
        with pytest.raises(qfa.QuflowHipError):
            qfa.rk4(np.zeros((2, 8, 8), dtype=complex), 0.1, 1)
        with pytest.raises(qfa.QuflowHipError):
            qfa.isomp(W.astype(np.complex64), 0.1, steps=1)
        with pytest.raises(qfa.QuflowHipError):
            qfa.solve_poisson(W.astype(np.complex64))
    with pytest.raises(AssertionError):
        qfa.magmp(np.zeros((2, 8, 8), dtype=complex), 0.1, 1, minit=0)
    with pytest.raises(ValueError):
        qfa.magmp(W, 0.1, 1)
    # without a device every compute entry point fails loudly (no CPU fallback)
    if qfa.device_count() < 1:
        with pytest.raises(qfa.QuflowHipError):
            qfa.isomp(np.zeros((2, 8, 8), dtype=complex), 0.1, steps=1)
        with pytest.raises(qfa.QuflowHipError):
            qfa.shr2mat(np.zeros(64), N=8)


def test_stepper_signature_matches_reference():
    """simulation.solve discovers `stats` via inspect.getfullargspec (simulation.py:730)."""
    import inspect
    import quflow_amd as qfa
    args = inspect.getfullargspec(qfa.isomp).args
    assert args[:15] == ["W", "dt", "steps", "hamiltonian", "time", "forcing", "strang_splitting", "stats",
                         "callback", "tol", "maxit", "minit", "verbatim", "compsum", "reinitialize"]
    spec = inspect.getfullargspec(qfa.isomp)
    defaults = dict(zip(spec.args[-len(spec.defaults):], spec.defaults))
    assert defaults["steps"] == 100 and defaults["tol"] == "auto" and defaults["maxit"] == 10
    assert defaults["minit"] == 1 and defaults["compsum"] is False and defaults["reinitialize"] is False
    assert 'stats' in inspect.getfullargspec(qfa.IsompHIP.__call__).args


def test_native_hamiltonian_recognition():
    from quflow_amd import integrators, laplacian

    def solve_poisson(W):
        return W
    solve_poisson.__module__ = "quflow.laplacian.cpu"   # what simulation.solve injects (simulation.py:729)
    assert integrators._is_native_hamiltonian(solve_poisson)
    assert integrators._is_native_hamiltonian(laplacian.solve_poisson)
    assert integrators._is_native_hamiltonian(None)
    assert not integrators._is_native_hamiltonian(lambda W: W)


def test_make_W0_and_shard():
    from quflow_amd import ensemble
    from oracle import isomp_oracle
    np.testing.assert_array_equal(ensemble.make_W0(32, 3), isomp_oracle.make_W0(32, 3))
    assert ensemble.shard(range(8), 1, 2) == [1, 3, 5, 7]
    assert ensemble.shard(range(3), 2, 4) == [2]
    assert ensemble.shard(range(3), 3, 4) == []
    parts = [ensemble.shard(range(11), r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == list(range(11))


@pytest.mark.gpu
def test_yielding_host_poll_gives_the_same_trajectory(tmp_path):
    """QUFLOW_HIP_POLL=yield (hosts where ranks outnumber cores) changes how the host waits, not what the
    device computes: the state after 6 steps is bit-identical to the default polling mode."""
    import subprocess
    import numpy as np
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import quflow_amd as qfa\n"
        "W = qfa.ensemble.make_W0(128, 7)\n"
        "tr = qfa.DeviceTrajectory(W)\n"
        "st = tr.advance(0.25 * qfa.hbar(128), 6)\n"
        "np.save(sys.argv[1], tr.download())\n"
        "print(st['iterations'])\n" % REPO)
    outs = []
    for mode in ("", "yield"):
        env = dict(os.environ, QUFLOW_HIP_POLL=mode)
        path = str(tmp_path / ("w_%s.npy" % (mode or "pause")))
        r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append((np.load(path), r.stdout.strip()))
    assert outs[0][1] == outs[1][1]
    np.testing.assert_array_equal(outs[0][0], outs[1][0])


def _run_py(code, **env_extra):
    import subprocess
    env = dict(os.environ, **env_extra)
    return subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n%s" % (REPO, code)], env=env,
                          capture_output=True, text=True, timeout=300)


def test_library_path_switch(built, tmp_path):
    """QUFLOW_HIP_LIB: another build of the same library is loaded instead of the in-tree one (A/B runs); a path that
    does not exist is an error that names it -- never a silent fall-back to another library."""
    import shutil
    other = tmp_path / "libquflow_hip_other.so"
    shutil.copy(os.path.join(REPO, "quflow_amd", "libquflow_hip.so"), other)
    code = "from quflow_amd import _lib\nlib = _lib.load()\nprint(_lib.LIB_PATH, lib.qf_version())"
    res = _run_py(code, QUFLOW_HIP_LIB=str(other))
    assert res.returncode == 0, res.stderr[-2000:]
    assert res.stdout.split() == [str(other), "100"]
    res = _run_py(code, QUFLOW_HIP_LIB=str(tmp_path / "missing.so"))
    assert res.returncode != 0 and "missing.so" in res.stderr and "no CPU fallback" in res.stderr


def test_device_switch():
    """QUFLOW_HIP_DEVICE: the default device of get_context(); one process per GPU falls back to LOCAL_RANK."""
    code = "from quflow_amd import context\nprint(context.default_device())"
    env = {k: v for k, v in os.environ.items() if k not in ("QUFLOW_HIP_DEVICE", "LOCAL_RANK")}
    import subprocess

    def run(**extra):
        r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n%s" % (REPO, code)],
                           env=dict(env, **extra), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return int(r.stdout.strip())
    assert run() == 0
    assert run(LOCAL_RANK="5") == 5
    assert run(LOCAL_RANK="5", QUFLOW_HIP_DEVICE="3") == 3


def test_rccl_library_switch(built):
    """QUFLOW_HIP_RCCL_LIB: the library the torch-free gather opens first.  Pointed at a library that is not RCCL, the
    first communicator call fails with the missing symbol's name instead of opening the system's librccl."""
    import ctypes.util
    libm = ctypes.util.find_library("m")
    assert libm
    code = ("import ctypes\nfrom quflow_amd import _lib\nlib = _lib.load()\nbuf = ctypes.create_string_buffer(128)\n"
            "rc = lib.qf_comm_unique_id(buf)\nprint(rc, lib.qf_last_error().decode())")
    res = _run_py(code, QUFLOW_HIP_RCCL_LIB=libm)
    assert res.returncode == 0, res.stderr[-2000:]
    rc, msg = res.stdout.strip().split(" ", 1)
    assert int(rc) != 0 and "ncclGetUniqueId missing" in msg


def test_reference_argument_rules_hold_before_any_device_call():
    """What the reference's steppers do with their state and step count before anything else (isospectral.py:463, 481-482, 592;
    erk.py, mhd.py): `range(steps)` refuses a float, the in-place complex updates refuse a real or integer array.  These
    answers come from the host mirror alone -- no library, no GPU needed."""
    import numpy as np
    import quflow_amd as qfa
    W = np.zeros((8, 8), dtype=np.complex128)
    for stepper in (qfa.isomp, qfa.rk4, qfa.heun, qfa.euler, qfa.isomp_simple, qfa.isomp_quasinewton):
        with pytest.raises(TypeError):
            stepper(W.copy(), 0.1, 2.0)
        with pytest.raises(TypeError):
            stepper(np.zeros((8, 8)), 0.1, 2)
        with pytest.raises(TypeError):
            stepper(np.zeros((8, 8), dtype=np.int64), 0.1, 2)
        with pytest.raises(TypeError):
            stepper([[0j, 1j], [1j, 0j]], 0.1, 2)
        # an empty loop (steps <= 0) attempts no in-place update: a real or integer W comes back as it is, before any device call
        for steps in (0, -3):
            Wr = np.arange(64, dtype=np.float64).reshape(8, 8)
            assert stepper(Wr, 0.1, steps) is Wr and Wr[1, 1] == 9.0
    with pytest.raises(TypeError):
        qfa.magmp(np.zeros((2, 8, 8)), 0.1, 2)
    with pytest.raises(AssertionError):
        qfa.isomp(W.copy(), 0.1, 2, minit=0)
    with pytest.raises(TypeError):
        qfa.isomp(W.copy(), 0.1, 2, tol="loose", strang_splitting=lambda h, X: X)


def test_laplacian_module_exports_the_references_names():
    """`from .cpu import *` of quflow/laplacian/__init__.py:1 puts these public names on the backend module
    (cpu.py:563-943); select_first / select_sum are host reductions and run without a device."""
    import quflow_amd as qfa
    from conftest import load_golden
    for name in ("laplacian", "laplace", "solve_poisson", "select_skewherm", "select_first", "select_sum",
                 "allocate_buffer", "solve_heat", "solve_helmholtz", "solve_viscdamp", "solve_globalqg"):
        assert callable(getattr(qfa.laplacian, name)), name
    g = load_golden("reduce")
    for N in (17, 33):
        S = g["N%d_S" % N]
        S4 = np.stack([S, 2.0 * S])
        first = qfa.laplacian.select_first(S)
        np.testing.assert_array_equal(first, g["N%d_first" % N])
        assert first.flags.c_contiguous
        np.testing.assert_array_equal(qfa.laplacian.select_sum(S), g["N%d_sum" % N])
        np.testing.assert_array_equal(qfa.laplacian.select_first(S4), g["N%d_first4" % N])
        np.testing.assert_array_equal(qfa.laplacian.select_sum(S4), g["N%d_sum4" % N])
    import inspect
    assert inspect.signature(qfa.laplacian.solve_poisson).parameters["reduce"].default is qfa.laplacian.select_first


def test_result_arrays_are_new_unless_the_last_one_was_dropped():
    """quflow_amd.context.result_array: results the reference returns as a NEW ndarray per call (laplace, the commutators) keep
    that meaning -- a caller that still holds the previous result (or a view of it) gets a distinct array -- while a dropped
    result's allocation is reused (no 16 N^2-byte mmap / page-fault / munmap cycle per call)."""
    from quflow_amd.context import result_array, _result_cache
    _result_cache.clear()
    a = result_array((4, 4), np.complex128, "t")
    a[...] = 1.0
    b = result_array((4, 4), np.complex128, "t")
    assert b is not a and not np.shares_memory(a, b)            # a is still held
    ida = id(b)
    del b
    c = result_array((4, 4), np.complex128, "t")
    assert id(c) == ida                                         # the dropped one is reused
    v = c[1]                                                    # a view keeps its base alive and counted
    del c
    d = result_array((4, 4), np.complex128, "t")
    assert not np.shares_memory(d, v)
    e = result_array((4, 4), np.complex64, "t")
    f = result_array((4, 4), np.complex128, "other")
    assert e.dtype == np.complex64 and not np.shares_memory(f, d)
    assert np.all(a == 1.0)
    _result_cache.clear()


def test_top_level_names_of_the_reference_around_the_hot_path():
    """`import quflow as qf` scripts use these names next to the stepper (quflow/__init__.py: simulation, utils, quantization,
    geometry, physics re-exports); the ones that belong to the hot path or stand next to it exist here under the same names."""
    import quflow_amd as qfa
    for name in ("isomp", "isomp_fixedpoint", "isomp_simple", "isomp_quasinewton", "magmp", "euler", "heun", "rk4", "solve_poisson",
                 "laplace", "laplacian", "solve", "QuSimulation", "hbar", "qtime2seconds", "seconds2qtime", "inner_L2", "norm_L2",
                 "norm_Linf", "norm_L1", "energy_euler", "enstrophy", "inner_H1", "inner_Hm1", "shr2mat", "mat2shr", "shc2mat",
                 "mat2shc", "elm2ind", "ind2elm", "berezin_multipliers", "commutator", "integrators", "geometry", "physics"):
        assert hasattr(qfa, name), name
    assert qfa.QuSimulation is qfa.Simulation
    for N in (2, 64, 1024):
        assert qfa.qtime2seconds(3.0, N) == 3.0 * qfa.hbar(N) and qfa.seconds2qtime(qfa.qtime2seconds(3.0, N), N) == pytest.approx(3.0, rel=1e-15)
    assert [qfa.ind2elm(qfa.elm2ind(el, m)) for el, m in ((0, 0), (3, -2), (5, 5))] == [(0, 0), (3, -2), (5, 5)]


def test_idle_contexts_make_room_when_a_new_one_does_not_fit(monkeypatch):
    """quflow_amd/context.py: contexts are cached per (device, N) for the life of the process; when a new one fails with the
    device's out-of-memory error, the cached contexts nobody else holds are closed and the creation is tried once more.  A
    context somebody holds (a PoissonHIP / IsompHIP object, a call in flight) is never closed; any other error propagates."""
    from quflow_amd import context, _lib

    live, closed = [0], []            # (a count, not the objects: a list of them would be a holder)

    class FakeContext:
        capacity = 3

        def __init__(self, N, device=None):
            if N < 0:
                raise _lib.QuflowHipError("QF_ERR_ARG: qf_ctx_create: bad N")
            if live[0] >= FakeContext.capacity:
                raise _lib.QuflowHipError("QF_ERR_HIP: hipMalloc((void **)m, mbytes) failed: out of memory (api_context.hip:165)")
            self.N, self.device, self.handle = N, device, object()
            live[0] += 1

        def close(self):
            if self.handle is not None:
                self.handle = None
                live[0] -= 1
                closed.append(self.N)

    monkeypatch.setattr(context, "Context", FakeContext)
    monkeypatch.setattr(context, "_contexts", {})
    monkeypatch.setattr(context, "_stepper_contexts", {})
    monkeypatch.setattr(context, "_default_device", 0)
    a = context.get_context(8)
    assert context.get_context(8) is a                      # cached
    context.get_context(16)
    context.get_stepper_context(16)
    assert live[0] == 3 and not closed
    c = context.get_context(32)                             # does not fit: 16 and the stepper's 16 go, 8 is held by `a`
    assert sorted(closed) == [16, 16] and c.N == 32 and a.handle is not None
    assert sorted(k[1] for k in context._contexts) == [8, 32] and not context._stepper_contexts
    held = [context.get_context(64)]                        # fits (2 live + 1)
    with pytest.raises(_lib.QuflowHipError, match="out of memory"):
        held.append(context.get_context(128))               # everything cached is held: nothing to close, the error stands
    del c
    with pytest.raises(_lib.QuflowHipError, match="bad N"):
        context.get_context(-1)                             # another error: nothing is closed for it
    assert sorted(closed) == [16, 16]
    assert context.get_context(128).N == 128 and 32 in closed    # 32 is idle now and makes room


def kernel_resources():
    """{demangled kernel name: {vgpr, agpr, scratch, occupancy, lds, dynamic_stack}} from the <unit>.res records the
    Makefile keeps of every compile (-Rpass-analysis=kernel-resource-usage)."""
    import glob
    import subprocess
    rows, cur = [], None
    for path in sorted(glob.glob(os.path.join(REPO, "quflow_amd", "csrc", "*.res"))):
        for line in open(path, errors="replace"):
            m = re.search(r"remark:\s+(.*?)\s+\[-Rpass-analysis", line)
            if not m:
                continue
            key, _, val = m.group(1).partition(": ")
            if key == "Function Name":
                cur = {"mangled": val.strip(), "unit": os.path.basename(path)}
                rows.append(cur)
            elif cur is not None:
                cur[key.strip()] = val.strip()
    names = subprocess.run(["c++filt"] + [r["mangled"] for r in rows], capture_output=True, text=True).stdout.splitlines() if rows else []
    out = {}
    for r, name in zip(rows, names):
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*", "", name)
        out[name] = {"unit": r["unit"], "vgpr": int(r["VGPRs"]), "agpr": int(r["AGPRs"]), "scratch": int(r["ScratchSize [bytes/lane]"]),
                     "occupancy": int(r["Occupancy [waves/SIMD]"]), "lds": int(r["LDS Size [bytes/block]"]),
                     "dynamic_stack": r["Dynamic Stack"] != "False"}
    return out


def test_kernel_resources(built):
    """What the code generator made of every kernel of the library, read from the build's own records: NO kernel uses
    scratch memory or a dynamic stack (a kernel with scratch costs ~0.2 ms of host time per launch on this runtime and its
    spills sit in the hot loops: DESIGN.md 3.1b's codegen trap), and the kernels of the hot path keep the occupancy their
    measured times were taken at -- a toolchain bump or an innocent edit that costs a wave per SIMD
    fails here, on the CPU, before any GPU run."""
    res = kernel_resources()
    assert len(res) >= 100, len(res)                 # every unit with kernels left its record
    bad = {k: v for k, v in res.items() if v["scratch"] != 0 or v["dynamic_stack"]}
    assert not bad, bad
    floors = {
        "k_zgemm<64, 64, 2, 2, false, true, false>": 1,      # first product, N % 64 == 0 and N >= 896: one workgroup owns its CU
        "k_zgemm_tri": 1,                                    # stream-K upper triangle
        "k_zgemm<32, 32, 2, 2, false, true, false>": 3,      # first product below
        "k_zgemm_tri32<true>": 2,                            # 32 x 32 upper triangle (two per CU: four replicas per GPU rest on it)
        "k_zgemm_tri32<false>": 2,
        "k_solve<double, 4, 1, 0>": 5,                       # N <= 256
        "k_solve<double, 8, 1, 0>": 3,                       # N <= 512
        "k_solve<double, 9, 1, 1>": 3,                       # folded walk slots, 768 <= N < ~1100
        "k_solve<double, 17, 1, 1>": 2,                      # ... up to 2175
        "k_solve<double, 32, 1, 0>": 1,                      # above
        "k_solve<float, 8, 1, 0>": 4,
        "k_decide": 6,
    }
    for name, floor in floors.items():
        assert name in res, (name, sorted(k for k in res if k.startswith(name.split("<")[0]))[:12])
        assert res[name]["occupancy"] >= floor, (name, res[name])
    # the 64 x 64 fp64 product kernels live in the whole register file of their SIMD (512 registers, accumulators in VGPR form)
    assert res["k_zgemm_tri"]["vgpr"] + res["k_zgemm_tri"]["agpr"] <= 512


def test_library_never_names_the_null_stream():
    """Every copy, fill and launch of the library goes to a stream it created: the NULL stream's hardware queue appears when it
    is first used and moves every stream created after it to another pipe (queue number mod 4) -- two replicas of an ensemble on
    one pipe lose 40 % of their combined rate (profiles/r06_x4_hardware_queues.txt).  Source scan: no synchronous
    hipMemcpy / hipMemset / hipMemcpyDtoH..., no launch on stream 0."""
    src_dir = os.path.join(REPO, "quflow_amd", "csrc")
    for f in sorted(os.listdir(src_dir)):
        if not f.endswith((".hip", ".h")):
            continue
        text = open(os.path.join(src_dir, f)).read()
        text = re.sub(r"//[^\n]*", "", text)
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for pat in (r"\bhipMemcpy\s*\(", r"\bhipMemset\s*\(", r"\bhipMemcpy(DtoH|HtoD|DtoD)\s*\(", r"\bhipMemcpy2D\s*\(",
                    r"hipLaunchKernelGGL\([^;]*,\s*0\s*,\s*0\s*,", r"<<<"):
            m = re.search(pat, text)
            assert not m, (f, m.group(0))
