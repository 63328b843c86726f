"""Checkpoint / output layout and chunk driver (quflow_amd/simulation.py) against the reference's
contract: dataset names, shapes and dtypes of quflow/simulation.py:357-478, the callback protocol of
solve (:782-798), item access (:203-276), and the restart semantics of tests/test_simulation.py:130-168.
CPU: the stepper is the oracle's isomp (tests only); the device-resident path has its own GPU test."""
import numpy as np
import pytest

import quflow_amd as qfa
from quflow_amd.simulation import Simulation, solve, DirectoryStore


def norm_L2(W):
    return float(np.sqrt((np.abs(W) ** 2).sum() / W.shape[-1]))


def first_column(W):
    return W[:, 0]


@pytest.fixture(params=["dir", "h5", "h5double"])
def simfile(request, tmp_path, monkeypatch):
    if request.param == "h5":
        pytest.importorskip("h5py")
        return str(tmp_path / "testsim.hdf5")
    if request.param == "h5double":
        # H5Store's own logic against a stand-in for the few h5py calls it makes (tests/h5py_double.py): runs where
        # h5py cannot be installed; says nothing about the HDF5 format itself
        import sys
        import h5py_double
        monkeypatch.setitem(sys.modules, "h5py", h5py_double)
        return str(tmp_path / "testsim_double.hdf5")
    return str(tmp_path / "testsim.qf")


def test_layout_and_item_access(simfile, oracle):
    N = 12
    W = oracle.make_W0(N, 3)
    sim = Simulation(simfile, overwrite=True, state=W, loggers={'normL2': norm_L2, 'vector': first_column})
    Ws = [W]
    for i in range(1, 6):
        Ws.append(oracle.make_W0(N, 3 + i))
        sim(W=Ws[-1], delta_time=0.1, delta_steps=4, iterations=2.5, tol_auto=1e-8, number_of_maxit=0.0, unknown_field=1.0)
    fields = sim.fieldnames
    assert fields['mat'] == ((6, N, N), np.dtype(np.complex128))
    assert fields['time'] == ((6,), np.dtype(np.float64))
    assert fields['step'][0] == (6,) and fields['step'][1].kind == 'i'
    for name in ('tol_auto', 'iterations', 'number_of_maxit', 'normL2'):
        assert fields[name] == ((6,), np.dtype(np.float64)), name
    assert fields['vector'] == ((6, N), np.dtype(np.complex128))
    assert 'unknown_field' not in fields                       # only fields created at initialisation are appended to
    np.testing.assert_allclose(sim['time'], 0.1 * np.arange(6))
    np.testing.assert_array_equal(sim['step'], 4 * np.arange(6))
    np.testing.assert_array_equal(sim['mat', -1], Ws[-1])
    np.testing.assert_array_equal(sim[3], Ws[3])               # a bare index means the state
    np.testing.assert_array_equal(sim['mat', 2, 1], Ws[2][1])
    assert sim['normL2', -1] == norm_L2(Ws[-1])
    np.testing.assert_array_equal(sim['vector', 3], first_column(Ws[3]))
    assert sim['iterations', 0] == 0.0 and sim['iterations', -1] == 2.5
    assert int(sim['N']) == N and sim['qutypes'] == {'mat': None}
    # arguments: plain values and pickled callables (simulation.py:203-233)
    sim['dt'] = 0.05
    sim['steps_out'] = 7
    sim['hamiltonian'] = norm_L2
    sim['info'] = "a run"
    re = Simulation(simfile)
    assert float(re['dt']) == 0.05 and int(re['steps_out']) == 7 and re['hamiltonian'] is norm_L2
    assert re['info'] == "a run" and dict(re.args()).keys() >= {'dt', 'steps_out', 'hamiltonian'}
    np.testing.assert_array_equal(re['mat', -1], Ws[-1])
    with pytest.raises(KeyError):
        re['nothing']
    with pytest.raises(ValueError):
        Simulation(simfile, state=W)                           # already initialised
    with pytest.raises(NotImplementedError):
        Simulation(simfile, overwrite=True, state=W, qutypes={'fun': np.float32})


def test_solve_chunks_and_restart_bit_identical(simfile, tmp_path, oracle):
    """tests/test_simulation.py:113-168: time / step columns of a chunked solve, and 50 + 50 steps
    through a re-opened file equal 100 steps in one run, bit for bit."""
    N = 16
    W = oracle.make_W0(N, 9)
    kw = dict(stepsize=0.1, steps_out=10, integrator=oracle.isomp, hamiltonian=oracle.solve_poisson, resident=False)
    sim = Simulation(simfile, overwrite=True, state=W, loggers={'normL2': norm_L2})
    solve(W.copy(), steps=50, callback=sim, **kw)
    sim2 = Simulation(simfile)
    solve(sim2['mat', -1], steps=50, callback=sim2, **kw)
    dt = 0.1 * qfa.hbar(N)
    np.testing.assert_allclose(sim['time'], 10 * dt * np.arange(11))
    np.testing.assert_array_equal(sim['step'], 10 * np.arange(11))
    assert sim['normL2', -1] == norm_L2(sim['mat', -1])
    assert sim['iterations', -1] > 0 and sim['tol_auto', -1] > 0
    sim3 = Simulation(str(tmp_path / "straight.qf"), overwrite=True, state=W)
    solve(W.copy(), steps=100, callback=sim3, **kw)
    np.testing.assert_array_equal(sim3['mat', -1], sim['mat', -1])
    np.testing.assert_array_equal(sim3['mat'], sim['mat'])
    # continuing from the Simulation object itself: state, time and stored arguments come from the file
    sim['stepsize'] = 0.1
    sim['steps_out'] = 10
    solve(sim, steps=20, integrator=oracle.isomp, hamiltonian=oracle.solve_poisson, resident=False)
    assert sim['step', -1] == 120 and sim.fieldnames['mat'][0][0] == 13
    np.testing.assert_allclose(sim['time', -1], 120 * dt)
    Wc = oracle.isomp(sim3['mat', -1].copy(), dt, steps=10)
    Wc = oracle.isomp(Wc, dt, steps=10)
    np.testing.assert_array_equal(sim['mat', -1], Wc)


def test_directory_store_is_plain_files(tmp_path, oracle):
    path = tmp_path / "plain.qf"
    sim = Simulation(str(path), overwrite=True, state=oracle.make_W0(8, 1))
    sim(W=oracle.make_W0(8, 2), delta_time=1.0, delta_steps=1)
    assert DirectoryStore.exists(path)
    raw = np.fromfile(path / "mat.bin", dtype=np.complex128).reshape(2, 8, 8)
    np.testing.assert_array_equal(raw[1], oracle.make_W0(8, 2))
    with pytest.raises(ImportError):
        try:
            import h5py  # noqa: F401
        except ImportError:
            Simulation(str(tmp_path / "x.hdf5"), overwrite=True, state=oracle.make_W0(8, 1))
        else:
            raise ImportError("h5py present: nothing to check")


def test_solve_accepts_the_references_call_form(tmp_path, oracle):
    """tests/test_simulation.py:137 and run_TEMPLATE.py call solve(W, stepsize=.., steps=.., steps_out=..,
    progress_bar=False, callback=sim): the progress arguments (and the deprecated inner_steps / inner_time)
    are solve's own, never the integrator's, and the caller's array is advanced in place."""
    import io
    N = 12
    W = oracle.make_W0(N, 4)
    kw = dict(integrator=oracle.isomp, hamiltonian=oracle.solve_poisson, resident=False)
    sim = Simulation(str(tmp_path / "ref_form.qf"), overwrite=True, state=W)
    Wa = W.copy()
    out = solve(Wa, stepsize=0.1, steps=20, steps_out=10, progress_bar=False, callback=sim, **kw)
    np.testing.assert_array_equal(out, Wa)
    assert sim['step', -1] == 20 and sim.fieldnames['mat'][0][0] == 3
    Wb = W.copy()
    buf = io.StringIO()
    out_b = solve(Wb, stepsize=0.1, steps=20, inner_steps=10, progress_bar=True, progress_file=buf, **kw)
    np.testing.assert_array_equal(out_b, out)
    assert "steps" in buf.getvalue()                      # tqdm wrote its bar to the file it was given
    dt = 0.1 * qfa.hbar(N)
    out_c = solve(W.copy(), dt=dt, steps=20, inner_time=10 * dt, progress_bar=False, **kw)
    np.testing.assert_array_equal(out_c, out)


def test_directory_store_never_deletes_foreign_content(tmp_path, oracle):
    W = oracle.make_W0(8, 1)
    foreign = tmp_path / "results"
    foreign.mkdir()
    (foreign / "precious.txt").write_text("user data")
    for overwrite in (False, True):
        with pytest.raises(FileExistsError):
            Simulation(str(foreign), overwrite=overwrite, state=W)
        assert (foreign / "precious.txt").read_text() == "user data"
    afile = tmp_path / "a_file"
    afile.write_text("x")
    with pytest.raises(FileExistsError):
        Simulation(str(afile), overwrite=True, state=W)
    assert afile.read_text() == "x"
    empty = tmp_path / "empty_dir"
    empty.mkdir()
    sim = Simulation(str(empty), state=W)                 # an empty directory is fine
    assert DirectoryStore.exists(empty)
    sim(W=oracle.make_W0(8, 2), delta_time=1.0, delta_steps=1)
    sim2 = Simulation(str(empty), overwrite=True, state=W)  # a record may be replaced when asked to
    assert sim2.fieldnames['mat'][0][0] == 1


def test_create_runfile_writes_a_script_that_names_the_record(tmp_path, oracle):
    """quflow.simulation.create_runfile (simulation.py:484-585): a stand-alone script next to the record, the record's
    `prerun` pasted in, the reference's default name (`<record>_runfile.py`); compiles; says so when no device is there."""
    import subprocess
    import sys
    import os
    from quflow_amd.simulation import create_runfile
    rec = str(tmp_path / "myrun.qf")
    sim = Simulation(rec, overwrite=True, state=oracle.make_W0(8, 0))
    sim['stepsize'] = 0.1
    sim['steps'] = 4
    sim['steps_out'] = 2
    sim['prerun'] = "import numpy as np\nMY_CONSTANT = 3\n"
    path = create_runfile(sim)
    assert path == str(tmp_path / "myrun_runfile.py")
    src = open(path).read()
    compile(src, path, "exec")
    assert "MY_CONSTANT = 3" in src and "'myrun.qf'" in src and "qf.solve(mysim" in src
    assert create_runfile(rec, str(tmp_path / "other.py")) == str(tmp_path / "other.py")      # a record name instead of the object
    if not os.path.exists("/dev/kfd"):
        env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        r = subprocess.run([sys.executable, path], capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode != 0 and "no HIP device visible" in (r.stderr + r.stdout)
