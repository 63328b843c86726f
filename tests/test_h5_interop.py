"""HDF5 interop of the output / checkpoint layout with the REFERENCE's own files.

`tests/golden/ref_qusim_n16*.hdf5` were written by quflow.QuSimulation + quflow.solve with the real h5py
(oracle/gen_h5_fixture.py, run under /opt/conda/bin/python3.9 in the build container);
`ref_qusim_n16_continued.npz` holds every dataset of the same file after the reference itself resumed it.

  direction 1 (any interpreter with h5py): quflow_amd.Simulation opens the reference's file, reads every dataset,
              attribute and pickled argument, resumes and appends; the appended rows are what the reference's own
              resume produced, and bit-identical to the same stepper run straight from the stored row;
  direction 2 (needs the reference too: build container only): the reference's QuSimulation opens a file that
              quflow_amd.Simulation (H5Store) wrote, reads it, resumes and appends to it, and quflow_amd reads the result.

h5py lives only under /opt/conda/bin/python3.9 in this image: under the default interpreter this module skips, and
tests/test_h5_interop_runner.py runs it in that interpreter.  The stepper here is the oracle's (CPU checker); the
device-resident resume of the same file is tests/test_h5_interop_runner.py::test_device_resume_of_reference_written_file.
"""
import os
import shutil
import sys

import numpy as np
import pytest

h5py = pytest.importorskip("h5py", reason="h5py is not installed under this interpreter (see tests/test_h5_interop_runner.py)")
if not hasattr(h5py, "version"):
    pytest.skip("a stand-in h5py is on the path, not the real package", allow_module_level=True)

import quflow_amd as qfa  # noqa: E402
from quflow_amd.simulation import Simulation, solve  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = os.environ.get("QUFLOW_REFERENCE", "/root/reference")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 16


@pytest.fixture
def ref_file(tmp_path):
    dst = tmp_path / "ref_qusim_n16.hdf5"
    shutil.copyfile(os.path.join(GOLDEN, "ref_qusim_n16.hdf5"), dst)
    return str(dst)


@pytest.fixture
def host_transforms(monkeypatch):
    """'shr' rows on a box without a GPU: the oracle's mat2shr stands in for the device transform (host logic under
    test here is the file handling; the device transform has its own parity tests)."""
    from oracle import quantization_oracle as qo
    monkeypatch.setattr(qfa.quantization, "mat2shr", qo.mat2shr)
    return qo


def test_open_reference_written_file(ref_file, oracle, host_transforms):
    W0 = oracle.make_W0(N, 16)
    sim = Simulation(ref_file)
    fields = sim.fieldnames
    assert fields['mat'] == ((3, N, N), np.dtype(np.complex128))
    assert fields['shr'] == ((3, N * N), np.dtype(np.float64))
    assert fields['time'] == ((3,), np.dtype(np.float64))
    assert fields['step'][0] == (3,) and fields['step'][1].kind == 'i'
    for name in ('tol_auto', 'iterations', 'number_of_maxit', 'energy', 'enstrophy'):
        assert fields[name] == ((3,), np.dtype(np.float64)), name
    assert set(fields) == {'mat', 'shr', 'time', 'step', 'tol_auto', 'iterations', 'number_of_maxit', 'energy', 'enstrophy'}
    # datasets
    np.testing.assert_array_equal(sim['mat', 0], W0)
    np.testing.assert_array_equal(sim[0], W0)
    dt = 0.25 * oracle.hbar(N)
    np.testing.assert_allclose(sim['time'], 10 * dt * np.arange(3), rtol=1e-15)
    np.testing.assert_array_equal(sim['step'], [0, 10, 20])
    for r in range(3):
        Wr = sim['mat', r]
        assert abs(sim['energy', r] - oracle.energy_euler(Wr)) <= 1e-14
        assert abs(sim['enstrophy', r] - oracle.enstrophy(Wr)) <= 1e-14
        np.testing.assert_allclose(sim['shr', r], host_transforms.mat2shr(Wr), atol=1e-13)
    # the reference's own stepper advanced the rows: the oracle's gives the same states
    Wc = W0.copy()
    for r in (1, 2):
        st = {"iterations": 0.0}
        Wc = oracle.isomp(Wc, dt, steps=10, stats=st)
        np.testing.assert_allclose(sim['mat', r], Wc, atol=1e-14)
        assert sim['iterations', r] == st['iterations']
        np.testing.assert_allclose(sim['tol_auto', r], st['tol_auto'], rtol=1e-12)
        assert sim['number_of_maxit', r] == st['number_of_maxit']
    assert sim['iterations', 0] == 0.0
    # attributes of the data path
    assert int(sim['N']) == N
    assert sim['qutypes'] == {'mat': None, 'shr': None} and sim.qutypes == {'mat': None, 'shr': None}
    assert sim['version'] == "0.1.0"
    assert sim['info'].startswith("reference-written fixture")
    assert "import quflow as qf" in sim['prerun']
    assert len(sim['created']) >= 19
    # solve arguments; pickled `quflow.*` callables arrive as their counterparts in this package
    args = dict(sim.args())
    assert set(args) == {'stepsize', 'steps', 'steps_out', 'hamiltonian', 'integrator'}
    assert float(sim['stepsize']) == 0.25 and int(sim['steps']) == 20 and int(sim['steps_out']) == 10
    assert sim['hamiltonian'] is qfa.laplacian.solve_poisson
    assert sim['integrator'] is qfa.integrators.isomp_fixedpoint and sim['integrator'] is qfa.isomp
    assert sim.loggers == {'energy': qfa.physics.energy_euler, 'enstrophy': qfa.physics.enstrophy}
    with pytest.raises(KeyError):
        sim['nothing']
    with pytest.raises(ValueError):
        Simulation(ref_file, state=W0)


def test_resume_and_append_to_reference_written_file(ref_file, oracle, host_transforms):
    """20 more steps through quflow_amd.solve on the reference's file = the reference's own resume
    (ref_qusim_n16_continued.npz), appended in place; rows already in the file keep their bytes."""
    cont = np.load(os.path.join(GOLDEN, "ref_qusim_n16_continued.npz"))
    before = {}
    with h5py.File(ref_file, "r") as f:
        for name in f["/"].keys():
            if isinstance(f[name], h5py.Dataset):
                before[name] = f[name][:]
    sim = Simulation(ref_file, loggers={'energy': oracle.energy_euler, 'enstrophy': oracle.enstrophy})
    # stepsize / steps / steps_out come from the file; the stored callables are the device pair (previous test):
    # this CPU test hands the oracle's pair to solve instead
    out = solve(sim, integrator=oracle.isomp, hamiltonian=oracle.solve_poisson, resident=False, progress_bar=False)
    with h5py.File(ref_file, "r") as f:
        after = {name: f[name][:] for name in before}
        assert f['mat'].maxshape == (None, N, N) and f['mat'].chunks == (1, N, N)
    for name in before:
        assert after[name].shape[0] == 5, name
        np.testing.assert_array_equal(after[name][:3], before[name], err_msg=name)      # untouched rows
        assert after[name].shape == cont[name].shape and after[name].dtype == cont[name].dtype, name
    np.testing.assert_array_equal(after['step'], cont['step'])
    np.testing.assert_allclose(after['time'], cont['time'], rtol=1e-15)
    np.testing.assert_array_equal(after['iterations'], cont['iterations'])
    np.testing.assert_array_equal(after['number_of_maxit'], cont['number_of_maxit'])
    np.testing.assert_allclose(after['tol_auto'], cont['tol_auto'], rtol=1e-12)
    np.testing.assert_allclose(after['mat'], cont['mat'], atol=1e-14)
    np.testing.assert_allclose(after['shr'], cont['shr'], atol=1e-13)
    np.testing.assert_allclose(after['energy'], cont['energy'], atol=1e-14)
    np.testing.assert_allclose(after['enstrophy'], cont['enstrophy'], atol=1e-14)
    np.testing.assert_array_equal(out, after['mat'][-1])
    # bit-identity of the appended rows with the same stepper run straight from the stored row
    dt = 0.25 * oracle.hbar(N)
    Wc = before['mat'][2].copy()
    for r in (3, 4):
        Wc = oracle.isomp(Wc, dt, steps=10)
        np.testing.assert_array_equal(after['mat'][r], Wc)
    # ... and a third resume through a fresh object keeps going where the file ends
    sim2 = Simulation(ref_file, loggers={'energy': oracle.energy_euler, 'enstrophy': oracle.enstrophy})
    solve(sim2, steps=10, integrator=oracle.isomp, hamiltonian=oracle.solve_poisson, resident=False, progress_bar=False)
    assert sim2['step', -1] == 50 and sim2.fieldnames['mat'][0] == (6, N, N)
    np.testing.assert_array_equal(sim2['mat', -1], oracle.isomp(Wc.copy(), dt, steps=10))


def test_reference_file_with_datapath_and_state_stack(tmp_path, oracle, host_transforms):
    src = os.path.join(GOLDEN, "ref_qusim_n16_datapath.hdf5")
    path = str(tmp_path / "dp.hdf5")
    shutil.copyfile(src, path)
    sim = Simulation(path, datapath="/run1/")
    f = sim.fieldnames
    assert f['mat'] == ((3, 2, N, N), np.dtype(np.complex128)) and f['shr'] == ((3, 2, N * N), np.dtype(np.float64))
    assert set(f) == {'mat', 'shr', 'time', 'step', 'tol_auto', 'iterations', 'number_of_maxit'}
    np.testing.assert_array_equal(sim['mat', 0, 0], oracle.make_W0(N, 16))
    np.testing.assert_array_equal(sim['mat', 0, 1], oracle.make_W0(N, 17))
    assert sim.loggers == {} and int(sim['N']) == N
    # append one row by hand (the callback protocol), as solve would
    Wn = sim['mat', -1]
    sim(W=Wn, delta_time=0.5, delta_steps=3, iterations=2.0, tol_auto=1e-9, number_of_maxit=0.0)
    assert sim['step', -1] == 23 and sim.fieldnames['shr'][0] == (4, 2, N * N)
    np.testing.assert_array_equal(sim['mat', -1], Wn)


def test_default_qutypes_of_the_reference_are_refused_on_append_only(tmp_path, oracle):
    """A file written with the reference's default qutypes holds 'fun' / 'funL2' rows (simulation.py:47): reading is
    fine, appending names what is missing instead of writing a short record."""
    import pickle
    path = str(tmp_path / "defaults.hdf5")
    W = oracle.make_W0(8, 0)
    with h5py.File(path, "w") as f:
        f.create_group("/args/")
        f["/"].attrs["qutypes"] = np.array([pickle.dumps({'mat': None, 'fun': np.float32, 'funL2': np.float32})])
        f["/"].attrs["loggers"] = np.array([pickle.dumps({})])
        f["/"].attrs["N"] = 8
        d = f.create_dataset("/mat", (1, 8, 8), dtype=W.dtype, maxshape=(None, 8, 8), chunks=(1, 8, 8))
        d[0] = W
        for name, dt_ in (("fun", np.float32), ("funL2", np.float32)):
            f.create_dataset("/" + name, (1, 8, 15), dtype=dt_, maxshape=(None, 8, 15))
        f.create_dataset("/time", (1,), dtype=np.float64, maxshape=(None,))
        f.create_dataset("/step", (1,), dtype=int, maxshape=(None,))
    sim = Simulation(path)
    np.testing.assert_array_equal(sim['mat', -1], W)
    assert sim['fun'].shape == (1, 8, 15)
    with pytest.raises(NotImplementedError, match="fun"):
        sim(W=W, delta_time=0.1, delta_steps=1)
    assert sim.fieldnames['mat'][0] == (1, 8, 8)              # nothing was appended


# ------------------------------------------------------------------------------------------------
# direction 2: the reference reads and continues a file this package wrote (build container only)
# ------------------------------------------------------------------------------------------------

@pytest.fixture
def reference():
    if not os.path.isdir(os.path.join(REF, "quflow")):
        pytest.skip("the reference is not mounted here (build container only)")
    shim = os.path.join(REPO, "oracle", "refshim")
    saved = list(sys.path)
    os.environ["QUFLOW_SAVE_COMPUTED_BASIS"] = "0"
    sys.path.insert(0, shim)              # numba / appdirs / ducc0 stand-ins; the real h5py is imported already
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    try:
        import quflow as qf
        assert qf.simulation.h5py is h5py
        yield qf
    finally:
        sys.path[:] = saved


def test_reference_reads_and_continues_a_file_written_by_h5store(tmp_path, oracle, host_transforms, reference):
    qf = reference
    path = str(tmp_path / "ours.hdf5")
    W0 = oracle.make_W0(N, 21)
    dt = 0.25 * oracle.hbar(N)
    sim = Simulation(path, overwrite=True, state=W0, qutypes={'mat': None, 'shr': None},
                     loggers={'normL2': qfa.geometry.norm_L2, 'enstrophy': oracle.enstrophy})
    sim['stepsize'] = 0.25
    sim['steps'] = 20
    sim['steps_out'] = 10
    sim['info'] = "written by quflow_amd.Simulation"
    solve(sim, integrator=oracle.isomp, hamiltonian=oracle.solve_poisson, resident=False, progress_bar=False)
    ours = {name: sim[name] for name in sim.fieldnames}

    rsim = qf.QuSimulation(path)                      # the reference opens it (simulation.py:150-166)
    assert rsim.qutypes == {'mat': None, 'shr': None}
    assert set(rsim.loggers) == {'normL2', 'enstrophy'}
    assert set(rsim.fieldnames) == set(ours)
    for name in ours:
        np.testing.assert_array_equal(rsim[name], ours[name], err_msg=name)
        assert rsim.fieldnames[name] == (ours[name].shape, ours[name].dtype), name
    np.testing.assert_array_equal(rsim['mat', -1], ours['mat'][-1])
    assert int(rsim['N']) == N and rsim['info'] == "written by quflow_amd.Simulation"
    assert dict(rsim.args()).keys() == {'stepsize', 'steps', 'steps_out'}
    assert float(rsim['stepsize']) == 0.25 and int(rsim['steps_out']) == 10
    # the reference resumes and appends with its own stepper, solver and transforms (tests/test_simulation.py:130-168)
    qf.solve(rsim, progress_bar=False)
    back = Simulation(path)
    assert back.fieldnames['mat'][0] == (5, N, N) and back.fieldnames['shr'][0] == (5, N * N)
    np.testing.assert_array_equal(back['step'], [0, 10, 20, 30, 40])
    np.testing.assert_allclose(back['time'], 10 * dt * np.arange(5), rtol=1e-15)
    for name in ours:
        np.testing.assert_array_equal(back[name][:3], ours[name], err_msg=name)
    Wc = ours['mat'][-1].copy()
    for r in (3, 4):
        st = {"iterations": 0.0}
        Wc = oracle.isomp(Wc, dt, steps=10, stats=st)
        np.testing.assert_allclose(back['mat', r], Wc, atol=1e-14)
        assert back['iterations', r] == st['iterations']
        np.testing.assert_allclose(back['shr', r], host_transforms.mat2shr(Wc), atol=1e-13)
        assert abs(back['enstrophy', r] - oracle.enstrophy(Wc)) <= 1e-14
        assert abs(back['normL2', r] - qfa.geometry.norm_L2(back['mat', r])) <= 1e-15
    # a pickled callable of this package is readable by the reference as long as quflow_amd is importable there
    sim['hamiltonian'] = qfa.geometry.norm_L2
    assert qf.QuSimulation(path)['hamiltonian'] is qfa.geometry.norm_L2
