#!/opt/conda/bin/python3.9
"""Device-resident resume of a REFERENCE-written HDF5 file, run as a child process
by tests/test_h5_interop_runner.py under an interpreter that has the real h5py (only /opt/conda/bin/python3.9 in
this image; quflow_amd needs nothing but ctypes + numpy, so it drives the GPU from there as well).

    h5_device_resume.py <scratch directory>

Opens a copy of tests/golden/ref_qusim_n16.hdf5 (written by quflow.QuSimulation + quflow.solve,
oracle/gen_h5_fixture.py) with quflow_amd.Simulation and calls quflow_amd.solve(sim): stepsize / steps /
steps_out, the stepper and the Hamiltonian all come out of the file (the pickled quflow.isomp / quflow.solve_poisson
arrive as this package's device pair), the trajectory stays resident on the GPU, the loggers and the 'shr' rows are
device diagnostics / transforms.  Checks the appended rows against the reference's own resume
(ref_qusim_n16_continued.npz) and, bit for bit, against chunked host-array calls of the stepper.  Prints one JSON
line; exit code 0 = all checks held."""
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np  # noqa: E402
import h5py  # noqa: E402

import quflow_amd as qfa  # noqa: E402
from quflow_amd.simulation import Simulation, solve  # noqa: E402


def main():
    scratch = sys.argv[1]
    gold = os.path.join(HERE, "golden")
    path = os.path.join(scratch, "ref_qusim_n16_device.hdf5")
    shutil.copyfile(os.path.join(gold, "ref_qusim_n16.hdf5"), path)
    cont = np.load(os.path.join(gold, "ref_qusim_n16_continued.npz"))
    N = 16
    with h5py.File(path, "r") as f:
        before = {n: f[n][:] for n in f["/"].keys() if isinstance(f[n], h5py.Dataset)}
    sim = Simulation(path)
    assert sim['integrator'] is qfa.isomp and sim['hamiltonian'] is qfa.solve_poisson
    assert sim.loggers == {'energy': qfa.energy_euler, 'enstrophy': qfa.enstrophy}
    from quflow_amd import simulation as _s
    from quflow_amd import integrators as _i
    made = []
    real_traj = _i.DeviceTrajectory

    class Spy(real_traj):
        def __init__(self, *a, **k):
            made.append(1)
            super().__init__(*a, **k)
    _i.DeviceTrajectory = Spy
    try:
        out = solve(sim, progress_bar=False)
    finally:
        _i.DeviceTrajectory = real_traj
    assert made, "the resume did not take the device-resident route"
    with h5py.File(path, "r") as f:
        after = {n: f[n][:] for n in before}
    err = {}
    for n in before:
        assert after[n].shape == cont[n].shape and after[n].dtype == cont[n].dtype, n
        assert np.array_equal(after[n][:3], before[n]), n
    assert np.array_equal(after['step'], cont['step'])
    assert np.array_equal(after['iterations'], cont['iterations']), (after['iterations'], cont['iterations'])
    assert np.array_equal(after['number_of_maxit'], cont['number_of_maxit'])
    np.testing.assert_allclose(after['time'], cont['time'], rtol=1e-15)
    np.testing.assert_allclose(after['tol_auto'], cont['tol_auto'], rtol=1e-12)
    for n, tol in (('mat', 1e-13), ('shr', 1e-12), ('energy', 1e-13), ('enstrophy', 1e-13)):
        err[n] = float(np.abs(after[n] - cont[n]).max())
        assert err[n] <= tol, (n, err[n])
    assert np.array_equal(out, after['mat'][-1])
    dt = 0.25 * qfa.hbar(N)
    Wc = before['mat'][2].copy()
    for r in (3, 4):
        Wc = qfa.isomp(Wc, dt, steps=10)
        assert np.array_equal(after['mat'][r], Wc), "row %d differs from the chunked host-array call" % r
    print(json.dumps({"ok": True, "rows": int(after['mat'].shape[0]), "max_abs_err_vs_reference_resume": err,
                      "h5py": h5py.version.version, "python": sys.version.split()[0]}))


if __name__ == "__main__":
    main()
