#!/opt/conda/bin/python3.9
"""Write the HDF5 interop fixtures of SURVEY.md section 8(f) row 4 by RUNNING THE REFERENCE's
`QuSimulation` + `solve` with the real h5py.

TEST INFRASTRUCTURE.  Runs only in the build container: the read-only reference is mounted at
/root/reference, and h5py (3.3.0) exists only under /opt/conda/bin/python3.9 there.  numba is not
importable in that interpreter either, so `oracle/refshim/numba` (identity decorators) stands in as
in oracle/gen_golden.py; h5py, numpy and scipy are the real packages (h5py is imported BEFORE the
shim directory goes on sys.path, so `refshim/h5py.py` never shadows it).

Run:   /opt/conda/bin/python3.9 oracle/gen_h5_fixture.py

Writes (data only -- no reference source travels):
  tests/golden/ref_qusim_n16.hdf5            a file exactly as quflow.QuSimulation leaves it after
                                             `solve(sim)`: N = 16, qutypes mat + shr, loggers energy /
                                             enstrophy, stats columns, 20 steps in two chunks of 10,
                                             arguments stepsize / steps / steps_out / hamiltonian /
                                             integrator (pickled), info, prerun
  tests/golden/ref_qusim_n16_continued.npz   every dataset of the same file after the REFERENCE
                                             re-opened it and ran 20 more steps (rows 3, 4) -- what a
                                             resume through quflow_amd must reproduce
  tests/golden/ref_qusim_n16_datapath.hdf5   the same first run under datapath "/run1/" with a (2,N,N)
                                             state stack and default loggers -- group handling
"""
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("QUFLOW_REFERENCE", "/root/reference")
GOLD = os.path.join(REPO, "tests", "golden")

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
os.environ["QUFLOW_SAVE_COMPUTED_BASIS"] = "0"

import h5py  # noqa: E402   the real one, first
assert hasattr(h5py, "File") and hasattr(h5py, "version"), "this script needs the real h5py"

sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import quflow as qf  # noqa: E402  (the reference)


def make_W0(N, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    W = A - A.conj().T
    W -= np.eye(N) * (np.trace(W) / N)
    W /= np.linalg.norm(W, "fro") / np.sqrt(N)
    return W


def first_run(path, W0, datapath="/", loggers=True):
    kw = dict(loggers={'energy': qf.energy_euler, 'enstrophy': qf.enstrophy}) if loggers else {}
    sim = qf.QuSimulation(path, overwrite=True, state=W0, qutypes={'mat': None, 'shr': None}, datapath=datapath, **kw)
    sim['stepsize'] = 0.25
    sim['steps'] = 20
    sim['steps_out'] = 10
    sim['hamiltonian'] = qf.solve_poisson
    sim['integrator'] = qf.isomp
    sim['info'] = "reference-written fixture (oracle/gen_h5_fixture.py)"
    sim['prerun'] = "import numpy as np\nimport quflow as qf\n"
    qf.solve(sim, progress_bar=False)
    return sim


def main():
    N = 16
    W0 = make_W0(N, 16)
    tmp = tempfile.mkdtemp(prefix="qf_h5_fixture_")
    try:
        path = os.path.join(tmp, "ref_qusim_n16.hdf5")
        first_run(path, W0)
        os.makedirs(GOLD, exist_ok=True)
        shutil.copyfile(path, os.path.join(GOLD, "ref_qusim_n16.hdf5"))

        # the reference's own resume (tests/test_simulation.py:130-168 pattern): re-open, 20 more steps
        sim2 = qf.QuSimulation(path)
        qf.solve(sim2, progress_bar=False)
        out = {}
        with h5py.File(path, "r") as f:
            for name in f["/"].keys():
                if isinstance(f[name], h5py.Dataset):
                    out[name] = f[name][:]
        out["W0"] = W0
        np.savez_compressed(os.path.join(GOLD, "ref_qusim_n16_continued.npz"), **out)

        # a data path other than "/" and a (k,N,N) stack (state 0 drives, isospectral.py:527-532)
        path3 = os.path.join(tmp, "ref_qusim_n16_datapath.hdf5")
        Wst = np.stack([W0, make_W0(N, 17)])
        first_run(path3, Wst, datapath="/run1/", loggers=False)
        shutil.copyfile(path3, os.path.join(GOLD, "ref_qusim_n16_datapath.hdf5"))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for fn in ("ref_qusim_n16.hdf5", "ref_qusim_n16_continued.npz", "ref_qusim_n16_datapath.hdf5"):
        p = os.path.join(GOLD, fn)
        print("wrote %-45s %7.1f KiB" % (os.path.relpath(p, REPO), os.path.getsize(p) / 1024))
    with h5py.File(os.path.join(GOLD, "ref_qusim_n16.hdf5"), "r") as f:
        print("datasets:", {k: (f[k].shape, str(f[k].dtype)) for k in f.keys() if isinstance(f[k], h5py.Dataset)})
        print("attrs   :", {k: type(v).__name__ for k, v in f["/"].attrs.items()})
        print("args    :", {k: type(v).__name__ for k, v in f["/args"].attrs.items()})


if __name__ == "__main__":
    main()
