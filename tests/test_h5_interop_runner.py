"""Runs the HDF5 interop tests where the real h5py is: /opt/conda/bin/python3.9 of this image (h5py 3.3.0, numpy 1.26;
the default interpreter has no h5py, so tests/test_h5_interop.py skips under it).  A child process per test here --
never an exec -- so the default `pytest tests/` of the driver exercises both directions of SURVEY.md section 8(f)
row 4 on CPU, and the device-resident resume of the reference-written file on the GPU box."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CANDIDATES = [os.environ.get("QUFLOW_H5PY_PYTHON"), sys.executable, "/opt/conda/bin/python3.9", "/opt/conda/bin/python"]


def interpreter_with_h5py():
    for exe in CANDIDATES:
        if not exe or not os.path.exists(exe):
            continue
        try:
            r = subprocess.run([exe, "-W", "ignore", "-c", "import h5py, numpy, pytest; print(h5py.version.version)"],
                               capture_output=True, text=True, timeout=120)
        except (OSError, subprocess.TimeoutExpired):
            continue
        if r.returncode == 0:
            return exe
    return None


@pytest.fixture(scope="module")
def h5python():
    exe = interpreter_with_h5py()
    if exe is None:
        pytest.skip("no interpreter with the real h5py on this machine (looked at: %s)" % [c for c in CANDIDATES if c])
    return exe


def test_h5_interop_both_directions(h5python):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([h5python, "-W", "ignore", "-m", "pytest", os.path.join(HERE, "test_h5_interop.py"), "-q", "-rs",
                        "-p", "no:cacheprovider"], cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
    if os.path.isdir("/root/reference/quflow"):
        # build container: nothing may have skipped (direction 2 needs the reference, which is here)
        assert "skipped" not in r.stdout, tail


@pytest.mark.gpu
def test_device_resume_of_reference_written_file(h5python, tmp_path):
    r = subprocess.run([h5python, "-W", "ignore", os.path.join(HERE, "h5_device_resume.py"), str(tmp_path)], cwd=REPO,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["ok"] and line["rows"] == 5
    assert line["max_abs_err_vs_reference_resume"]["mat"] <= 1e-13
