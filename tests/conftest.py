import os
import sys

# BLAS/OpenMP pools sized by the visible cpus (256 on the GPU host) exhaust a CPU-quota cgroup
# and get the whole process throttled; 16 = the GPU box's share (see bench.py)
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, str(min(16, os.cpu_count() or 1)))

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only; see oracle/isomp_oracle.py)."""
    from oracle import isomp_oracle
    isomp_oracle.build()
    return isomp_oracle


def have_gpu():
    """True when a HIP device is present (without initialising torch)."""
    return os.path.exists("/dev/kfd") and os.path.isdir("/dev/dri")
