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


def guard_report():
    """(allocations, damaged zones, first damage) of the library's guard zones (QUFLOW_HIP_DEBUG_GUARD=1; csrc/guard.hip)."""
    import quflow_amd
    return quflow_amd.guard_report()


@pytest.fixture(autouse=True)
def _guard_zones_after_every_test(request):
    """Under QUFLOW_HIP_DEBUG_GUARD=1 every device allocation of the library is fenced by two 64 KiB pattern zones: a test
    after which a zone is damaged FAILS (a kernel stored outside its operand), named here rather than at session end."""
    yield
    if os.environ.get("QUFLOW_HIP_DEBUG_GUARD", "0") not in ("", "0") and have_gpu() and request.node.get_closest_marker("gpu"):
        before = getattr(_guard_zones_after_every_test, "seen", 0)
        allocs, damaged, first = guard_report()
        _guard_zones_after_every_test.seen = damaged
        assert damaged == before, "guard zone damaged during this test: %s" % first


def pytest_terminal_summary(terminalreporter):
    if os.environ.get("QUFLOW_HIP_DEBUG_GUARD", "0") not in ("", "0") and have_gpu():
        try:
            allocs, damaged, first = guard_report()
        except Exception as exc:      # (no library: the build test says so)
            terminalreporter.write_line("guard zones: not read (%s)" % exc)
            return
        terminalreporter.write_line("guard zones (QUFLOW_HIP_DEBUG_GUARD): %d device allocations fenced, %d damaged zones%s"
                                    % (allocs, damaged, (": " + first) if damaged else ""))
