"""Negative tests of the device-fault paths (GPU): a bounded device-side wait that runs out and a step end that
never comes must turn into an error return -- never a hang, never a silently wrong state -- and leave the
context usable.  The faults are injected by a debug switch of the library (QUFLOW_HIP_DEBUG_DROP_FLAG, honoured
only while QUFLOW_HIP_DEBUG is set, read when a context is created, spent on its first due second product):
    1 = no workgroup of one upper-triangle stream-K product publishes its piece flag (k_zgemm_tri: the heads'
        waits run out and raise qf_host_record::fault, which qf_isomp turns into QF_ERR_STATE; with the int8 products:
        no upper tile of k_oz_gemm publishes its result tile, the mirrored tiles' waits run out);
    2 = one epilogue of one second product takes no step-end ticket (the iteration never closes: the host's
        progress watchdog in qf_isomp fires).
Recovery is an error return only: nothing re-executes, the process and the context live on."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qfa():
    import quflow_amd
    if quflow_amd.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the MI355X box")
    return quflow_amd


def _faulty_trajectory(qfa, W0, mode):
    old = {k: os.environ.get(k) for k in ("QUFLOW_HIP_DEBUG", "QUFLOW_HIP_DEBUG_DROP_FLAG")}
    os.environ["QUFLOW_HIP_DEBUG"] = "1"
    os.environ["QUFLOW_HIP_DEBUG_DROP_FLAG"] = str(mode)
    try:
        return qfa.DeviceTrajectory(W0)          # the switch is read when the context is created
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("N,mode,needle,products", [
    (1024, 1, "device-side wait", "f64"),       # stream-K piece flag never published (k_zgemm_tri, N >= 960; under
                                                # QUFLOW_HIP_GEMM=auto the "f64" cases land in whatever auto selects)
    (1024, 2, "progress stuck", "f64"),         # a step-end ticket lost in the upper-triangle product
    (768, 2, "progress stuck", "f64"),          # ... in the 32x32 triangle product at a size above 512 (no deferral)
    (256, 2, "progress stuck", "f64"),          # ... and in the small-N second product
    (1024, 1, "device-side wait", "i8x65"),     # config 3's products: no upper tile of k_oz_gemm publishes its result tile,
                                                # the mirrored tiles' bounded waits (OZ_SPIN_LIMIT) run out
    (1024, 2, "progress stuck", "i8x65"),       # ... and a step-end ticket lost in the int8 second product
])
def test_injected_fault_is_an_error_and_the_context_survives(qfa, N, mode, needle, products, monkeypatch):
    if products != "f64":
        monkeypatch.setenv("QUFLOW_HIP_GEMM", products)
    W0 = qfa.ensemble.make_W0(N, 0)
    dt = 0.25 * qfa.hbar(N)
    ref = qfa.DeviceTrajectory(W0)
    st_ref = ref.advance(dt, 6)
    W_ref = ref.download()
    ref.ctx.close()

    tr = _faulty_trajectory(qfa, W0, mode)
    t0 = time.monotonic()
    with pytest.raises(qfa.QuflowHipError) as ei:
        tr.advance(dt, 6)
    assert time.monotonic() - t0 < 60.0                              # bounded: an error, not a hang
    assert needle in str(ei.value), str(ei.value)
    assert "QF_ERR_STATE" in str(ei.value)
    msg = tr.ctx._lib.qf_last_error().decode()
    assert needle in msg                                             # qf_last_error() names the wait
    # the switch is spent; the same context, re-uploaded, runs the same steps to the same bits
    tr.upload(W0)
    st = tr.advance(dt, 6)
    assert st["total_iterations"] == st_ref["total_iterations"]
    np.testing.assert_array_equal(tr.download(), W_ref)
    # ... and is cleanly destroyable
    tr.ctx.close()
    # the switch is ignored without QUFLOW_HIP_DEBUG
    os.environ["QUFLOW_HIP_DEBUG_DROP_FLAG"] = str(mode)
    try:
        os.environ.pop("QUFLOW_HIP_DEBUG", None)
        tr2 = qfa.DeviceTrajectory(W0)
    finally:
        os.environ.pop("QUFLOW_HIP_DEBUG_DROP_FLAG", None)
    tr2.advance(dt, 6)
    np.testing.assert_array_equal(tr2.download(), W_ref)
    tr2.ctx.close()
