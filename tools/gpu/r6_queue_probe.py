"""round 6 probe: why do four replicas at N = 512 drop from ~18,000 to ~11,000 timesteps/s under QUFLOW_HIP_DEBUG_GUARD (any zone
size)?  Hypothesis: the guard's hipMemset / hipMemcpy touch the NULL stream before the contexts create theirs, and the runtime's
hardware-queue assignment then puts replicas' streams on shared queues.  Variants: a NULL-stream memset before the library's first
context (no guard); guard with GPU_MAX_HW_QUEUES raised.  Usage: python tools/gpu/r6_queue_probe.py <variant>"""
import ctypes
import json
import os
import sys
import time

variant = sys.argv[1] if len(sys.argv) > 1 else "plain"
if variant.startswith("guard"):
    os.environ["QUFLOW_HIP_DEBUG_GUARD"] = "1"
if variant.endswith("q16"):
    os.environ["GPU_MAX_HW_QUEUES"] = "16"
if variant.endswith("q4"):
    os.environ["GPU_MAX_HW_QUEUES"] = "4"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(v, "8")
import quflow_amd as qfa  # noqa: E402

if variant.startswith("nullmemset"):
    hip = ctypes.CDLL("libamdhip64.so")
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 20)) == 0
    assert hip.hipMemset(p, 0, ctypes.c_size_t(1 << 20)) == 0
    assert hip.hipDeviceSynchronize() == 0
    if variant == "nullmemset_free":
        hip.hipFree(p)


def rate(N, k, steps, warmup=20):
    dt = 0.25 * qfa.hbar(N)
    ens = qfa.DeviceEnsemble([qfa.ensemble.make_W0(N, s) for s in range(k)])
    t_end = time.perf_counter() + 0.15
    while time.perf_counter() < t_end:
        ens.advance(dt, 10)
    ens.advance(dt, warmup)
    ens.sync()
    t0 = time.perf_counter()
    ens.advance(dt, steps)
    ens.sync()
    el = time.perf_counter() - t0
    ens.close()
    return k * steps / el


out = {"variant": variant, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
for k in (4, 1, 4, 2):
    out.setdefault("x%d" % k, []).append(round(rate(512, k, 300)))
print(json.dumps(out), flush=True)
