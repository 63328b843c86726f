"""round 6 probe 3: which hardware queues four replicas sit on decides their throughput?  Members of a DeviceEnsemble are created
one at a time; after member i, d dummy streams are created and used once (a 4 KiB fill each), which takes d hardware queue ids
out of the sequence.  Usage: python tools/gpu/r6_queue_probe3.py i:d[,i:d...]   e.g. 0:1 (ids 1,3,4,5), 0:4 (ids 1,6,7,8), none"""
import ctypes
import json
import os
import sys
import time

spec = sys.argv[1] if len(sys.argv) > 1 else "none"
plan = {}
if spec != "none":
    for part in spec.split(","):
        i, d = part.split(":")
        plan[int(i)] = int(d)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(v, "8")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import quflow_amd as qfa  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
scratch = ctypes.c_void_p()
dummies = []


def dummy_stream():
    if not scratch.value:
        assert hip.hipMalloc(ctypes.byref(scratch), ctypes.c_size_t(1 << 16)) == 0
    st = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(st), ctypes.c_uint(1)) == 0      # hipStreamNonBlocking
    assert hip.hipMemsetAsync(scratch, 0, ctypes.c_size_t(4096), st) == 0
    assert hip.hipStreamSynchronize(st) == 0
    dummies.append(st)


N, k, steps = 512, 4, 300
dt = 0.25 * qfa.hbar(N)
members = []
for s in range(k):
    members.append(qfa.DeviceTrajectory(qfa.ensemble.make_W0(N, s)))
    for _ in range(plan.get(s, 0)):
        dummy_stream()
ens = qfa.DeviceEnsemble.__new__(qfa.DeviceEnsemble)
ens.members, ens.c64, ens._lib = members, False, members[0]._lib


def timed():
    t_end = time.perf_counter() + 0.15
    while time.perf_counter() < t_end:
        ens.advance(dt, 10)
    ens.advance(dt, 20)
    ens.sync()
    t0 = time.perf_counter()
    ens.advance(dt, steps)
    ens.sync()
    return round(k * steps / (time.perf_counter() - t0))


print(json.dumps({"dummies_after_member": spec, "x4": [timed(), timed()]}), flush=True)
