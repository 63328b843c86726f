"""round 6 probe 2: four replicas at N = 512 (~18,000 timesteps/s) drop to ~11,000 when ANOTHER stream of the process executes a
fill while the contexts are being created (profiles/r06_x4_other_streams.txt).  What about ordinary neighbours?  Variants:
  plain            nothing else in the process
  memset_after     one hipMemset (NULL stream) on a scratch allocation AFTER the ensemble exists, before the timed advance
  memset_between   a hipMemset between two timed advances of the same ensemble
  torch_before     import torch, one CUDA tensor op + synchronize BEFORE the library's first context
  torch_after      the same AFTER the ensemble exists
Usage: python tools/gpu/r6_queue_probe2.py <variant>"""
import ctypes
import json
import os
import sys
import time

variant = sys.argv[1] if len(sys.argv) > 1 else "plain"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(v, "8")
torch = None
if variant.startswith("torch"):
    import torch                     # (first: its HIP runtime is the one the library binds to)
    if variant == "torch_before":
        x = torch.zeros(1 << 20, device="cuda"); x += 1; torch.cuda.synchronize()
import quflow_amd as qfa  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so") if torch is None else None


def null_memset():
    if torch is not None:
        y = torch.zeros(1 << 20, device="cuda"); y += 1; torch.cuda.synchronize()
        return
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 20)) == 0
    assert hip.hipMemset(p, 0, ctypes.c_size_t(1 << 20)) == 0
    assert hip.hipDeviceSynchronize() == 0


N, k, steps = 512, 4, 300
dt = 0.25 * qfa.hbar(N)
ens = qfa.DeviceEnsemble([qfa.ensemble.make_W0(N, s) for s in range(k)])
if variant in ("memset_after", "torch_after"):
    null_memset()


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


out = {"variant": variant, "x4": [timed()]}
if variant == "memset_between":
    null_memset()
out["x4"].append(timed())
out["x4"].append(timed())
print(json.dumps(out), flush=True)
