"""Device contexts: one `qf_ctx` per (device, N), created once and cached --
the analogue of constructing IsompCUDA(N, dtype) / DiagTriDiagOp(N, dtype) once
(quflow/experimental/isospectral_cuda.py:52-80, quflow/simulation.py:554-562)."""
import ctypes
import os

import numpy as np

from . import _lib

_contexts = {}
_default_device = None


def default_device():
    """LOCAL_RANK selects the GPU in one-process-per-GPU launches (torch.distributed.run)."""
    global _default_device
    if _default_device is None:
        _default_device = int(os.environ.get("QUFLOW_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    return _default_device


def set_device(index):
    global _default_device
    _default_device = int(index)


class Context:
    def __init__(self, N, device=None):
        self.N = int(N)
        self.device = default_device() if device is None else int(device)
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(self._lib.qf_ctx_create(self.N, self.device, ctypes.byref(h)))
        self.handle = h

    def plan(self):
        """What this context launched for each role of the hot path since it was created (qf_plan_describe): a dict
        with the kernel, tile, workgroups and -- for the second product -- the share of the tile grid it multiplies,
        recorded by the launchers themselves; None for a role that has not run."""
        import json
        n = self._lib.qf_plan_describe(self.handle, None, 0)
        if n < 0:
            _lib.check(-n)
        buf = ctypes.create_string_buffer(n + 1)
        self._lib.qf_plan_describe(self.handle, buf, n + 1)
        return json.loads(buf.value.decode())

    def close(self):
        if self.handle:
            self._lib.qf_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _cached(cache, N, device):
    dev = default_device() if device is None else int(device)
    key = (dev, int(N))
    ctx = cache.get(key)
    if ctx is None:
        try:
            ctx = Context(N, dev)
        except _lib.QuflowHipError as exc:
            # Contexts are kept for the life of the process (0.25 GB at N = 1024, 15 GB at N = 8192): a sweep over many sizes
            # can fill the device with contexts of sizes it has left behind.  Drop the ones nobody holds and try once more.
            if "out of memory" not in str(exc).lower() or release_idle_contexts() == 0:
                raise
            ctx = Context(N, dev)
        cache[key] = ctx
    return ctx


def get_context(N, device=None):
    return _cached(_contexts, N, device)


_stepper_contexts = {}


def get_stepper_context(N, device=None):
    """A context of its own for a stepper call that runs user hooks while its trajectory is resident on
    the device: the hooks may call the host-in/host-out entry points (solve_poisson, energy_euler,
    solve_viscdamp, ...), which stage through the shared per-N context of get_context() and would
    otherwise overwrite the resident state."""
    return _cached(_stepper_contexts, N, device)


def release_idle_contexts():
    """Closes every cached context that nothing but the cache refers to (device objects such as PoissonHIP / IsompHIP hold
    theirs and keep them); returns how many were closed.  Called when a new context does not fit on the device."""
    import sys
    closed = 0
    for cache in (_contexts, _stepper_contexts):
        for key in list(cache):
            ctx = cache[key]
            if sys.getrefcount(ctx) <= 3:       # the cache, `ctx`, getrefcount's own argument
                del cache[key]
                ctx.close()
                closed += 1
            del ctx
    return closed


def release_contexts():
    for ctx in list(_contexts.values()) + list(_stepper_contexts.values()):
        ctx.close()
    _contexts.clear()
    _stepper_contexts.clear()
    _result_cache.clear()


def guard_report():
    """(device allocations fenced, zones found damaged, description of the first damage) of the library's guard zones --
    all zero unless the process was started with QUFLOW_HIP_DEBUG_GUARD=1 (csrc/guard.hip: every device allocation framed
    by two 64 KiB pattern zones that are read back here and whenever an allocation is released)."""
    lib = _lib.load()
    a, d = ctypes.c_longlong(0), ctypes.c_longlong(0)
    text = ctypes.create_string_buffer(512)
    _lib.check(lib.qf_debug_guard_check(ctypes.byref(a), ctypes.byref(d), text, 512))
    return int(a.value), int(d.value), text.value.decode()


_result_cache = {}


def result_array(shape, dtype, tag):
    """Host array for a result the reference hands back as a NEW ndarray per call (`laplace`, the commutators, products:
    `np.zeros_like` + fill).  A brand-new 16 N^2-byte allocation per call is what such a call costs most at large N -- mmap,
    page faults under the download, munmap: 25-38 ms at N = 2048 where the persistent-buffer `solve_poisson` takes 2.4 ms
    (profiles/r06_d2h_fresh_array.txt, `per_call` of the bench line) -- so the last array handed out under `tag` is kept and
    REUSED when nobody else holds a reference to it any more (its reference count says so: views and slices of it count).  A
    caller that kept the previous result gets a distinct array, exactly as with the reference; every entry is overwritten by
    the download either way."""
    import sys
    key = (tag, tuple(shape), np.dtype(dtype).str)
    a = _result_cache.get(key)
    if a is not None and sys.getrefcount(a) <= 3:       # the cache's reference, this local, getrefcount's argument
        return a
    a = np.empty(shape, dtype=dtype)
    _result_cache[key] = a
    return a


def as_c128(a, name="array"):
    """C-contiguous complex128 view/copy of a host matrix."""
    a = np.asarray(a)
    if a.ndim != 2 or a.shape[0] != a.shape[1]:
        raise ValueError("%s must be a square matrix, got shape %s" % (name, a.shape))
    return np.ascontiguousarray(a, dtype=np.complex128)


def ptr(a):
    return ctypes.c_void_p(a.ctypes.data)
