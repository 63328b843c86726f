"""ctypes binding of libquflow_hip.so (C ABI: include/quflow_hip.h).

The library is the product: there is NO Python/CPU fallback.  If the shared object
is missing, or no HIP device is visible, every compute entry point raises.

Note on PyTorch: torch's ROCm wheel bundles its own libamdhip64.so (same SONAME).
If a process needs both (bench.py with --gpus > 1 uses torch.distributed/RCCL for the
diagnostics gather), import torch BEFORE quflow_amd so that both bind to one HIP runtime.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# QUFLOW_HIP_LIB: another build of the SAME library (A/B runs of kernel variants, tools/ab/); it
# must export every symbol of include/quflow_hip.h like the in-tree one
LIB_PATH = os.environ.get("QUFLOW_HIP_LIB") or os.path.join(_HERE, "libquflow_hip.so")

QF_OK = 0
ERR_NAMES = {1: "QF_ERR_INVALID", 2: "QF_ERR_NO_DEVICE", 3: "QF_ERR_HIP", 4: "QF_ERR_STATE", 5: "QF_ERR_CALLBACK",
             6: "QF_ERR_UNSUPPORTED", 7: "QF_ERR_NONFINITE"}

KERNEL_IDS = {"poisson": 0, "gemm1": 1, "gemm2": 2, "norm": 3, "update": 4, "slice": 5}
ERK_METHODS = {"euler": 0, "heun": 1, "rk4": 2}
BUFFER_IDS = {"W": 0, "dW": 1, "Whalf": 2, "Phalf": 3, "PW": 4}


class QuflowHipError(RuntimeError):
    pass


# host hooks of qf_isomp_hooked / qf_erk_hooked (include/quflow_hip.h: qf_isomp_hooks)
HAMILTONIAN_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double)
FORCING_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double)
STRANG_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p)
CALLBACK_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)


class IsompHooks(ctypes.Structure):
    _fields_ = [("user", ctypes.c_void_p),
                ("hamiltonian", HAMILTONIAN_CB),
                ("forcing", FORCING_CB),
                ("strang", STRANG_CB),
                ("callback", CALLBACK_CB),
                ("hamiltonian_takes_time", ctypes.c_int),
                ("forcing_takes_time", ctypes.c_int),
                ("has_time", ctypes.c_int),
                ("time", ctypes.c_double),
                ("skewh", ctypes.c_int),
                ("solve_skewh", ctypes.c_int),
                ("strang_table", ctypes.c_void_p),
                ("strang_key", ctypes.c_ulonglong),
                ("magnetic", ctypes.c_int),
                ("states_p", ctypes.c_int)]


class IsompStats(ctypes.Structure):
    _fields_ = [("total_iterations", ctypes.c_longlong),
                ("number_of_maxit", ctypes.c_longlong),
                ("tol_used", ctypes.c_double),
                ("last_resnorm", ctypes.c_double)]


# every symbol include/quflow_hip.h declares: (restype, argtypes)
_vp = ctypes.c_void_p
_dp = ctypes.POINTER(ctypes.c_double)
SIGNATURES = {
    "qf_version": (ctypes.c_int, []),
    "qf_last_error": (ctypes.c_char_p, []),
    "qf_device_count": (ctypes.c_int, []),
    "qf_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp)]),
    "qf_ctx_destroy": (ctypes.c_int, [_vp]),
    "qf_ctx_size": (ctypes.c_int, [_vp]),
    "qf_sync": (ctypes.c_int, [_vp]),
    "qf_hbar": (ctypes.c_double, [ctypes.c_int]),
    "qf_laplacian_table": (ctypes.c_int, [_vp, ctypes.c_int, _vp]),
    "qf_solve_poisson": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int]),
    "qf_laplace": (ctypes.c_int, [_vp, _vp, _vp]),
    "qf_solve_tridiagonal": (ctypes.c_int, [_vp, _vp, ctypes.c_ulonglong, _vp, _vp, ctypes.c_int]),
    "qf_factor_cache_stats": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_ulonglong)]),
    "qf_comm_unique_id": (ctypes.c_int, [_vp]),
    "qf_comm_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    "qf_comm_allgather_f64": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp]),
    "qf_comm_barrier": (ctypes.c_int, [_vp]),
    "qf_comm_destroy": (ctypes.c_int, [_vp]),
    "qf_upload_W": (ctypes.c_int, [_vp, _vp]),
    "qf_download_W": (ctypes.c_int, [_vp, _vp]),
    "qf_isomp": (ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(IsompStats)]),
    "qf_isomp_diag": (ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(IsompStats),
                                     ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "qf_isomp_continue": (ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(IsompStats)]),
    "qf_isomp_multi": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double,
                                      ctypes.c_int, ctypes.c_int, ctypes.POINTER(IsompStats)]),
    "qf_c64_isomp_multi": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double,
                                          ctypes.c_int, ctypes.c_int, ctypes.POINTER(IsompStats)]),
    "qf_erk": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int]),
    "qf_erk_states": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int]),
    "qf_isomp_simple_hooked": (ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_int, ctypes.POINTER(IsompHooks)]),
    "qf_isomp_quasinewton_hooked": (ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                                   ctypes.POINTER(IsompStats), ctypes.POINTER(IsompHooks)]),
    "qf_isomp_simple": (ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_int]),
    "qf_isomp_quasinewton": (ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                            ctypes.POINTER(IsompStats)]),
    "qf_isomp_states": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.POINTER(IsompStats)]),
    "qf_isomp_hooked": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(IsompHooks),
                                       ctypes.POINTER(IsompStats)]),
    "qf_erk_hooked": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.POINTER(IsompHooks)]),
    "qf_erk_states_hooked": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                            ctypes.POINTER(IsompHooks)]),
    "qf_basis_upload": (ctypes.c_int, [_vp, _vp, ctypes.c_longlong]),
    "qf_basis_compute": (ctypes.c_int, [_vp]),
    "qf_basis_download": (ctypes.c_int, [_vp, _vp, ctypes.c_longlong]),
    "qf_shr2mat": (ctypes.c_int, [_vp, _vp, ctypes.c_longlong, _vp]),
    "qf_mat2shr": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_longlong]),
    "qf_shc2mat": (ctypes.c_int, [_vp, _vp, _vp]),
    "qf_mat2shc": (ctypes.c_int, [_vp, _vp, _vp]),
    "qf_diagnostics": (ctypes.c_int, [_vp, _dp, _dp]),
    "qf_norm_inf_W": (ctypes.c_int, [_vp, _dp]),
    "qf_profile_enable": (ctypes.c_int, [_vp, ctypes.c_int]),
    "qf_profile_reset": (ctypes.c_int, [_vp]),
    "qf_profile_read": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong), _dp]),
    "qf_plan_describe": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_int]),
    "qf_device_info": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_int]),
    "qf_debug_modulus": (ctypes.c_int, [_vp, ctypes.c_int, _dp, _dp, _dp, _dp]),
    "qf_profile_stride": (ctypes.c_int, [_vp, ctypes.c_int]),
    "qf_profile_seen": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong)]),
    "qf_timer_start": (ctypes.c_int, [_vp]),
    "qf_timer_stop": (ctypes.c_int, [_vp, _dp]),
    "qf_download_buffer": (ctypes.c_int, [_vp, ctypes.c_int, _vp]),
    "qf_debug_guard_check": (ctypes.c_int, [ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong), ctypes.c_char_p, ctypes.c_int]),
    "qf_zgemm": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "qf_commutator": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int]),
    "qf_zgemm_i8": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "qf_fixedpoint_products": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int, _vp, _vp, _vp]),
    "qf_c64_laplacian_table": (ctypes.c_int, [_vp, ctypes.c_int, _vp]),
    "qf_c64_solve_poisson": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int]),
    "qf_c64_laplace": (ctypes.c_int, [_vp, _vp, _vp]),
    "qf_c64_solve_tridiagonal": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int]),
    "qf_c64_upload_W": (ctypes.c_int, [_vp, _vp]),
    "qf_c64_download_W": (ctypes.c_int, [_vp, _vp]),
    "qf_c64_isomp": (ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(IsompStats)]),
    "qf_c64_isomp_continue": (ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(IsompStats)]),
    "qf_c64_diagnostics": (ctypes.c_int, [_vp, _dp, _dp]),
    "qf_cgemm": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "qf_c64_fixedpoint_products": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "qf_c64_fixedpoint_products_tri": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


def load():
    """Load libquflow_hip.so (built by __graft_entry__.build() / make -C quflow_amd/csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise QuflowHipError(
                "libquflow_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C quflow_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc == 7:
        # QF_ERR_NONFINITE: the residual of a stepper's exit test is inf / NaN -- exactly where the reference's
        # scipy.linalg.norm(..., ord=inf) raises (isospectral.py:534, check_finite), with its message
        raise ValueError("array must not contain infs or NaNs")
    if rc != QF_OK:
        msg = load().qf_last_error()
        raise QuflowHipError("%s: %s" % (ERR_NAMES.get(rc, "error %d" % rc),
                                         msg.decode("utf-8", "replace") if msg else ""))


def device_count():
    return load().qf_device_count()


def device_info(device):
    """The HIP device with that ordinal as this process sees it (qf_device_info): ordinal, PCI bus id, name, arch,
    compute units, memory -- what a rank of a multi-GPU launch reports about the device it bound."""
    import json
    lib = load()
    n = lib.qf_device_info(int(device), None, 0)
    if n < 0:
        check(-n)
    buf = ctypes.create_string_buffer(n + 1)
    lib.qf_device_info(int(device), buf, n + 1)
    return json.loads(buf.value.decode())
