"""ctypes binding of libadmm_hip.so (the C ABI declared in include/admm_hip.h).

The library is built in-tree by `python -m admm_amd.build` (hipcc, gfx950).  There is no CPU
fallback: if the shared object is missing or no HIP device is usable, calls raise.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADMM_HIP_LIB") or os.path.join(_HERE, "lib", "libadmm_hip.so")      # ADMM_HIP_LIB: A/B builds (dev)

ADMM_MEM_HOST = 0
ADMM_MEM_DEVICE = 1


class AdmmOpts(ctypes.Structure):
    _fields_ = [("maxit", ctypes.c_int), ("eps_abs", ctypes.c_double),
                ("eps_rel", ctypes.c_double), ("rho", ctypes.c_double)]


class AdmmHipOptions(ctypes.Structure):
    """include/admm_hip.h: admm_hip_options (variant selectors of the calling thread; every field 0 = library default)."""
    _fields_ = [(k, ctypes.c_int) for k in (
        "struct_size", "gram_backend", "gram_split", "factor_backend", "inverse_precision", "tall_xupdate", "tall_refine",
        "consensus_two_pass", "consensus_unfused", "bp_two_pass", "lad_no_hat", "wide_no_persist", "wide_unfused", "wide_gram_sprad",
        "sharing_bp_direct", "cv_downdate", "peer_exchange", "batch_iters", "profile_stride", "pool_mb", "screen", "lad_two_pass")] + [("reserved", ctypes.c_int * 10)]


class AdmmStats(ctypes.Structure):
    _fields_ = [("t_h2d", ctypes.c_double), ("t_standardize", ctypes.c_double), ("t_gram", ctypes.c_double),
                ("t_eigs", ctypes.c_double), ("t_factor", ctypes.c_double), ("t_loop", ctypes.c_double),
                ("t_total", ctypes.c_double), ("loop_ms_events", ctypes.c_double),
                ("xupdate_ms_avg", ctypes.c_double), ("xupdate_samples", ctypes.c_longlong),
                ("total_iter", ctypes.c_longlong), ("xupdate_launches", ctypes.c_longlong),
                ("rho", ctypes.c_double), ("eig_est", ctypes.c_double),
                ("branch", ctypes.c_int), ("xupdate_variant", ctypes.c_int),
                ("exchange_variant", ctypes.c_int), ("refine", ctypes.c_int), ("persist_iter", ctypes.c_longlong),
                ("factor_flops", ctypes.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class AdmmHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libadmm_hip error {code}: {msg}")
        self.code = code


_lib = None

_DP = ctypes.c_void_p      # raw pointers: host numpy buffers or device addresses
_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_float_p = ctypes.POINTER(ctypes.c_float)
_c_int_p = ctypes.POINTER(ctypes.c_int)

# every symbol include/admm_hip.h declares
EXPORTS = ["admm_hip_lasso", "admm_hip_enet", "admm_hip_parlasso", "admm_hip_lad", "admm_hip_bp",
           "admm_hip_last_error", "admm_hip_version", "admm_hip_device_count", "admm_hip_set_device",
           "admm_hip_device_synchronize", "admm_hip_lasso_plan_create", "admm_hip_lasso_plan_run",
           "admm_hip_lasso_plan_destroy", "admm_hip_comm_unique_id", "admm_hip_comm_init", "admm_hip_comm_finalize", "admm_hip_comm_info",
           "admm_hip_parlasso_dist", "admm_hip_lasso_plan_create_dist",
           "admm_hip_lasso_plan_trace_enable", "admm_hip_lasso_plan_trace_read",
           "admm_hip_lasso_plan_state_enable", "admm_hip_lasso_plan_state_read", "admm_hip_lasso_plan_system_read",
           "admm_hip_host_lanczos", "admm_hip_test_symv",
           "admm_hip_comm_peer_prepare", "admm_hip_comm_init_peer", "admm_hip_comm_init_shm", "admm_hip_comm_test_allreduce", "admm_hip_comm_test_reduce_scatter",
           "admm_hip_lasso_dist", "admm_hip_test_gram", "admm_hip_test_spd_inverse", "admm_hip_test_cv_fold_system",
           "admm_hip_lasso_dist_cols", "admm_hip_test_gemv_t", "admm_hip_lad_traced", "admm_hip_bp_traced",
           "admm_hip_lasso_plan_create_dist_cols", "admm_hip_lasso_cv", "admm_hip_lasso_multi",
           "admm_hip_parbp", "admm_hip_parbp_traced", "admm_hip_parbp_dist", "admm_hip_dantzig", "admm_hip_dantzig_traced",
           "admm_hip_lad_state", "admm_hip_bp_state", "admm_hip_lasso_plan_data_read", "admm_hip_trim_memory", "admm_hip_test_gather",
           "admm_hip_options_default", "admm_hip_options_set", "admm_hip_option_set", "admm_hip_options_reset", "admm_hip_option_get"]

TRACE_FIELDS = 12
TRACE_COLD, TRACE_CONVERGED, TRACE_ACCELERATE, TRACE_RESTART = -1, 0, 1, 2
TRACE_CONTINUE = 1


def load():
    """Load the shared library (once). Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("ADMM_HIP_LIB") or LIB_PATH     # ADMM_HIP_LIB: another build of the SAME library (the host-sanitizer variant)
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: build it with `python -m admm_amd.build` (hipcc --offload-arch=gfx950). "
            "admm_amd has no CPU fallback.")
    if not os.environ.get("ADMM_HIP_LIB"):
        # the in-tree library must have been linked from the sources beside it (admm_amd/build.py keeps their hash in a side file):
        # a stale prebuilt .so against a newer ctypes layout is an ABI mismatch, not just stale behaviour
        from . import build as _build
        if os.path.isdir(_build.CSRC) and not _build._lib_is_current(path):
            raise RuntimeError(f"{path} was not built from the sources in {_build.CSRC} (source hash differs or is missing): "
                               "run `python -m admm_amd.build`")
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    lasso_args = [_DP, _DP, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                  _DP, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int]
    tail = [ctypes.POINTER(AdmmOpts), _c_double_p, _c_float_p, _c_int_p, ctypes.POINTER(AdmmStats)]
    lib.admm_hip_lasso.argtypes = lasso_args + tail
    lib.admm_hip_enet.argtypes = lasso_args + [ctypes.c_double] + tail
    lib.admm_hip_parlasso.argtypes = lasso_args + [ctypes.c_int] + tail
    lib.admm_hip_lasso_cv.argtypes = ([_DP, _DP, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_int_p, ctypes.c_int,
                                       _DP, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                       ctypes.POINTER(AdmmOpts), _c_double_p, _c_float_p, _c_int_p,
                                       _c_double_p, _c_double_p, _c_double_p, _c_int_p, _c_float_p, _c_int_p, _c_int_p,
                                       ctypes.POINTER(AdmmStats)])
    lib.admm_hip_lasso_cv.restype = ctypes.c_int
    lib.admm_hip_lasso_multi.argtypes = ([_DP, _DP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          _DP, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                          ctypes.POINTER(AdmmOpts), _c_double_p, _c_float_p, _c_int_p, ctypes.POINTER(AdmmStats)])
    lib.admm_hip_lasso_multi.restype = ctypes.c_int
    lib.admm_hip_lad.argtypes = [_DP, _DP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 ctypes.POINTER(AdmmOpts), _c_double_p, _c_int_p, ctypes.POINTER(AdmmStats)]
    lib.admm_hip_bp.argtypes = [_DP, _DP, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                ctypes.POINTER(AdmmOpts), _c_double_p, _c_int_p, ctypes.POINTER(AdmmStats)]
    lib.admm_hip_lad_traced.argtypes = lib.admm_hip_lad.argtypes + [_c_double_p, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong)]
    lib.admm_hip_lad_traced.restype = ctypes.c_int
    lib.admm_hip_bp_traced.argtypes = lib.admm_hip_bp.argtypes + [_c_double_p, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong)]
    lib.admm_hip_bp_traced.restype = ctypes.c_int
    lib.admm_hip_lad_state.argtypes = lib.admm_hip_lad_traced.argtypes + [_c_double_p, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong)]
    lib.admm_hip_lad_state.restype = ctypes.c_int
    lib.admm_hip_bp_state.argtypes = lib.admm_hip_bp_traced.argtypes + [_c_double_p, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong)]
    lib.admm_hip_bp_state.restype = ctypes.c_int
    lib.admm_hip_dantzig.argtypes = [_DP, _DP, ctypes.c_int, ctypes.c_int, ctypes.c_int, _DP, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                     ctypes.c_int, ctypes.c_int, ctypes.POINTER(AdmmOpts), _c_double_p, _c_double_p, _c_int_p, ctypes.POINTER(AdmmStats)]
    lib.admm_hip_dantzig.restype = ctypes.c_int
    lib.admm_hip_dantzig_traced.argtypes = lib.admm_hip_dantzig.argtypes + [_c_double_p, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong)]
    lib.admm_hip_dantzig_traced.restype = ctypes.c_int
    lib.admm_hip_parbp.argtypes = [_DP, _DP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.POINTER(AdmmOpts), _c_double_p, _c_int_p, ctypes.POINTER(AdmmStats)]
    lib.admm_hip_parbp.restype = ctypes.c_int
    lib.admm_hip_parbp_traced.argtypes = lib.admm_hip_parbp.argtypes + [_c_double_p, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong)]
    lib.admm_hip_parbp_traced.restype = ctypes.c_int
    lib.admm_hip_parbp_dist.argtypes = [_DP, _DP, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
                                        ctypes.POINTER(AdmmOpts), _c_double_p, _c_int_p, ctypes.POINTER(AdmmStats)]
    lib.admm_hip_parbp_dist.restype = ctypes.c_int
    for name in ("admm_hip_lasso", "admm_hip_enet", "admm_hip_parlasso", "admm_hip_lad", "admm_hip_bp",
                 "admm_hip_device_count", "admm_hip_set_device", "admm_hip_device_synchronize"):
        getattr(lib, name).restype = ctypes.c_int
    lib.admm_hip_set_device.argtypes = [ctypes.c_int]
    lib.admm_hip_last_error.restype = ctypes.c_char_p
    lib.admm_hip_version.restype = ctypes.c_char_p
    lib.admm_hip_lasso_plan_create.argtypes = lasso_args + [ctypes.c_double, ctypes.c_int, ctypes.POINTER(AdmmOpts),
                                               ctypes.POINTER(ctypes.c_void_p), _c_int_p]
    lib.admm_hip_lasso_plan_create.restype = ctypes.c_int
    lib.admm_hip_lasso_plan_run.argtypes = [ctypes.c_void_p, _c_double_p, _c_float_p, _c_int_p, ctypes.POINTER(AdmmStats)]
    lib.admm_hip_lasso_plan_run.restype = ctypes.c_int
    lib.admm_hip_lasso_plan_destroy.argtypes = [ctypes.c_void_p]
    lib.admm_hip_lasso_plan_destroy.restype = ctypes.c_int
    dist_args = [_DP, _DP, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
                 _DP, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(AdmmOpts)]
    lib.admm_hip_parlasso_dist.argtypes = dist_args + [_c_double_p, _c_float_p, _c_int_p, ctypes.POINTER(AdmmStats)]
    lib.admm_hip_parlasso_dist.restype = ctypes.c_int
    lib.admm_hip_lasso_plan_create_dist.argtypes = dist_args + [ctypes.POINTER(ctypes.c_void_p), _c_int_p]
    lib.admm_hip_lasso_plan_create_dist.restype = ctypes.c_int
    lib.admm_hip_lasso_dist.argtypes = [_DP, _DP, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
                                        _DP, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                        ctypes.POINTER(AdmmOpts), _c_double_p, _c_float_p, _c_int_p, ctypes.POINTER(AdmmStats)]
    lib.admm_hip_lasso_dist.restype = ctypes.c_int
    lib.admm_hip_lasso_dist_cols.argtypes = [_DP, _DP, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_int,
                                             _DP, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                             ctypes.POINTER(AdmmOpts), _c_double_p, _c_float_p, _c_int_p, ctypes.POINTER(AdmmStats)]
    lib.admm_hip_lasso_dist_cols.restype = ctypes.c_int
    lib.admm_hip_lasso_plan_create_dist_cols.argtypes = lib.admm_hip_lasso_dist_cols.argtypes[:15] + [ctypes.POINTER(ctypes.c_void_p), _c_int_p]
    lib.admm_hip_lasso_plan_create_dist_cols.restype = ctypes.c_int
    lib.admm_hip_comm_unique_id.argtypes = [ctypes.c_void_p]
    lib.admm_hip_comm_unique_id.restype = ctypes.c_int
    lib.admm_hip_comm_init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.admm_hip_comm_init.restype = ctypes.c_int
    lib.admm_hip_comm_finalize.argtypes = []
    lib.admm_hip_comm_finalize.restype = ctypes.c_int
    lib.admm_hip_options_default.argtypes = [ctypes.POINTER(AdmmHipOptions)]
    lib.admm_hip_options_default.restype = ctypes.c_int
    lib.admm_hip_options_set.argtypes = [ctypes.POINTER(AdmmHipOptions)]
    lib.admm_hip_options_set.restype = ctypes.c_int
    lib.admm_hip_option_set.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    lib.admm_hip_option_set.restype = ctypes.c_int
    lib.admm_hip_options_reset.argtypes = []
    lib.admm_hip_options_reset.restype = ctypes.c_int
    lib.admm_hip_option_get.argtypes = [ctypes.c_char_p]
    lib.admm_hip_option_get.restype = ctypes.c_char_p
    lib.admm_hip_comm_info.argtypes = [ctypes.POINTER(ctypes.c_int)] * 3
    lib.admm_hip_comm_info.restype = ctypes.c_int
    lib.admm_hip_comm_peer_prepare.argtypes = [ctypes.c_int, ctypes.c_void_p]
    lib.admm_hip_comm_peer_prepare.restype = ctypes.c_int
    lib.admm_hip_comm_init_peer.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.admm_hip_comm_init_peer.restype = ctypes.c_int
    lib.admm_hip_comm_init_shm.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_ulonglong]
    lib.admm_hip_comm_init_shm.restype = ctypes.c_int
    lib.admm_hip_comm_test_allreduce.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int]
    lib.admm_hip_comm_test_allreduce.restype = ctypes.c_int
    lib.admm_hip_comm_test_reduce_scatter.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
    lib.admm_hip_comm_test_reduce_scatter.restype = ctypes.c_int
    lib.admm_hip_lasso_plan_trace_enable.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
    lib.admm_hip_lasso_plan_trace_enable.restype = ctypes.c_int
    lib.admm_hip_lasso_plan_trace_read.argtypes = [ctypes.c_void_p, _c_double_p, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong)]
    lib.admm_hip_lasso_plan_trace_read.restype = ctypes.c_int
    lib.admm_hip_lasso_plan_state_enable.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
    lib.admm_hip_lasso_plan_state_enable.restype = ctypes.c_int
    lib.admm_hip_lasso_plan_state_read.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_longlong,
                                                   ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)]
    lib.admm_hip_lasso_plan_state_read.restype = ctypes.c_int
    lib.admm_hip_lasso_plan_data_read.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_longlong, ctypes.POINTER(ctypes.c_float)]
    lib.admm_hip_lasso_plan_data_read.restype = ctypes.c_int
    lib.admm_hip_lasso_plan_system_read.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_longlong]
    lib.admm_hip_lasso_plan_system_read.restype = ctypes.c_int
    lib.admm_hip_test_symv.argtypes = [_c_float_p, ctypes.c_int, _c_float_p, _c_float_p, _c_float_p, _c_float_p]
    lib.admm_hip_test_symv.restype = ctypes.c_int
    lib.admm_hip_test_gram.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.admm_hip_test_gram.restype = ctypes.c_int
    lib.admm_hip_test_gemv_t.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.admm_hip_test_gemv_t.restype = ctypes.c_int
    lib.admm_hip_test_spd_inverse.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.admm_hip_test_spd_inverse.restype = ctypes.c_int
    lib.admm_hip_test_cv_fold_system.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.admm_hip_test_cv_fold_system.restype = ctypes.c_int
    lib.admm_hip_host_lanczos.argtypes = [_c_float_p, ctypes.c_int, _c_float_p, _c_int_p]
    lib.admm_hip_host_lanczos.restype = ctypes.c_int
    lib.admm_hip_test_gather.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, _c_double_p]
    lib.admm_hip_test_gather.restype = ctypes.c_int
    lib.admm_hip_trim_memory.argtypes = []
    lib.admm_hip_trim_memory.restype = ctypes.c_int
    _lib = lib
    return lib


def check(code):
    if code != 0:
        raise AdmmHipError(code, load().admm_hip_last_error().decode())


class DevicePtr:
    """A raw device address (e.g. torch_tensor.data_ptr()) of column-major float64 data."""

    def __init__(self, ptr):
        self.ptr = int(ptr)


def as_input(a, n=None, p=None):
    """Return (c_void_p, mem, keepalive) for a host array (copied to column-major float64) or a DevicePtr."""
    if isinstance(a, DevicePtr):
        return ctypes.c_void_p(a.ptr), ADMM_MEM_DEVICE, a
    arr = np.asfortranarray(np.asarray(a, dtype=np.float64))
    return ctypes.c_void_p(arr.ctypes.data), ADMM_MEM_HOST, arr


class options:
    """Variant selectors / tuning values of the CALLING THREAD for the duration of a `with` block (admm_hip_option_set; names as in
    INTEGRATION.md, case-insensitive, with or without the ADMM_HIP_ prefix; value None = library default):

        with admm_amd.options(GRAM_SPLIT="f16x2", INVERSE="f64"):
            fit = admm_amd.admm_lasso(x, y).penalty(nlambda=20).fit()

    A prepared problem (LassoPlan) reads them when it is created.  On exit the previous values of exactly these names are restored.
    Fields of the typed struct go through `options.struct(gram_split=2, ...)` (admm_hip_options_set: replaces ALL of the thread's settings)."""

    def __init__(self, **kw):
        self.kw = {k.upper(): (None if v is None else str(v)) for k, v in kw.items()}
        self.old = {}

    def __enter__(self):
        lib = load()
        for k, v in self.kw.items():
            cur = lib.admm_hip_option_get(k.encode())
            self.old[k] = cur.decode() if cur is not None else None
            check(lib.admm_hip_option_set(k.encode(), None if v is None else v.encode()))
        return self

    def __exit__(self, *exc):
        lib = load()
        for k, v in self.old.items():
            check(lib.admm_hip_option_set(k.encode(), None if v is None else v.encode()))
        return False

    @staticmethod
    def set(**kw):
        lib = load()
        for k, v in kw.items():
            check(lib.admm_hip_option_set(k.upper().encode(), None if v is None else str(v).encode()))

    @staticmethod
    def reset():
        check(load().admm_hip_options_reset())

    @staticmethod
    def struct(**fields):
        lib = load()
        o = AdmmHipOptions()
        check(lib.admm_hip_options_default(ctypes.byref(o)))
        for k, v in fields.items():
            setattr(o, k, int(v))
        check(lib.admm_hip_options_set(ctypes.byref(o)))
        return o
