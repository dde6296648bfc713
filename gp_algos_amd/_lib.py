"""ctypes binding of libgpcore.so (include/gpcore.h).  There is no CPU fallback: if the HIP library
is missing or no gfx950 device is usable, calls raise."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPCORE_LIB_PATH") or os.path.join(_HERE, "libgpcore.so")   # GPCORE_LIB_PATH: lab builds (tools/lab)

GP_OK, GP_EINVAL, GP_ENOTPD, GP_ENOMEM, GP_EHIP, GP_ERANGE, GP_ERCCL, GP_EPEER = range(8)
GP_DIST_ID_BYTES = 128
GP_LOWER, GP_FULL = 0, 1
GP_GET_L, GP_GET_ALPHA, GP_GET_LML = 0, 1, 2
GP_PROF_OFF, GP_PROF_GEMM, GP_PROF_SYRK, GP_PROF_GRAM, GP_PROF_TRSM, GP_PROF_POTRF_DIAG, GP_PROF_PANEL_UPD = range(7)
GP_EP_GET_L, GP_EP_GET_SIGMA, GP_EP_GET_MU, GP_EP_GET_CAV_TAU, GP_EP_GET_CAV_NU = range(5)
GP_SMALL_GET_L, GP_SMALL_GET_LINV, GP_SMALL_GET_ALPHA = range(3)
GP_SMALL_MAX_N = 2048

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p
_i, _d, _sz = C.c_int, C.c_double, C.c_size_t

# name -> (restype, argtypes); mirrors include/gpcore.h declaration by declaration
SIGNATURES = {
    "gp_version": (C.c_char_p, []),
    "gp_ctx_create": (_i, [_i, _vp, C.POINTER(_vp)]),
    "gp_ctx_destroy": (None, [_vp]),
    "gp_ctx_sync": (_i, [_vp]),
    "gp_ctx_trim": (_i, [_vp]),
    "gp_last_error": (C.c_char_p, [_vp]),
    "gp_ctx_profile": (_i, [_vp, _i]),
    "gp_ctx_set_lookahead": (_i, [_vp, _i]),
    "gp_ctx_profile_read": (_i, [_vp, _i, C.POINTER(C.c_int64), _dp, _dp]),
    "gp_chol_plan_info": (_i, [_i, _i, _i, _ip, _dp]),
    "gp_probe_mfma_f64": (_i, [_vp, _dp]),
    "gp_probe_mfma_f64_ex": (_i, [_vp, _i, _dp, _dp, _dp]),
    "gp_dev_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "gp_dev_free": (_i, [_vp, _vp]),
    "gp_dev_upload": (_i, [_vp, _vp, _vp, _sz]),
    "gp_dev_download": (_i, [_vp, _vp, _vp, _sz]),
    "gp_hp_get_at_position": (_i, [_dp, _i, _i, _dp]),
    "gp_gram_rbf": (_i, [_vp, _dp, _i, _i, _i, _dp, _dp, _i, _i]),
    "gp_dgram_rbf": (_i, [_vp, _dp, _i, _i, _i, _dp, _i, _dp, _i]),
    "gp_gram_rbf_dev": (_i, [_vp, _vp, _i, _i, _i, _dp, _vp, _i, _i]),
    "gp_cross_gram_rbf": (_i, [_vp, _dp, _i, _i, _dp, _i, _i, _i, _dp, _dp, _i]),
    "gp_potrf_lower": (_i, [_vp, _dp, _i, _i, _ip]),
    "gp_trsm_lower": (_i, [_vp, _i, _dp, _i, _i, _dp, _i, _i]),
    "gp_inv_lower": (_i, [_vp, _dp, _i, _i, _dp, _i]),
    "gp_fit_rbf": (_i, [_vp, _dp, _i, _i, _i, _dp, _dp, _d, C.POINTER(_vp), _ip]),
    "gp_fit_rbf_dev": (_i, [_vp, _vp, _i, _i, _i, _vp, _dp, _d, C.POINTER(_vp), _ip]),
    "gp_fit_from_gram": (_i, [_vp, _dp, _i, _i, _dp, C.POINTER(_vp), _ip]),
    "gp_model_refit_dev": (_i, [_vp, _dp, _d]),
    "gp_model_status": (_i, [_vp, _ip]),
    "gp_model_get": (_i, [_vp, _i, _dp, _i]),
    "gp_model_destroy": (None, [_vp]),
    "gp_predict": (_i, [_vp, _dp, _i, _i, _dp, _dp, _dp, _i]),
    "gp_predict_dev": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "gp_posterior_from_factor": (_i, [_vp, _dp, _i, _i, _i, _dp, _dp, _i, _dp, _dp, _i, _i, _dp, _dp, _dp, _i, _dp, _i]),
    "gp_posterior_from_gram": (_i, [_vp, _dp, _i, _i, _i, _dp, _i, _dp, _dp, _i, _dp, _dp, _dp, _dp, _i, _dp, _i]),
    "gp_predict_from_gram": (_i, [_vp, _dp, _i, _i, _dp, _i, _dp, _dp, _dp, _dp, _i]),
    "gp_lml_grad_rbf_batched": (_i, [_vp, _dp, _i, _i, _i, _dp, _dp, _i, _i, _d, _dp, _dp, _ip]),
    "gp_lml_grad_from_gram": (_i, [_vp, _dp, _i, _i, _dp, C.POINTER(_dp), _i, _i, _d, _dp, _dp, _ip]),
    "gp_optimize_rbf": (_i, [_vp, _dp, _i, _i, _i, _dp, _dp, _i, _d, _i, _i, _dp, _dp, _ip, _ip]),
    "gp_ep_create": (_i, [_vp, _dp, _i, _i, C.POINTER(C.c_int32), C.POINTER(_vp)]),
    "gp_ep_sweep": (_i, [_vp, _i, _dp, _dp, _ip]),
    "gp_ep_set_site_params": (_i, [_vp, _dp, _dp, _ip]),
    "gp_ep_lml": (_i, [_vp, _i, _dp]),
    "gp_ep_lml_grad_rbf": (_i, [_vp, _dp, _i, _i, _dp, _i, _dp]),
    "gp_ep_get": (_i, [_vp, _i, _dp, _i]),
    "gp_ep_predict": (_i, [_vp, _dp, _i, _i, _dp, _dp]),
    "gp_ep_lml_rbf_batched": (_i, [_vp, _dp, _i, _i, _i, C.POINTER(C.c_int32), _dp, _i, _d, _i, _i, _dp, _ip, _ip]),
    "gp_ep_lml_grad_rbf_batched": (_i, [_vp, _dp, _i, _i, _i, C.POINTER(C.c_int32), _dp, _i, _d, _i, _i, _dp, _dp, _ip, _ip]),
    "gp_ep_optimize_rbf": (_i, [_vp, _dp, _i, _i, _i, C.POINTER(C.c_int32), _dp, _d, _i, _i, _i, _i, _dp, _dp, _ip, _ip]),
    "gp_ep_destroy": (None, [_vp]),
    "gp_gram_co2": (_i, [_vp, _dp, _i, _dp, _dp, _i, _i]),
    "gp_dgram_co2": (_i, [_vp, _dp, _i, _dp, _i, _dp, _i]),
    "gp_cross_gram_co2": (_i, [_vp, _dp, _i, _dp, _i, _dp, _dp, _i]),
    "gp_fit_co2": (_i, [_vp, _dp, _i, _dp, _dp, _d, C.POINTER(_vp), _ip]),
    "gp_lml_grad_co2_batched": (_i, [_vp, _dp, _i, _dp, _dp, _i, _i, _d, _dp, _dp, _ip]),
    "gp_optimize_co2": (_i, [_vp, _dp, _i, _dp, _dp, _i, _d, _i, _i, _dp, _dp, _ip, _ip]),
    "gp_small_fit": (_i, [_vp, _dp, _i, _i, _i, _dp, _i, _i, _dp, _d, _i, C.POINTER(_vp), _ip]),
    "gp_small_from_factors": (_i, [_vp, _dp, _i, _i, _i, _dp, _i, _dp, _i, _dp, _i, C.POINTER(_vp)]),
    "gp_small_destroy": (None, [_vp]),
    "gp_small_size": (_i, [_vp, _ip, _ip, _ip]),
    "gp_small_get": (_i, [_vp, _i, _i, _dp, _i]),
    "gp_small_posterior": (_i, [_vp, _dp, _i, _i, _dp, _dp]),
    "gp_small_ucb": (_i, [_vp, _i, _dp, _i, _i, _d, _dp, _dp]),
    "gp_small_append": (_i, [_vp, _dp, _dp, _ip]),
    "gp_small_maximize_ucb": (_i, [_vp, _i, _dp, _i, _i, _d, _i, _i, _dp, _dp, _ip]),
    "gp_dist_unique_id": (_i, [_vp, C.c_char_p]),
    "gp_dist_init": (_i, [_vp, C.c_char_p, _i, _i, C.POINTER(_vp)]),
    "gp_dist_destroy": (None, [_vp]),
    "gp_dist_status_scan": (_i, [_dp, _i, _i, _ip]),
    "gp_dist_inject_failure": (_i, [_vp]),
    "gp_dist_shard": (_i, [_vp, _i, _ip, _ip]),
    "gp_dist_lml_grad_batched": (_i, [_vp, _dp, _i, _i, _i, _dp, _dp, _i, _i, _d, _dp, _dp, _ip]),
    "gp_dist_predict": (_i, [_vp, _vp, _dp, _i, _i, _dp, _dp]),
}

_lib = None


class GpCoreError(RuntimeError):
    def __init__(self, status, msg, info=0):
        super().__init__("gpcore status %d: %s" % (status, msg))
        self.status = status
        self.info = info


class PeerFailure(GpCoreError):
    """gp_dist_* / dist.py: another rank of the group failed; nothing was exchanged and this rank is intact (GP_EPEER)."""


class NotPositiveDefinite(GpCoreError):
    """Breeze's cholesky throws on a non-PD matrix; .info is the 1-based failing pivot."""


def load():
    """dlopen libgpcore.so and attach prototypes.  Raises if the HIP extension is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libgpcore.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def dptr(a):
    return a.ctypes.data_as(_dp)


def f64(a):
    """float64 array; 2-D arrays become Fortran (column-major) contiguous like Breeze DenseMatrix."""
    a = np.asarray(a, dtype=np.float64)
    return np.asfortranarray(a) if a.ndim == 2 else np.ascontiguousarray(a)
