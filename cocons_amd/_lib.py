"""ctypes binding of libcocons_hip.so (the C ABI declared in include/cocons_hip.h).

There is NO CPU fallback: if the HIP library is missing or a call fails, an
exception is raised.  The library is built in-tree by `__graft_entry__.build()`
(or `make -C cocons_amd/csrc`).
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COCONS_HIP_LIB") or os.path.join(_HERE, "csrc", "libcocons_hip.so")   # (override: A/B of two builds)
P_MAX = 32

_lib = None

c_dp = ctypes.POINTER(ctypes.c_double)
UNIQUE_ID_BYTES = 128
# int (*)(void *user, void *dev_ptr, long long bytes, int root, void *stream)
BCAST_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p)
# int (*)(void *user, double *host_inout, int count, int op)
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int)
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p)
c_int = ctypes.c_int
c_vp = ctypes.c_void_p


class CoconsHipError(RuntimeError):
    pass


class CholeskyError(CoconsHipError):
    """Leading minor not positive (the reference's `chol` error condition)."""

    def __init__(self, k):
        super().__init__("Cholesky error (leading minor %d not positive)" % k)
        self.minor = k


# every symbol include/cocons_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "cocons_last_error": (ctypes.c_char_p, []),
    "cocons_abi_version": (c_int, []),
    "cocons_device_count": (c_int, []),
    "cocons_cov_rns": (c_int, [c_int, c_int, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "cocons_cov_rns_classic": (c_int, [c_int, c_int, c_dp, c_dp, c_dp, c_dp]),
    "cocons_cov_rns_pred": (c_int, [c_int, c_int, c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "cocons_cov_rns_taper": (c_int, [c_int, c_int, c_dp, c_dp, c_dp, c_dp, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_dp]),
    "cocons_cov_rns_taper_pred": (c_int, [c_int, c_int, c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_int,
                                          ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_dp]),
    "cocons_cov_rows": (c_int, [c_vp, c_dp, c_int, c_int, ctypes.POINTER(c_int), c_int, c_dp]),
    "cocons_sumsmoothlone": (ctypes.c_double, [c_dp, c_int, ctypes.c_double, ctypes.c_double]),
    "cocons_fit_create": (c_vp, [c_int, c_int, c_int, c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_int]),
    "cocons_predict_taper": (c_int, [c_vp, c_dp, c_dp, c_int, c_int, c_dp, c_dp, c_int, ctypes.POINTER(c_int),
                                     ctypes.POINTER(c_int), c_dp, c_dp, c_dp]),
    "cocons_fit_create_taper": (c_vp, [c_int, c_int, c_int, c_dp, c_dp, c_dp, c_dp, c_int, c_int,
                                       ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_dp]),
    "cocons_fit_destroy": (None, [c_vp]),
    "cocons_neg2loglik_dense": (c_int, [c_vp, c_dp, c_dp, c_dp, c_dp]),
    "cocons_neg2loglik_batch": (c_int, [c_vp, c_int, c_dp, c_dp, c_dp, ctypes.POINTER(c_int)]),
    "cocons_neg2loglik_profile": (c_int, [c_vp, c_dp, c_dp, c_dp]),
    "cocons_neg2loglik_reml": (c_int, [c_vp, c_dp, c_int, c_dp, c_dp]),
    "cocons_predict_dense": (c_int, [c_vp, c_dp, c_dp, c_int, c_int, c_dp, c_dp, c_dp, c_dp]),
    "cocons_sim_dense": (c_int, [c_vp, c_dp, c_dp, c_int, c_int, c_dp, c_dp]),
    "cocons_sim_cond_dense": (c_int, [c_vp, c_dp, c_dp, c_int, c_int, c_dp, c_dp, c_dp, c_int, c_dp, c_dp]),
    "cocons_chol_solve": (c_int, [c_int, c_dp, c_int, c_dp, c_dp, c_dp, c_dp]),
    "cocons_fit_profile": (c_int, [c_vp, c_dp, c_dp, c_int, c_dp]),
    "cocons_fit_engine_state": (c_int, [c_vp, ctypes.POINTER(c_int)]),
    "cocons_comm_unique_id": (c_int, [c_vp]),
    "cocons_fit_comm_init": (c_int, [c_vp, c_int, c_int, c_vp]),
    "cocons_fit_set_collectives": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "cocons_fit_world": (c_int, [c_vp]),
    "cocons_multi_create": (c_vp, [c_int, c_int, c_int, c_dp, c_dp, c_dp, c_dp, c_int, ctypes.POINTER(c_int)]),
    "cocons_multi_destroy": (None, [c_vp]),
    "cocons_multi_neg2loglik_dense": (c_int, [c_vp, c_dp, c_dp, c_dp, c_dp]),
    "cocons_multi_predict_dense": (c_int, [c_vp, c_dp, c_dp, c_int, c_int, c_dp, c_dp, c_dp, c_dp]),
    "cocons_multi_neg2loglik_batch": (c_int, [c_vp, c_int, c_dp, c_dp, c_dp, ctypes.POINTER(c_int)]),
    "cocons_multi_comm_ranks": (c_int, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "cocons_fit_comm_info": (c_int, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "cocons_shard_block_owner": (c_int, [c_int, c_int]),
    "cocons_shard_num_blocks": (c_int, [c_vp]),
    "cocons_fit_set_allgather": (c_int, [c_vp, c_vp]),
    "cocons_fit_same_data": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "cocons_fit_stream": (c_vp, [c_vp]),
    "cocons_fit_set_stream": (c_int, [c_vp, c_vp]),
    "cocons_fit_sync": (c_int, [c_vp]),
}


# every symbol include/cocons_hip_diag.h declares (probes and pointwise diagnostics: not the drop-in boundary)
DIAG_SIGNATURES = {
    "cocons_debug_matern": (c_int, [c_int, c_dp, c_dp, c_dp]),
    "cocons_debug_tune": (c_int, [ctypes.c_char_p, c_int]),
    "cocons_debug_dag_replay": (c_int, [c_vp, c_dp, c_dp, c_int, c_dp]),
    "cocons_debug_host_enqueue": (c_int, [c_vp, c_dp]),
    "cocons_debug_dag_words": (c_int, [c_vp, c_int, ctypes.POINTER(ctypes.c_uint)]),
    "cocons_debug_assembly_loop": (c_int, [c_vp, c_dp, c_int, c_dp]),
    "cocons_debug_dag_trace": (ctypes.c_longlong, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                                   ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong)]),
}


# include/cocons_hip_probes.h: bare-instruction probes, in a library of their own (tools only)
PROBES_PATH = os.path.join(_HERE, "csrc", "libcocons_hip_probes.so")
PROBE_SIGNATURES = {
    "cocons_probe_last_error": (ctypes.c_char_p, []),
    "cocons_mfma_f64_probe": (c_int, [c_int, c_dp]),
    "cocons_mfma_f64_probe_ex": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_dp]),
    "cocons_vfma_f64_probe": (c_int, [c_int, c_dp]),
    "cocons_corun_probe": (c_int, [c_int, c_int, c_int, c_int, c_dp]),
}
_probes = None


def load_probes():
    """The probe library (tools/probe_mfma*.py); never loaded by the product path."""
    global _probes
    if _probes is None:
        L = ctypes.CDLL(PROBES_PATH)
        for name, (res, args) in PROBE_SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _probes = L
    return _probes


def load():
    """Load the HIP library (no HIP call is made by loading it)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CoconsHipError(
                "HIP extension not built: %s is missing (run __graft_entry__.build() or "
                "`make -C cocons_amd/csrc`); there is no CPU fallback" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(DIAG_SIGNATURES.items()):
            if name in DIAG_SIGNATURES and os.environ.get("COCONS_HIP_LIB") and not hasattr(L, name):
                continue                   # (an older build alternated on the same box: diagnostics it does not have yet)
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    return load().cocons_last_error().decode("utf-8", "replace")


def check(rc: int, what: str):
    """0 -> ok; k>0 -> CholeskyError(k); <0 -> CoconsHipError."""
    if rc == 0:
        return
    if rc > 0:
        raise CholeskyError(rc)
    raise CoconsHipError("%s failed (%d): %s" % (what, rc, last_error()))
