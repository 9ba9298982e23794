"""ctypes binding of the C ABI declared in ``include/hetmogp_hip.h`` (``libhetmogp_hip.so``).

The north-star asks for a "thin C-ABI cffi layer"; cffi is not installed in this image, ``ctypes`` is, and the
boundary is the C ABI itself, so the stub below is what a maintainer of the reference would add.  There is no
CPU fallback: if the shared library is missing, importing this module raises, loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HMOGP_LIB_PATH") or os.path.join(_HERE, "libhetmogp_hip.so")   # (override: A/B experiments)

ABI_VERSION = 8
# likelihood ids (class names of the reference's likelihoods/<name>.py)
LIK_GAUSSIAN, LIK_BERNOULLI, LIK_HETGAUSSIAN, LIK_CATEGORICAL, LIK_POISSON, LIK_EXPONENTIAL, LIK_GAMMA, LIK_BETA = range(8)
LIK_IDS_BY_NAME = dict(Gaussian=0, Bernoulli=1, HetGaussian=2, Categorical=3, Poisson=4, Exponential=5, Gamma=6, Beta=7)
E_INVALID, E_NO_DEVICE, E_NOT_PD, E_SQI_UNSTABLE, E_STATE, E_COMM = -1, -2, -3, -4, -5, -6
NTIMINGS = 11
COMM_ID_BYTES = 128
FLAG_V_NEGATIVE = 1
FLAG_ILL_CONDITIONED = 2
EVAL_STRICT_QF = 1
EVAL_NO_G_L = 2           # (ABI v7) skip dL/dS L of this evaluation: natural-gradient E-steps
GROUP_QU, GROUP_HYPER, GROUP_Z, GROUP_ALL = 1, 2, 4, 7
CFG_EXACT_ZERO_WINDOWS = 1
CFG_CACHE_KUU = 2
CFG_NO_SMALL_PATH = 4
CFG_STRICT_QF = 8
QUIRK_GAMMA_BETA_PI, QUIRK_CATEGORICAL_DM, QUIRK_STALE_W, QUIRK_W_DIAG, QUIRK_KAPPA_DIAG = 1, 2, 4, 8, 16
QUIRKS_REFERENCE, QUIRKS_EXACT = 31, 0


def quirk_mask(q):
    """"reference" | "exact" | int mask -> int."""
    if isinstance(q, str):
        return {"reference": QUIRKS_REFERENCE, "exact": QUIRKS_EXACT}[q]
    return int(q)

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)
c_uint32_p = C.POINTER(C.c_uint32)


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("T", C.c_int32), ("Q", C.c_int32), ("M", C.c_int32), ("P", C.c_int32),
                ("Df", C.c_int32), ("lik_id", c_int32_p), ("lik_param", c_double_p), ("f_index", c_int32_p),
                ("d_index", c_int32_p), ("device", C.c_int32), ("chunk_rows", C.c_int64), ("flags", C.c_uint32),
                ("quirks", C.c_uint32)]


# (pointer fields are declared void*: same ABI as the typed pointers of include/hetmogp_hip.h, and an array's address -- a plain
#  integer -- can be stored without building a typed ctypes pointer object first: ~1 us per field instead of ~2, which is a
#  measurable share of a 0.2 ms small-model evaluation)
_vp = C.c_void_p


class Params(C.Structure):
    _fields_ = [("Z", _vp), ("m_u", _vp), ("L_flat", _vp), ("variance", _vp), ("lengthscale", _vp), ("W", _vp), ("kappa", _vp),
                ("W0", _vp), ("kappa0", _vp), ("batch_scale", _vp), ("row_begin", _vp), ("row_end", _vp), ("forced_rung", _vp),
                ("group_mask", C.c_uint32), ("eval_flags", C.c_uint32)]


class Outputs(C.Structure):
    _fields_ = [("elbo", _vp), ("g_m_u", _vp), ("g_L_u", _vp), ("g_variance", _vp), ("g_lengthscale", _vp), ("g_W", _vp),
                ("g_kappa", _vp), ("g_Z", _vp), ("dL_dS", _vp), ("rung", _vp), ("flags", _vp), ("kl", _vp), ("cond_est", _vp)]


EXPORTS = {
    # name: (restype, argtypes)  -- one entry per function declared in include/hetmogp_hip.h
    "hmogp_abi_version": (C.c_int, []),
    "hmogp_create": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "hmogp_destroy": (None, [C.c_void_p]),
    "hmogp_last_error": (C.c_char_p, [C.c_void_p]),
    "hmogp_set_task_data": (C.c_int, [C.c_void_p, C.c_int32, c_double_p, c_double_p, C.c_int64]),
    "hmogp_elbo_grad": (C.c_int, [C.c_void_p, C.POINTER(Params), C.POINTER(Outputs)]),
    "hmogp_step_begin": (C.c_int, [C.c_void_p, C.POINTER(Params)]),
    "hmogp_stats_buffer": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), c_int64_p]),
    "hmogp_step_finish": (C.c_int, [C.c_void_p, C.POINTER(Outputs)]),
    "hmogp_comm_available": (C.c_int, []),
    "hmogp_comm_unique_id": (C.c_int, [C.c_void_p]),
    "hmogp_comm_init": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "hmogp_comm_destroy": (C.c_int, [C.c_void_p]),
    "hmogp_comm_info": (C.c_int, [C.c_void_p, c_int32_p, c_int32_p]),
    "hmogp_step_exchange": (C.c_int, [C.c_void_p]),
    "hmogp_elbo_grad_sharded": (C.c_int, [C.c_void_p, C.POINTER(Params), C.POINTER(Outputs)]),
    "hmogp_stats_read": (C.c_int, [C.c_void_p, c_double_p]),
    "hmogp_stats_write": (C.c_int, [C.c_void_p, c_double_p]),
    "hmogp_wire_buffer": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), c_int64_p]),
    "hmogp_wire_pack": (C.c_int, [C.c_void_p]),
    "hmogp_wire_unpack": (C.c_int, [C.c_void_p]),
    "hmogp_wire_read": (C.c_int, [C.c_void_p, c_double_p]),
    "hmogp_wire_write": (C.c_int, [C.c_void_p, c_double_p]),
    "hmogp_posterior_u": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "hmogp_natgrad_step": (C.c_int, [C.c_void_p, C.c_double, c_double_p, c_double_p]),
    "hmogp_qu_natgrad": (C.c_int, [C.c_void_p, C.c_double]),
    "hmogp_qu_natgrad_async": (C.c_int, [C.c_void_p, C.c_double]),
    "hmogp_qu_natgrad_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "hmogp_graph_stats": (C.c_int, [C.c_void_p, c_int64_p, c_int64_p]),
    "hmogp_predict_f": (C.c_int, [C.c_void_p, c_double_p, C.c_int64, c_double_p, c_double_p]),
    "hmogp_last_timings": (C.c_int, [C.c_void_p, c_double_p, c_int64_p]),
    "hmogp_rbf_cross_cov": (C.c_int, [C.c_int32, c_double_p, C.c_int64, c_double_p, C.c_int32, C.c_int32, C.c_double,
                                      C.c_double, c_double_p]),
    "hmogp_rbf_cross_cov_ex": (C.c_int, [C.c_int32, c_double_p, C.c_int64, c_double_p, C.c_int32, C.c_int32, C.c_double,
                                         C.c_double, C.c_int32, c_double_p]),
    "hmogp_qu_load": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "hmogp_qu_read": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "hmogp_qu_adadelta": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]),
    "hmogp_debug_raw_grads": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p]),
    "hmogp_var_exp_ex": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_uint32, C.c_int64, c_double_p, c_double_p,
                                   c_double_p, c_double_p, c_double_p, c_double_p]),
    "hmogp_jitchol_inv": (C.c_int, [C.c_int32, c_double_p, C.c_int32, C.c_int32, c_int32_p, c_double_p, c_double_p,
                                    c_int32_p]),
    "hmogp_potri": (C.c_int, [C.c_int32, c_double_p, C.c_int32, C.c_int32, c_double_p]),
    "hmogp_potrs_rows": (C.c_int, [C.c_int32, c_double_p, C.c_int32, c_double_p, C.c_int64, c_double_p]),
    "hmogp_gemm_f64": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, c_double_p,
                                 C.c_int32, c_double_p, C.c_int32, C.c_double, c_double_p, C.c_int32]),
    "hmogp_predictive": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int64, c_double_p, c_double_p, c_double_p,
                                   c_double_p]),
    "hmogp_log_predictive": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_int64, C.c_int32, C.c_uint64, c_double_p,
                                       c_double_p, c_double_p, c_double_p]),
    "hmogp_sample": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_int64, C.c_uint64, c_double_p, c_double_p]),
    "hmogp_bench_contraction": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, c_double_p]),
    "hmogp_host_alloc": (C.c_void_p, [C.c_uint64]),
    "hmogp_host_free": (None, [C.c_void_p]),
    "hmogp_var_exp": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_int64, c_double_p, c_double_p, c_double_p,
                                c_double_p, c_double_p, c_double_p]),
}


def _preload_torch_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so while this library links the system one (/opt/rocm).  Both can
    live in one process, but only if PyTorch's copy initialises first (observed on ROCm 7.2 + torch 2.10/rocm7.0: with the
    system runtime initialised first, torch.cuda.is_available() turns False and RCCL finds no GPU).  torch is optional
    plumbing (device memory aliasing, torch.distributed); when it is installed, touch it before loading the library."""
    try:
        import torch
        torch.cuda.is_available()
    except Exception:
        pass


def _load():
    _preload_torch_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "hetmogp_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C hetmogp_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.hmogp_abi_version() != ABI_VERSION:
        raise ImportError("hetmogp_amd: ABI version mismatch between _lib.py and libhetmogp_hip.so")
    return lib


lib = _load()


class HetMOGPError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "hetmogp_hip error %d: %s" % (code, msg))
        self.code = code
        self.msg = msg


class InvalidArgument(ValueError):
    """HMOGP_E_INVALID: a programming / ABI error (bad row range, missing array, non-positive lengthscale ...).  A
    ValueError subclass so existing handlers keep working, but distinct from the reference's numerical ValueError
    ("Sqi: Cholesky representation unstable") so that optimisers do not swallow it as a failed evaluation."""


def check(rc, handle=None):
    """Map C-ABI status codes onto the exception types the reference raises on the same conditions."""
    if rc == 0:
        return
    msg = lib.hmogp_last_error(handle)
    msg = msg.decode("utf-8", "replace") if msg else ""
    if rc == E_NOT_PD:
        import numpy as np
        raise np.linalg.LinAlgError(msg)             # GPy jitchol (reached from util.py:198)
    if rc == E_SQI_UNSTABLE:
        raise ValueError(msg)                         # svmogp_inf.py:126-127
    if rc == E_INVALID:
        raise InvalidArgument(msg)
    raise HetMOGPError(rc, msg)
