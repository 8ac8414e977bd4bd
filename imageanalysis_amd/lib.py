"""ctypes binding of libiak.so (include/iak.h).  No fallback: a missing or unloadable
library is an error."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# IAK_LIBRARY names another build of the same library (used to compare builds)
LIB_PATH = os.environ.get("IAK_LIBRARY") or os.path.join(_HERE, "lib", "libiak.so")

IAK_E_ARG, IAK_E_UNSUPPORTED, IAK_E_TOO_LARGE = -1, -2, -3
IAK_LINEAR_CLAMP, IAK_LINEAR_AVERAGE = 0, 1
IAK_PIPE_STAGEWISE, IAK_PIPE_FUSED, IAK_PIPE_FUSED_POINTWISE = 0, 1, 2
(IAK_OP_EQUALIZE, IAK_OP_GAUSS, IAK_OP_SALT_PEPPER, IAK_OP_LINEAR, IAK_OP_AVERAGE, IAK_OP_MEDIAN, IAK_OP_SOBEL,
 IAK_OP_COMPASS, IAK_OP_DILATE, IAK_OP_ERODE, IAK_OP_THRESHOLD, IAK_OP_OTSU, IAK_OP_QUANTIZE, IAK_OP_KMEANS,
 IAK_OP_FEATURES) = range(1, 16)
IAK_INPUT_LUMINANCE, IAK_INPUT_AVERAGE, IAK_INPUT_RED, IAK_INPUT_GREEN, IAK_INPUT_BLUE = range(5)
IAK_N_FEATURES = 7
IAK_GLIBC_RAND_WORDS = 35


class IakOp(C.Structure):
    """iak_op of include/iak.h."""
    _fields_ = [("type", C.c_int), ("kW", C.c_int), ("kH", C.c_int), ("fkernel", C.c_void_p),
                ("ikernel", C.c_void_p), ("iparam", C.c_int), ("mean", C.c_float), ("stddev", C.c_float)]

_vp, _i, _ll, _f = C.c_void_p, C.c_int, C.c_longlong, C.c_float

# name -> argtypes; every function returns int unless listed in _RESTYPE
SIGNATURES = {
    "iak_version": [],
    "iak_set_contraction": [_i],
    "iak_get_contraction": [],
    "iak_error_string": [_i],
    "iak_malloc": [C.POINTER(_vp), C.c_size_t],
    "iak_free": [_vp],
    "iak_memcpy_h2d": [_vp, _vp, C.c_size_t, _vp],
    "iak_memcpy_d2h": [_vp, _vp, C.c_size_t, _vp],
    "iak_memset": [_vp, _i, C.c_size_t, _vp],
    "iak_stream_sync": [_vp],
    "iak_device_count": [C.POINTER(_i)],
    "iak_set_device": [_i],
    "iak_gray": [_vp, _vp, _i, _i, _i, _ll, _ll, _i, _vp],
    "iak_channel_hist": [_vp, _vp, _vp, _i, _i, _i, _ll, _ll, _i, _vp],
    "iak_gray_hist": [_vp, _vp, _vp, _i, _i, _i, _ll, _ll, _i, _vp],
    "iak_hist": [_vp, _vp, _ll, _i, _ll, _vp],
    "iak_eq_map": [_vp, _vp, _i, _i, _vp],
    "iak_apply_map": [_vp, _vp, _ll, _i, _ll, _ll, _vp, _ll, _vp],
    "iak_quantize": [_vp, _vp, _ll, _i, _ll, _ll, _vp, _i, _vp],
    "iak_threshold": [_vp, _vp, _ll, _i, _ll, _ll, _i, _vp],
    "iak_threshold_per_image": [_vp, _vp, _ll, _i, _ll, _ll, _vp, _vp],
    "iak_otsu_threshold": [_vp, _vp, _i, _i, _vp],
    "iak_mean_std": [_vp, _vp, _i, _i, _vp],
    "iak_msqe": [_vp, _vp, _vp, _vp, _ll, _i, _ll, _ll, _vp],
    "iak_rng_planes": [_vp, _vp, _ll, C.c_ulonglong, _vp],
    "iak_rng_planes_from_state": [_vp, _vp, _ll, _vp, _vp],
    "iak_rv_plane": [_vp, _vp, _ll, _vp],
    "iak_sp_plane": [_vp, _vp, _vp, _ll, _i, _vp],
    "iak_salt_pepper": [_vp, _vp, _ll, _i, _ll, _ll, _vp, _vp],
    "iak_gauss_table": [_vp, _i, _i, _f, _f, _vp, _vp],
    "iak_gauss_noise": [_vp, _vp, _ll, _i, _ll, _ll, _vp, _vp, _ll, _vp],
    "iak_linear": [_vp, _vp, _i, _i, _i, _ll, _ll, _vp, _i, _i, _i, _vp],
    "iak_median": [_vp, _vp, _i, _i, _i, _ll, _ll, _vp, _i, _i, _vp],
    "iak_sobel": [_vp, _vp, _i, _i, _i, _ll, _ll, _vp],
    "iak_compass": [_vp, _vp, _i, _i, _i, _ll, _ll, _vp],
    "iak_dilate": [_vp, _vp, _i, _i, _i, _ll, _ll, _vp, _i, _i, _vp],
    "iak_erode": [_vp, _vp, _i, _i, _i, _ll, _ll, _vp, _i, _i, _vp],
    "iak_features_workspace_bytes": [_i, _i, _i],
    "iak_features": [_vp, _i, _i, _i, _ll, _vp, _vp, _vp, _vp],
    "iak_features_static_max_images": [],
    "iak_features_static": [_vp, _i, _i, _i, _ll, _vp, _vp, _vp],
    "iak_dataset_sums": [_vp, _vp, _vp, _i, _i, _vp, _vp],
    "iak_kmeans2_map": [_vp, _i, _i, _vp, _vp, _vp, _vp],
    "iak_glibc_rand_seed": [_vp, C.c_uint, _vp],
    "iak_glibc_rand_centroids": [_vp, _i, _i, _vp, _vp],
    "iak_knn_predict": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "iak_pipeline_bind_features": [_vp, _vp],
    "iak_pipeline_set_input": [_vp, _i],
    "iak_pipeline_create": [C.POINTER(_vp), _i, _i, _i, _vp, _i, C.c_ulonglong, _i],
    "iak_pipeline_destroy": [_vp],
    "iak_pipeline_run": [_vp, _vp, _vp, _i, _vp, _vp, _vp],
    "iak_pipeline_set_lanes": [_vp, _i],
    "iak_pipeline_skip_images": [_vp, _ll, _vp],
    "iak_pipeline_timed_images": [_vp],
    "iak_pipeline_set_timing": [_vp, _i],
    "iak_pipeline_set_timing_every": [_vp, _i],
    "iak_pipeline_timings": [_vp, _i, _vp, _vp, _vp, _vp, _i],
    "iak_pipeline_group_bytes": [_vp, _vp, _i],
}
_RESTYPE = {"iak_error_string": C.c_char_p, "iak_features_workspace_bytes": C.c_longlong}

_lib = None


class IakError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = load().iak_error_string(code).decode()
        super().__init__(f"{where}: {msg} ({code})")


def load():
    """Load libiak.so once.  When torch is in the process, import it first so both share
    one HIP runtime (same soname)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C imageanalysis_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the symbol is not exported
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, C.c_int)
    _lib = L
    return L


def check(rc, where="iak"):
    if rc != 0:
        raise IakError(rc, where)
