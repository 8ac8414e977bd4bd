"""ctypes binding of the CPU oracle (oracle/ia_oracle.c) -- the checker used by the
tests, the smoke run and bench.py's cpu_baseline leg.  Never imported by the product
package."""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "_build", "libia_oracle.so")

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")

OP_EQUALIZE, OP_GAUSS, OP_SALT_PEPPER, OP_LINEAR, OP_AVERAGE, OP_MEDIAN, OP_SOBEL, OP_COMPASS, \
    OP_DILATE, OP_ERODE, OP_THRESHOLD, OP_OTSU, OP_QUANTIZE, OP_KMEANS, OP_FEATURES = range(1, 16)


class IaoOp(C.Structure):
    _fields_ = [("type", C.c_int), ("kW", C.c_int), ("kH", C.c_int),
                ("fkernel", C.c_void_p), ("ikernel", C.c_void_p),
                ("iparam", C.c_int), ("mean", C.c_float), ("stddev", C.c_float)]


def build(force=False):
    src = os.path.join(_ROOT, "oracle", "ia_oracle.c")
    if force or not os.path.exists(_SO) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SO)):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle"), "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    L.iao_set_contraction.argtypes = [C.c_int]
    L.iao_set_contraction.restype = None
    L.iao_gray.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int]
    L.iao_gray.restype = C.c_int
    L.iao_hist.argtypes = [u8p, C.c_long, i32p]
    L.iao_eq_map.argtypes = [i32p, i32p, C.c_int]
    L.iao_apply_map.argtypes = [u8p, u8p, C.c_long, i32p]
    L.iao_quantize.argtypes = [u8p, u8p, C.c_long, i32p, C.c_int]
    L.iao_threshold.argtypes = [u8p, u8p, C.c_long, C.c_int]
    L.iao_otsu.argtypes = [u8p, C.c_void_p, C.c_int, C.c_int]
    L.iao_otsu.restype = C.c_int
    L.iao_mean_std.argtypes = [u8p, C.c_long, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.iao_mean_std_from_hist.argtypes = [i64p, C.c_long, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.iao_xorwow_pair.argtypes = [C.c_ulonglong, C.c_uint, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.iao_xorwow_planes.argtypes = [C.c_ulonglong, C.c_long, u32p, u32p]
    L.iao_xorwow_stream.argtypes = [C.c_ulonglong, C.c_uint, C.c_int, u32p]
    L.iao_xorwow_jump_matrix.argtypes = [C.c_int, u32p]
    L.iao_uniform.argtypes = [C.c_uint32]
    L.iao_uniform.restype = C.c_float
    L.iao_salt_pepper.argtypes = [u8p, u8p, C.c_long, u32p, u32p, C.c_int]
    L.iao_gauss_table.argtypes = [C.c_float, C.c_float, u8p, C.c_int]
    L.iao_gauss_noise.argtypes = [u8p, u8p, C.c_long, u32p, C.c_float, C.c_float]
    for name in ("iao_linear", "iao_average"):
        getattr(L, name).argtypes = [u8p, u8p, C.c_int, C.c_int, f32p, C.c_int, C.c_int]
    L.iao_median.argtypes = [u8p, u8p, C.c_int, C.c_int, i32p, C.c_int, C.c_int]
    L.iao_median.restype = C.c_int
    L.iao_sobel.argtypes = [u8p, u8p, C.c_int, C.c_int]
    L.iao_compass.argtypes = [u8p, u8p, C.c_int, C.c_int]
    for name in ("iao_dilate", "iao_dilate_seq", "iao_erode"):
        getattr(L, name).argtypes = [u8p, u8p, C.c_int, C.c_int, i32p, C.c_int, C.c_int]
    L.iao_kmeans2.argtypes = [u8p, u8p, C.c_long, C.c_int, C.c_int]
    L.iao_kmeans2.restype = C.c_int
    L.iao_features.argtypes = [u8p, C.c_int, C.c_int, f32p]
    L.iao_pipeline.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.POINTER(IaoOp), C.c_int,
                               C.c_void_p, C.c_void_p, i64p, C.c_int]
    L.iao_pipeline_ex.argtypes = L.iao_pipeline.argtypes + [C.c_void_p, C.c_void_p, C.c_int]
    L.iao_knn_predict.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, i32p]
    L.iao_knn_crossval.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p]
    L.iao_knn_crossval.restype = C.c_float
    _lib = L
    return L


# ---- numpy-level wrappers (image = 2-D uint8 array [H, W]) -------------------------

def _img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 2
    return a


def gray(rgb, method=0):
    """rgb: [3, H, W] planar."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    _, H, W = rgb.shape
    out = np.empty((H, W), np.uint8)
    rc = lib().iao_gray(rgb.reshape(-1), out.reshape(-1), W, H, method)
    return out if rc == 0 else None


def hist(img):
    img = _img(img)
    h = np.zeros(256, np.int32)
    lib().iao_hist(img.reshape(-1), img.size, h)
    return h


def eq_map(h, n):
    m = np.zeros(256, np.int32)
    lib().iao_eq_map(np.ascontiguousarray(h, np.int32), m, int(n))
    return m


def apply_map(img, m):
    img = _img(img)
    out = np.empty_like(img)
    lib().iao_apply_map(img.reshape(-1), out.reshape(-1), img.size, np.ascontiguousarray(m, np.int32))
    return out


def equalize(img):
    img = _img(img)
    return apply_map(img, eq_map(hist(img), img.size))


def quantize(img, levels):
    img = _img(img)
    lv = np.ascontiguousarray(np.asarray(levels, np.int32).reshape(-1))
    if lv.size == 0:
        lv = np.zeros(3, np.int32)
        n = 0
    else:
        n = lv.size // 3
    out = np.empty_like(img)
    lib().iao_quantize(img.reshape(-1), out.reshape(-1), img.size, lv, n)
    return out


def threshold(img, t):
    img = _img(img)
    out = np.empty_like(img)
    lib().iao_threshold(img.reshape(-1), out.reshape(-1), img.size, int(t))
    return out


def otsu(img):
    img = _img(img)
    out = np.empty_like(img)
    thr = lib().iao_otsu(img.reshape(-1), out.ctypes.data, img.shape[1], img.shape[0])
    return out, thr


def mean_std(img):
    img = _img(img)
    m, s = C.c_float(), C.c_float()
    lib().iao_mean_std(img.reshape(-1), img.size, C.byref(m), C.byref(s))
    return np.float32(m.value), np.float32(s.value)


def xorwow_planes(n, seed=1234):
    x0 = np.empty(n, np.uint32)
    x1 = np.empty(n, np.uint32)
    lib().iao_xorwow_planes(seed, n, x0, x1)
    return x0, x1


def xorwow_stream(seed, subsequence, n):
    out = np.empty(n, np.uint32)
    lib().iao_xorwow_stream(seed, subsequence, n, out)
    return out


def jump_matrix(log2steps):
    m = np.empty(800, np.uint32)
    lib().iao_xorwow_jump_matrix(log2steps, m)
    return m


def salt_pepper(img, x0, x1, level):
    img = _img(img)
    out = np.empty_like(img)
    lib().iao_salt_pepper(img.reshape(-1), out.reshape(-1), img.size, x0, x1, int(level))
    return out


def gauss_table(mean, variance, use_libm=False):
    t = np.empty(256, np.uint8)
    lib().iao_gauss_table(float(mean), float(variance), t, int(use_libm))
    return t


def gauss_noise(img, x0, mean=-1.0, stddev=-1.0):
    img = _img(img)
    out = np.empty_like(img)
    lib().iao_gauss_noise(img.reshape(-1), out.reshape(-1), img.size, x0, float(mean), float(stddev))
    return out


def _stencil_f(fn, img, kernel, kW, kH):
    img = _img(img)
    out = np.empty_like(img)
    k = np.ascontiguousarray(np.asarray(kernel, np.float32).reshape(-1))
    getattr(lib(), fn)(img.reshape(-1), out.reshape(-1), img.shape[1], img.shape[0], k, kW, kH)
    return out


def linear(img, kernel, kW, kH):
    return _stencil_f("iao_linear", img, kernel, kW, kH)


def average(img, kernel, kW, kH):
    return _stencil_f("iao_average", img, kernel, kW, kH)


def _stencil_i(fn, img, kernel, kW, kH):
    img = _img(img)
    out = np.empty_like(img)
    k = np.ascontiguousarray(np.asarray(kernel, np.int32).reshape(-1))
    rc = getattr(lib(), fn)(img.reshape(-1), out.reshape(-1), img.shape[1], img.shape[0], k, kW, kH)
    return out, rc


def median(img, weights, kW, kH):
    out, rc = _stencil_i("iao_median", img, weights, kW, kH)
    return out if rc == 0 else None


def dilate(img, se, kW, kH):
    return _stencil_i("iao_dilate", img, se, kW, kH)[0]


def dilate_seq(img, se, kW, kH):
    return _stencil_i("iao_dilate_seq", img, se, kW, kH)[0]


def erode(img, se, kW, kH):
    return _stencil_i("iao_erode", img, se, kW, kH)[0]


def sobel(img):
    img = _img(img)
    out = np.empty_like(img)
    lib().iao_sobel(img.reshape(-1), out.reshape(-1), img.shape[1], img.shape[0])
    return out


def compass(img):
    img = _img(img)
    out = np.empty_like(img)
    lib().iao_compass(img.reshape(-1), out.reshape(-1), img.shape[1], img.shape[0])
    return out


def kmeans2(img, c0, c1):
    img = _img(img)
    out = np.empty_like(img)
    it = lib().iao_kmeans2(img.reshape(-1), out.reshape(-1), img.size, int(c0), int(c1))
    return out, it


def features(img):
    img = _img(img)
    f = np.zeros(7, np.float32)
    lib().iao_features(img.reshape(-1), img.shape[1], img.shape[0], f)
    return f


def make_ops(ops):
    """ops: list of dicts {type, kW, kH, fkernel, ikernel, iparam, mean, stddev}.  Returns
    (ctypes array, keepalive list)."""
    arr = (IaoOp * max(len(ops), 1))()
    keep = []
    for i, o in enumerate(ops):
        arr[i].type = o["type"]
        arr[i].kW = o.get("kW", 0)
        arr[i].kH = o.get("kH", 0)
        if o.get("fkernel") is not None:
            k = np.ascontiguousarray(np.asarray(o["fkernel"], np.float32).reshape(-1))
            keep.append(k)
            arr[i].fkernel = k.ctypes.data
        if o.get("ikernel") is not None:
            k = np.ascontiguousarray(np.asarray(o["ikernel"], np.int32).reshape(-1))
            keep.append(k)
            arr[i].ikernel = k.ctypes.data
        arr[i].iparam = o.get("iparam", 0)
        arr[i].mean = o.get("mean", -1.0)
        arr[i].stddev = o.get("stddev", -1.0)
    return arr, keep


def knn_predict(train, test, k):
    train = np.ascontiguousarray(train, np.float32)
    test = np.ascontiguousarray(test, np.float32)
    pred = np.zeros(test.shape[0], np.int32)
    lib().iao_knn_predict(train.reshape(-1), test.reshape(-1), train.shape[0], test.shape[0], train.shape[1], int(k), pred)
    return pred


def knn_crossval(dataset, k):
    """Returns (accuracy, precision, recall, per-fold accuracies)."""
    d = np.ascontiguousarray(dataset, np.float32)
    p, r, folds = np.zeros(1, np.float32), np.zeros(1, np.float32), np.zeros(10, np.float32)
    acc = lib().iao_knn_crossval(d.reshape(-1), d.shape[0], d.shape[1], int(k), p, r, folds)
    return float(acc), float(p[0]), float(r[0]), folds


_libc = None


def libc_random_0_256(n, skip=0):
    """The first skip+n values of the reference's random(0, 256) (imageAnalysis.cu:39-41) in a
    program that never calls srand, from the C library itself: srand(1), then
    rand() / (RAND_MAX / 257 + 1).  Returns the last n."""
    global _libc
    if _libc is None:
        _libc = C.CDLL("libc.so.6")
        _libc.rand.restype = C.c_int
    _libc.srand(1)
    vals = [_libc.rand() // (2147483647 // 257 + 1) for _ in range(skip + n)]
    return np.asarray(vals[skip:], np.int32)


def pipeline_ex(rgb, ops, x0=None, x1=None, threads=1, centroids=None, input=0):
    """As pipeline(); ops may hold OP_KMEANS (centroids: int32 [B, nk, 2], the rand() draws in
    the reference's order) and OP_FEATURES.  Returns (out, hist, features [B, 7])."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    B, _, H, W = rgb.shape
    out = np.empty((B, H, W), np.uint8)
    hs = np.zeros(256, np.int64)
    feats = np.zeros((B, 7), np.float32)
    arr, keep = make_ops(ops)
    cen = np.ascontiguousarray(centroids, np.int32) if centroids is not None else None
    lib().iao_pipeline_ex(rgb.reshape(-1), out.reshape(-1), W, H, B, arr, len(ops),
                          x0.ctypes.data if x0 is not None else None,
                          x1.ctypes.data if x1 is not None else None, hs, threads,
                          cen.ctypes.data if cen is not None else None, feats.ctypes.data, input)
    del keep
    return out, hs, feats


def pipeline(rgb, ops, x0=None, x1=None, threads=1):
    """rgb: [B, 3, H, W].  Returns (out [B, H, W], dataset histogram int64[256])."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    B, _, H, W = rgb.shape
    out = np.empty((B, H, W), np.uint8)
    hs = np.zeros(256, np.int64)
    arr, keep = make_ops(ops)
    lib().iao_pipeline(rgb.reshape(-1), out.reshape(-1), W, H, B, arr, len(ops),
                       x0.ctypes.data if x0 is not None else None,
                       x1.ctypes.data if x1 is not None else None, hs, threads)
    del keep
    return out, hs


class contraction:
    """with contraction(): ... -- the oracle in contraction mode (the luminance and the filter accumulation as
    fused multiply-adds, as nvcc's default build of the reference evaluates them); back to separate roundings
    afterwards."""

    def __enter__(self):
        lib().iao_set_contraction(1)
        return self

    def __exit__(self, *exc):
        lib().iao_set_contraction(0)
        return False
