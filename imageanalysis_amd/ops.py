"""Torch-tensor conveniences over the C ABI, mirroring the reference's operator names
(imageAnalysisLib/imageAnalysis.h:46-111) for batches.  Tensors are CUDA uint8,
images [B, H, W] (or [H, W]), RGB [B, 3, H, W] planar.  Each function allocates its output
(as the reference's wrappers do) and enqueues on torch's current stream."""
import ctypes as C

import numpy as np
import torch

from . import lib as _l


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _b3(img):
    if img.dim() == 2:
        img = img.unsqueeze(0)
    assert img.dim() == 3 and img.dtype == torch.uint8 and img.is_cuda
    assert img.stride(2) == 1 and img.stride(1) == img.shape[2], "rows must be dense"
    return img


def _like(img, squeeze):
    out = torch.empty_like(img, memory_format=torch.contiguous_format)
    return out


def _ret(out, ref):
    return out[0] if ref.dim() == 2 else out


def _istride(t):
    return t.stride(0) if t.shape[0] > 1 else t.shape[1] * t.shape[2]


def convertRGBToGrayscale(rgb, method=0, with_hist=False):
    single = rgb.dim() == 3
    if single:
        rgb = rgb.unsqueeze(0)
    assert rgb.dim() == 4 and rgb.shape[1] == 3 and rgb.dtype == torch.uint8 and rgb.is_contiguous()
    B, _, H, W = rgb.shape
    gray = torch.empty((B, H, W), dtype=torch.uint8, device=rgb.device)
    L = _l.load()
    if with_hist:
        hist = torch.empty((B, 256), dtype=torch.int32, device=rgb.device)
        _l.check(L.iak_gray_hist(rgb.data_ptr(), gray.data_ptr(), hist.data_ptr(), W, H, B, 3 * H * W, H * W,
                                 method, _stream()), "iak_gray_hist")
        return (gray[0], hist[0]) if single else (gray, hist)
    _l.check(L.iak_gray(rgb.data_ptr(), gray.data_ptr(), W, H, B, 3 * H * W, H * W, method, _stream()), "iak_gray")
    return gray[0] if single else gray


def extractSingleColorChannel(rgb, color):
    """Pointer aliasing in the reference (imageAnalysis.cu:72-92): a view, no copy."""
    if color not in (0, 1, 2):
        raise ValueError("invalid option")
    return rgb[..., color, :, :]


def calculateHistogram(img):
    x = _b3(img)
    B, H, W = x.shape
    hist = torch.empty((B, 256), dtype=torch.int32, device=x.device)
    _l.check(_l.load().iak_hist(x.data_ptr(), hist.data_ptr(), H * W, B, _istride(x), _stream()), "iak_hist")
    return _ret(hist, img)


def equalizeHistogram(hist, numPixels):
    h = hist if hist.dim() == 2 else hist.unsqueeze(0)
    h = h.contiguous()
    m = torch.empty_like(h)
    _l.check(_l.load().iak_eq_map(h.data_ptr(), m.data_ptr(), int(numPixels), h.shape[0], _stream()), "iak_eq_map")
    return m if hist.dim() == 2 else m[0]


def equalizeImageWithHist(img, mappings):
    x = _b3(img)
    B, H, W = x.shape
    m = mappings.contiguous()
    out = _like(x, False)
    stride = 256 if m.dim() == 2 and m.shape[0] == B and B > 1 else (256 if m.dim() == 2 else 0)
    if m.dim() == 1:
        stride = 0
    _l.check(_l.load().iak_apply_map(x.data_ptr(), out.data_ptr(), H * W, B, _istride(x), H * W, m.data_ptr(), stride,
                                     _stream()), "iak_apply_map")
    return _ret(out, img)


def histogramEqualization(img):
    """test/test.cu:404-419: histogram -> map -> apply."""
    x = _b3(img)
    h = calculateHistogram(x)
    m = equalizeHistogram(h, x.shape[1] * x.shape[2])
    return _ret(equalizeImageWithHist(x, m), img)


def imageQuantization(img, levels):
    x = _b3(img)
    B, H, W = x.shape
    lv = np.ascontiguousarray(np.asarray(levels, np.int32).reshape(-1))
    n = lv.size // 3
    out = _like(x, False)
    ptr = lv.ctypes.data if n else None
    _l.check(_l.load().iak_quantize(x.data_ptr(), out.data_ptr(), H * W, B, _istride(x), H * W, ptr, n, _stream()),
             "iak_quantize")
    return _ret(out, img)


def thresholdImage(img, threshold):
    x = _b3(img)
    B, H, W = x.shape
    out = _like(x, False)
    _l.check(_l.load().iak_threshold(x.data_ptr(), out.data_ptr(), H * W, B, _istride(x), H * W, int(threshold),
                                     _stream()), "iak_threshold")
    return _ret(out, img)


def otsuThresholdImage(img, return_threshold=False):
    x = _b3(img)
    B, H, W = x.shape
    h = calculateHistogram(x)
    thr = torch.empty((B,), dtype=torch.int32, device=x.device)
    L = _l.load()
    _l.check(L.iak_otsu_threshold(h.data_ptr(), thr.data_ptr(), H * W, B, _stream()), "iak_otsu_threshold")
    out = _like(x, False)
    _l.check(L.iak_threshold_per_image(x.data_ptr(), out.data_ptr(), H * W, B, _istride(x), H * W, thr.data_ptr(),
                                       _stream()), "iak_threshold_per_image")
    if return_threshold:
        return _ret(out, img), thr
    return _ret(out, img)


def calculateMeanAndStdDev(img):
    x = _b3(img)
    B, H, W = x.shape
    h = calculateHistogram(x)
    ms = torch.empty((B, 2), dtype=torch.float32, device=x.device)
    _l.check(_l.load().iak_mean_std(h.data_ptr(), ms.data_ptr(), H * W, B, _stream()), "iak_mean_std")
    return ms if img.dim() == 3 else ms[0]


class Randomness:
    """setupRandomness (imageAnalysis.cu:740-746): per-pixel generator outputs for images of
    n pixels, kept on the device; derived planes are cached."""

    def __init__(self, n, device="cuda", seed=1234):
        self.n = int(n)
        self.x0 = torch.empty((self.n,), dtype=torch.int32, device=device)
        self.x1 = torch.empty((self.n,), dtype=torch.int32, device=device)
        _l.check(_l.load().iak_rng_planes(self.x0.data_ptr(), self.x1.data_ptr(), self.n, seed, _stream()),
                 "iak_rng_planes")
        self._rv = None
        self._sp = {}

    def rv(self):
        if self._rv is None:
            self._rv = torch.empty((self.n,), dtype=torch.uint8, device=self.x0.device)
            _l.check(_l.load().iak_rv_plane(self.x0.data_ptr(), self._rv.data_ptr(), self.n, _stream()), "iak_rv_plane")
        return self._rv

    def sp(self, level):
        level = int(level)
        if level not in self._sp:
            p = torch.empty((self.n,), dtype=torch.uint8, device=self.x0.device)
            _l.check(_l.load().iak_sp_plane(self.x0.data_ptr(), self.x1.data_ptr(), p.data_ptr(), self.n, level,
                                            _stream()), "iak_sp_plane")
            self._sp[level] = p
        return self._sp[level]


def saltAndPepperNoise(img, rng, level):
    x = _b3(img)
    B, H, W = x.shape
    assert rng.n >= H * W
    out = _like(x, False)
    _l.check(_l.load().iak_salt_pepper(x.data_ptr(), out.data_ptr(), H * W, B, _istride(x), H * W,
                                       rng.sp(level).data_ptr(), _stream()), "iak_salt_pepper")
    return _ret(out, img)


def addGaussianNoise(img, rng, mean=-1.0, stdDev=-1.0, hist=None):
    x = _b3(img)
    B, H, W = x.shape
    assert rng.n >= H * W
    L = _l.load()
    from_image = (mean == -1 or stdDev == -1)
    if from_image and hist is None:
        hist = calculateHistogram(x)
    nt = B if from_image else 1
    table = torch.empty((nt, 256), dtype=torch.int16, device=x.device)
    _l.check(L.iak_gauss_table(hist.data_ptr() if hist is not None else None, H * W, nt, float(mean), float(stdDev),
                               table.data_ptr(), _stream()), "iak_gauss_table")
    out = _like(x, False)
    _l.check(L.iak_gauss_noise(x.data_ptr(), out.data_ptr(), H * W, B, _istride(x), H * W, rng.rv().data_ptr(),
                               table.data_ptr(), 256 if from_image else 0, _stream()), "iak_gauss_noise")
    return _ret(out, img)


def _stencil(name, img, host_arr, kW, kH, *extra):
    x = _b3(img)
    B, H, W = x.shape
    out = _like(x, False)
    fn = getattr(_l.load(), name)
    args = [x.data_ptr(), out.data_ptr(), W, H, B, _istride(x), H * W]
    if host_arr is not None:
        args += [host_arr.ctypes.data, kW, kH]
    args += list(extra) + [_stream()]
    _l.check(fn(*args), name)
    return _ret(out, img)


def linearFilter(img, kernel, kW, kH):
    k = np.ascontiguousarray(np.asarray(kernel, np.float32).reshape(-1))
    return _stencil("iak_linear", img, k, kW, kH, _l.IAK_LINEAR_CLAMP)


def averageFilter(img, kernel, kW, kH):
    k = np.ascontiguousarray(np.asarray(kernel, np.float32).reshape(-1))
    return _stencil("iak_linear", img, k, kW, kH, _l.IAK_LINEAR_AVERAGE)


def medianFilter(img, weights, kW, kH):
    k = np.ascontiguousarray(np.asarray(weights, np.int32).reshape(-1))
    return _stencil("iak_median", img, k, kW, kH)


def sobelFilter(img):
    return _stencil("iak_sobel", img, None, 0, 0)


def compassFilter(img):
    return _stencil("iak_compass", img, None, 0, 0)


def imageDilation(img, se, kW, kH):
    k = np.ascontiguousarray(np.asarray(se, np.int32).reshape(-1))
    return _stencil("iak_dilate", img, k, kW, kH)


def imageErosion(img, se, kW, kH):
    k = np.ascontiguousarray(np.asarray(se, np.int32).reshape(-1))
    return _stencil("iak_erode", img, k, kW, kH)


class LibcRand:
    """The libc rand() stream the reference's k-means draws its centroids from
    (imageAnalysis.cu:39-41,359-361), reproduced on the device: state after srand(seed);
    a program that never calls srand behaves as srand(1)."""

    def __init__(self, seed=1, device=None):
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.state = torch.zeros(_l.IAK_GLIBC_RAND_WORDS, dtype=torch.int32, device=dev)
        _l.check(_l.load().iak_glibc_rand_seed(self.state.data_ptr(), int(seed), _stream()), "iak_glibc_rand_seed")

    def centroids(self, images, ops_per_image=1):
        """The next images*ops_per_image pairs of random(0,256), as int32 [ops_per_image, images, 2]."""
        out = torch.empty((ops_per_image, images, 2), dtype=torch.int32, device=self.state.device)
        _l.check(_l.load().iak_glibc_rand_centroids(self.state.data_ptr(), images, ops_per_image, out.data_ptr(),
                                                    _stream()), "iak_glibc_rand_centroids")
        return out


def kMeansThresholding(img, init_centroids, return_iterations=False):
    """imageAnalysis.cu:351-379 for a batch: init_centroids int32 [B, 2] (CUDA), e.g.
    LibcRand().centroids(B)[0]."""
    x = _b3(img)
    B, H, W = x.shape
    hist = calculateHistogram(x)
    ini = init_centroids.contiguous().view(-1, 2)
    assert ini.shape[0] == B and ini.dtype == torch.int32 and ini.is_cuda
    m = torch.empty((B, 256), dtype=torch.int32, device=x.device)
    it = torch.empty((B,), dtype=torch.int32, device=x.device)
    _l.check(_l.load().iak_kmeans2_map(hist.data_ptr(), H * W, B, ini.data_ptr(), m.data_ptr(), it.data_ptr(), _stream()),
             "iak_kmeans2_map")
    out = equalizeImageWithHist(x, m)
    out = _ret(out, img)
    return (out, it) if return_iterations else out


def featureExtraction(img, hist=None):
    """imageMachineLearning.cu:393-415 for a batch: float32 [B, 7] = mean, std, area, perimeter,
    mode, rangeMin, rangeMax."""
    x = _b3(img)
    B, H, W = x.shape
    L = _l.load()
    nbytes = L.iak_features_workspace_bytes(W, H, B)
    if nbytes < 0:
        raise _l.IakError(int(nbytes), "iak_features_workspace_bytes")
    ws = torch.empty((nbytes + 256,), dtype=torch.uint8, device=x.device)
    base = (ws.data_ptr() + 255) & ~255
    feats = torch.empty((B, _l.IAK_N_FEATURES), dtype=torch.float32, device=x.device)
    h = hist.contiguous() if hist is not None else None
    _l.check(L.iak_features(x.data_ptr(), W, H, B, _istride(x), h.data_ptr() if h is not None else None, feats.data_ptr(),
                            base, _stream()), "iak_features")
    torch.cuda.current_stream().synchronize()  # ws is freed on return
    return feats if img.dim() == 3 else feats[0]


def knn(train, test, k):
    """imageMachineLearning.cu:158-197 on CUDA float32 tensors [n, numAttributes] (last column =
    class).  Returns int32 predictions [numTest]."""
    tr, te = train.contiguous(), test.contiguous()
    assert tr.dtype == torch.float32 and te.dtype == torch.float32 and tr.shape[1] == te.shape[1]
    pred = torch.empty((te.shape[0],), dtype=torch.int32, device=tr.device)
    _l.check(_l.load().iak_knn_predict(tr.data_ptr(), te.data_ptr(), tr.shape[0], te.shape[0], tr.shape[1], int(k),
                                       pred.data_ptr(), _stream()), "iak_knn_predict")
    return pred
