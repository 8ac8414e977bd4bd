"""Batched pipeline over the C ABI's executor (iak_pipeline_*): the device-resident form of
the reference driver's per-image operation loop (test/test.cu:266-564)."""
import ctypes as C

import numpy as np
import torch

from . import lib as _l

_NAMES = {
    "histogram-equalization": _l.IAK_OP_EQUALIZE, "gaussian-noise": _l.IAK_OP_GAUSS,
    "salt-and-pepper": _l.IAK_OP_SALT_PEPPER, "linear-filter": _l.IAK_OP_LINEAR,
    "average-filter": _l.IAK_OP_AVERAGE, "median-filter": _l.IAK_OP_MEDIAN, "sobel-filter": _l.IAK_OP_SOBEL,
    "compass-filter": _l.IAK_OP_COMPASS, "dilation": _l.IAK_OP_DILATE, "erosion": _l.IAK_OP_ERODE,
    "threshold": _l.IAK_OP_THRESHOLD, "otsu-thresh": _l.IAK_OP_OTSU, "quantization": _l.IAK_OP_QUANTIZE,
    "kMeans-thresh": _l.IAK_OP_KMEANS, "feature-extract": _l.IAK_OP_FEATURES,
}


def ops_from_config(operations):
    """Translate the `operations` list of a reference JSON config (test/configs/*.json, already
    parsed) into executor op dicts.  Unknown types raise, as nothing else could run them."""
    out = []
    for o in operations:
        t = o["type"]
        if t not in _NAMES:
            raise ValueError(f"Unsupported Operation: {t}")
        d = {"type": _NAMES[t]}
        if t in ("linear-filter", "average-filter"):
            k = o["kernel"]
            d.update(kW=k["width"], kH=k["height"], fkernel=[float(v) for v in k["values"]])
        elif t == "median-filter":
            k = o["kernel"]
            d.update(kW=k["width"], kH=k["height"], ikernel=[int(v) for v in k["values"]])
        elif t in ("dilation", "erosion"):
            k = o["structuring-element"]
            d.update(kW=k["width"], kH=k["height"], ikernel=[int(v) for v in k["values"]])
        elif t == "gaussian-noise":
            d.update(mean=float(o["mean"]), stddev=float(o["std_dev"]))
        elif t == "salt-and-pepper":
            d.update(iparam=int(o["intensity"]))
        elif t == "threshold":
            d.update(iparam=int(o["threshold"]))
        elif t == "quantization":
            lv = []
            for l in o["levels"]:
                lv += [int(l["min"]), int(l["max"]), int(l["val"])]
            d.update(ikernel=lv, iparam=len(o["levels"]))
        out.append(d)
    return out


class Pipeline:
    """Owns an iak_pipeline.  ops: list of dicts with keys type, kW, kH, fkernel, ikernel,
    iparam, mean, stddev (same meaning as iak_op).  fused: True (per-pixel runs fused, and fed
    straight into the 3x3 stencil behind them), "pointwise" (per-pixel runs fused only) or False (one
    pass per operation)."""

    def __init__(self, W, H, maxB, ops, seed=1234, fused=True, device=None):
        self.W, self.H, self.maxB = int(W), int(H), int(maxB)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        arr = (_l.IakOp * max(len(ops), 1))()
        keep = []
        for i, o in enumerate(ops):
            arr[i].type = o["type"]
            arr[i].kW = o.get("kW", 0)
            arr[i].kH = o.get("kH", 0)
            if o.get("fkernel") is not None:
                k = np.ascontiguousarray(np.asarray(o["fkernel"], np.float32).reshape(-1))
                keep.append(k)
                arr[i].fkernel = k.ctypes.data
            if o.get("ikernel") is not None and len(o["ikernel"]):
                k = np.ascontiguousarray(np.asarray(o["ikernel"], np.int32).reshape(-1))
                keep.append(k)
                arr[i].ikernel = k.ctypes.data
            arr[i].iparam = o.get("iparam", 0)
            arr[i].mean = o.get("mean", -1.0)
            arr[i].stddev = o.get("stddev", -1.0)
        self._h = C.c_void_p()
        L = _l.load()
        mode = (_l.IAK_PIPE_FUSED_POINTWISE if fused == "pointwise" else
                _l.IAK_PIPE_FUSED if fused else _l.IAK_PIPE_STAGEWISE)
        _l.check(L.iak_pipeline_create(C.byref(self._h), self.W, self.H, self.maxB, arr, len(ops), seed, mode),
                 "iak_pipeline_create")
        del keep  # the executor copied every host array

    def close(self):
        if self._h:
            _l.load().iak_pipeline_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, rgb, out=None, hist_sum=None, gray_hist=None):
        """rgb: [B,3,H,W] uint8 CUDA contiguous.  hist_sum: int64[256] CUDA tensor the batch's
        grayscale histogram is added to.  Enqueues on torch's current stream."""
        assert rgb.is_cuda and rgb.dtype == torch.uint8 and rgb.is_contiguous()
        B = rgb.shape[0]
        assert rgb.shape[1:] == (3, self.H, self.W) and B <= self.maxB
        if out is None:
            out = torch.empty((B, self.H, self.W), dtype=torch.uint8, device=rgb.device)
        assert out.is_contiguous() and out.shape == (B, self.H, self.W)
        if hist_sum is not None:
            assert hist_sum.dtype == torch.int64 and hist_sum.numel() == 256 and hist_sum.is_contiguous()
        if gray_hist is not None:
            assert gray_hist.dtype == torch.int32 and gray_hist.shape == (B, 256) and gray_hist.is_contiguous()
        _l.check(_l.load().iak_pipeline_run(
            self._h, rgb.data_ptr(), out.data_ptr(), B, hist_sum.data_ptr() if hist_sum is not None else None,
            gray_hist.data_ptr() if gray_hist is not None else None, torch.cuda.current_stream().cuda_stream),
            "iak_pipeline_run")
        return out

    def bind_features(self, features):
        """features: float32 [maxB, 7] CUDA tensor the feature-extract operation writes."""
        assert features.is_cuda and features.dtype == torch.float32 and features.is_contiguous()
        assert features.shape == (self.maxB, _l.IAK_N_FEATURES)
        self._features = features  # keep it alive
        _l.check(_l.load().iak_pipeline_bind_features(self._h, features.data_ptr()), "iak_pipeline_bind_features")

    def set_input(self, what):
        """What the first pass makes of the RGB input: "grey" (luminance, default), "average",
        "red", "green" or "blue" (the driver's extract_channel)."""
        code = {"grey": 0, "gray": 0, "average": 1, "red": 2, "green": 3, "blue": 4}[what]
        _l.check(_l.load().iak_pipeline_set_input(self._h, code), "iak_pipeline_set_input")

    def set_lanes(self, lanes):
        """Run every batch as `lanes` concurrent sub-batches (see iak_pipeline_set_lanes)."""
        _l.check(_l.load().iak_pipeline_set_lanes(self._h, int(lanes)), "iak_pipeline_set_lanes")

    def timed_images(self):
        return _l.load().iak_pipeline_timed_images(self._h)

    def set_timing(self, slots, every=1):
        """Record per-group HIP events for the next runs (every `every`-th of them), into `slots`
        rotating sets."""
        _l.check(_l.load().iak_pipeline_set_timing_every(self._h, int(every)), "iak_pipeline_set_timing_every")
        _l.check(_l.load().iak_pipeline_set_timing(self._h, int(slots)), "iak_pipeline_set_timing")

    def groups(self):
        """[(name, algorithmic bytes per pixel)] of the planned launch groups."""
        cap = 64
        names = (C.c_char_p * cap)()
        ab = (C.c_longlong * cap)()
        n = _l.load().iak_pipeline_timings(self._h, 0, None, None, names, ab, cap)
        if n < 0:
            raise _l.IakError(n, "iak_pipeline_timings")
        return [(names[i].decode(), int(ab[i])) for i in range(n)]

    def moved_bytes(self):
        """Bytes per pixel each launch group moves through memory as executed (see
        iak_pipeline_group_bytes)."""
        cap = 64
        mb = (C.c_longlong * cap)()
        n = _l.load().iak_pipeline_group_bytes(self._h, mb, cap)
        if n < 0:
            raise _l.IakError(n, "iak_pipeline_group_bytes")
        return [int(mb[i]) for i in range(n)]

    def timings(self, slot=0):
        """[(name, ms, algorithmic bytes per pixel, ms of the group's image kernel alone)] of the
        timed run in `slot` (synchronises)."""
        cap = 64
        ms = (C.c_float * cap)()
        main = (C.c_float * cap)()
        names = (C.c_char_p * cap)()
        ab = (C.c_longlong * cap)()
        n = _l.load().iak_pipeline_timings(self._h, int(slot), ms, main, names, ab, cap)
        if n < 0:
            raise _l.IakError(n, "iak_pipeline_timings")
        return [(names[i].decode(), float(ms[i]), int(ab[i]), float(main[i])) for i in range(n)]
