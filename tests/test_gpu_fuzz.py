"""Differential fuzzing of the executor against the oracle: random image shapes (down to one
row or one column), random operation lists with random parameters, every planning mode.
Seeds are fixed, so a failure names its case."""
import numpy as np
import pytest
import torch

import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu

MODES = [False, "pointwise", True]


def random_op(rs):
    t = int(rs.choice([O.OP_EQUALIZE, O.OP_GAUSS, O.OP_SALT_PEPPER, O.OP_LINEAR, O.OP_AVERAGE, O.OP_MEDIAN, O.OP_SOBEL,
                       O.OP_COMPASS, O.OP_DILATE, O.OP_ERODE, O.OP_THRESHOLD, O.OP_OTSU, O.OP_QUANTIZE, O.OP_KMEANS,
                       O.OP_FEATURES]))
    op = {"type": t}
    if t == O.OP_GAUSS:
        if rs.random() < 0.5:
            op.update(mean=-1.0, stddev=-1.0)
        else:
            op.update(mean=float(rs.integers(0, 256)), stddev=float(rs.choice([0.0, 1.0, 7.5, 40.0])))
    elif t == O.OP_SALT_PEPPER:
        op["iparam"] = int(rs.choice([0, 1, 5, 50, 255]))
    elif t in (O.OP_LINEAR, O.OP_AVERAGE):
        k = int(rs.choice([3, 3, 3, 5, 1, 2, 7, 9]))
        kind = rs.integers(0, 5) if k <= 5 else rs.integers(3, 5)
        if kind == 0:
            w = rs.integers(-3, 12, k * k).astype(np.float32)
        elif kind == 1:
            w = (rs.integers(0, 9, k * k) / 16.0).astype(np.float32)
        elif kind == 2:
            w = ((rs.random(k * k) - 0.3) * 1.5).astype(np.float32)
        elif kind == 3:
            w = np.full(k * k, 1.0 / (k * k), np.float32)
        else:
            w = np.full(k * k, float(rs.choice([1.0, 0.5, 0.03])), np.float32)
        op.update(kW=k, kH=k, fkernel=[float(v) for v in w])
    elif t == O.OP_MEDIAN:
        kw, kh = (3, 3) if rs.random() < 0.6 else (int(rs.choice([1, 3, 5, 7])), int(rs.choice([1, 3, 5])))
        w = [1] * (kw * kh) if rs.random() < 0.7 else [int(v) for v in rs.integers(0, 4, kw * kh)]
        if sum(w) == 0:
            w[0] = 1
        op.update(kW=kw, kH=kh, ikernel=w)
    elif t in (O.OP_DILATE, O.OP_ERODE):
        kw, kh = (3, 3) if rs.random() < 0.6 else (int(rs.choice([1, 3, 5, 7, 9])), int(rs.choice([1, 3, 5, 7, 9])))
        se = [int(v) for v in rs.integers(0, 2, kw * kh)] if rs.random() < 0.6 else [1] * (kw * kh)
        op.update(kW=kw, kH=kh, ikernel=se)
    elif t == O.OP_THRESHOLD:
        op["iparam"] = int(rs.integers(-1, 257))
    elif t == O.OP_QUANTIZE:
        n = int(rs.integers(0, 4))
        lv = []
        for _ in range(n):
            lo = int(rs.integers(0, 250))
            lv += [lo, lo + int(rs.integers(1, 100)), int(rs.integers(0, 256))]
        op.update(ikernel=lv, iparam=n)
    return op


@pytest.mark.parametrize("case", range(300))
def test_fuzz_pipeline(case):
    from imageanalysis_amd.pipeline import Pipeline
    rs = np.random.default_rng(9000 + case)
    if rs.random() < 0.5:
        W = int(rs.choice([16, 32, 48, 64, 96, 128, 256]))
    else:
        W = int(rs.integers(1, 90))
    H = int(rs.integers(1, 45))
    B = int(rs.integers(1, 4))
    kind = str(rs.choice(["C", "U", "K"]))
    ops = [random_op(rs) for _ in range(int(rs.integers(0, 8)))]
    nk = sum(1 for o in ops if o["type"] == O.OP_KMEANS)
    rgb = synth.rgb_batch(kind, W, H, B, first_index=case)
    if rs.random() < 0.3:
        rgb[0, :, : H // 2] = 0
    x0, x1 = O.xorwow_planes(W * H)
    cent = O.libc_random_0_256(2 * nk * B).reshape(B, nk, 2) if nk else None
    want, want_hist, want_feats = O.pipeline_ex(rgb, ops, x0, x1, centroids=cent)
    has_features = any(o["type"] == O.OP_FEATURES for o in ops)
    for mode in MODES:
        p = Pipeline(W, H, B, ops, fused=mode)
        feats = torch.zeros((B, 7), dtype=torch.float32, device="cuda")
        p.bind_features(feats)
        hs = torch.zeros(256, dtype=torch.int64, device="cuda")
        got = p.run(torch.from_numpy(rgb).cuda(), hist_sum=hs).cpu().numpy()
        names = [n for n, _ in p.groups()]
        if not np.array_equal(got, want):
            bad = np.argwhere(got != want)
            raise AssertionError(f"case {case} mode {mode} {W}x{H}x{B} {kind} groups {names}: {len(bad)} mismatches, "
                                 f"first {bad[0]}, ops {ops}")
        assert np.array_equal(hs.cpu().numpy(), want_hist), (case, mode)
        if has_features:
            assert np.array_equal(feats.cpu().numpy(), want_feats, equal_nan=True), (case, mode, names)  # 1x1 images: std is NaN
        p.close()


def _placed(imgs, rs):
    """The batch copied into a larger device buffer at a random byte offset with a random image
    stride, as a [B, H, W] view: exercises the unaligned / strided paths of every entry point."""
    B, H, W = imgs.shape
    N = H * W
    off = int(rs.choice([0, 0, 1, 3, 4, 8, 16, 20]))
    stride = N + int(rs.choice([0, 0, 1, 5, 16, 64]))
    buf = torch.zeros(off + stride * B + 64, dtype=torch.uint8, device="cuda")
    view = torch.as_strided(buf[off:], (B, H, W), (stride, W, 1))
    view.copy_(torch.from_numpy(imgs).cuda())
    return view


@pytest.mark.parametrize("case", range(260))
def test_fuzz_entry_points(case):
    from imageanalysis_amd import ops
    rs = np.random.default_rng(5000 + case)
    W = int(rs.choice([8, 16, 24, 32, 64, 128, 200])) if rs.random() < 0.6 else int(rs.integers(1, 70))
    H = int(rs.integers(1, 40))
    B = int(rs.integers(1, 4))
    kind = str(rs.choice(["C", "U", "K"]))
    imgs = synth.gray_batch(kind, W, H, B, first_index=case)
    if rs.random() < 0.3:
        imgs[0, :, : W // 2] = 0
    x = _placed(imgs, rs)
    N = W * H
    x0, x1 = O.xorwow_planes(N)
    rng = ops.Randomness(N)

    def check(got, fn, what):
        got = got.cpu().numpy()
        for b in range(B):
            want = fn(imgs[b])
            want = want[0] if isinstance(want, tuple) else want
            assert np.array_equal(got[b], want), f"case {case} {what} {W}x{H} image {b} offset {x.data_ptr() % 16} stride {x.stride(0)}"

    which = int(rs.integers(0, 13))
    if which == 0:
        check(ops.calculateHistogram(x), O.hist, "hist")
        check(ops.histogramEqualization(x), O.equalize, "equalize")
    elif which == 1:
        t = int(rs.integers(-1, 257))
        check(ops.thresholdImage(x, t), lambda g: O.threshold(g, t), "threshold")
        check(ops.otsuThresholdImage(x), O.otsu, "otsu")
    elif which == 2:
        lv = [10, 20, 15, 30, 250, 77]
        check(ops.imageQuantization(x, lv), lambda g: O.quantize(g, lv), "quantize")
    elif which == 3:
        level = int(rs.choice([0, 1, 50, 255]))
        check(ops.saltAndPepperNoise(x, rng, level), lambda g: O.salt_pepper(g, x0, x1, level), "salt_pepper")
    elif which == 4:
        m, s = (-1.0, -1.0) if rs.random() < 0.5 else (float(rs.integers(0, 256)), float(rs.choice([0.0, 3.0, 50.0])))
        check(ops.addGaussianNoise(x, rng, m, s), lambda g: O.gauss_noise(g, x0, m, s), "gauss")
    elif which == 5:
        k = int(rs.choice([3, 3, 5, 7]))
        w = [float(v) for v in ((rs.random(k * k) - 0.2) * 2 / k).astype(np.float32)]
        check(ops.linearFilter(x, w, k, k), lambda g: O.linear(g, w, k, k), "linear")
        check(ops.averageFilter(x, w, k, k), lambda g: O.average(g, w, k, k), "average")
    elif which == 6:
        w = [float(v) for v in (rs.integers(0, 5, 9) / 8.0)]
        check(ops.linearFilter(x, w, 3, 3), lambda g: O.linear(g, w, 3, 3), "linear dyadic")
        wi = [float(v) for v in rs.integers(-2, 9, 9)]
        check(ops.averageFilter(x, wi, 3, 3), lambda g: O.average(g, wi, 3, 3), "average integral")
    elif which == 7:
        kw, kh = int(rs.choice([3, 5, 7])), int(rs.choice([3, 5]))
        w = [1] * (kw * kh)
        check(ops.medianFilter(x, w, kw, kh), lambda g: O.median(g, w, kw, kh), "median")
        w3 = [int(v) for v in rs.integers(0, 4, 9)]
        w3[4] += 1
        check(ops.medianFilter(x, w3, 3, 3), lambda g: O.median(g, w3, 3, 3), "weighted median")
    elif which == 8:
        check(ops.sobelFilter(x), O.sobel, "sobel")
        check(ops.compassFilter(x), O.compass, "compass")
    elif which == 9:
        kw, kh = (3, 3) if rs.random() < 0.4 else (int(rs.choice([1, 3, 5, 7, 9, 4])), int(rs.choice([1, 3, 5, 7, 9, 6])))
        shape = rs.integers(0, 3)  # random, all ones, rows all equal
        if shape == 0:
            se = [int(v) for v in rs.integers(0, 2, kw * kh)]
        elif shape == 1:
            se = [1] * (kw * kh)
        else:
            se = [int(v) for v in rs.integers(0, 2, kw)] * kh
        check(ops.imageDilation(x, se, kw, kh), lambda g: O.dilate(g, se, kw, kh), "dilate")
        check(ops.imageErosion(x, se, kw, kh), lambda g: O.erode(g, se, kw, kh), "erode")
    elif which == 12:
        k = int(rs.choice([3, 5, 7, 9, 8, 11]))
        wt = float(rs.choice([1.0, 0.5, 1.0 / (k * k), 0.02, 2.0, 0.999, 1.0 / 3]))
        w = [wt] * (k * k)
        check(ops.linearFilter(x, w, k, k), lambda g: O.linear(g, w, k, k), "box linear")
        check(ops.averageFilter(x, w, k, k), lambda g: O.average(g, w, k, k), "box average")
    elif which == 10:
        cent = rs.integers(0, 257, (B, 2)).astype(np.int32)
        got = ops.kMeansThresholding(x, torch.from_numpy(cent).cuda()).cpu().numpy()
        for b in range(B):
            assert np.array_equal(got[b], O.kmeans2(imgs[b], int(cent[b, 0]), int(cent[b, 1]))[0]), (case, b)
    else:
        if W % 16 == 0 and (x.data_ptr() % 4 or (B > 1 and x.stride(0) % 4)):
            from imageanalysis_amd.lib import IakError
            with pytest.raises(IakError):
                ops.featureExtraction(x)
        else:
            got = ops.featureExtraction(x).cpu().numpy()
            for b in range(B):
                assert np.array_equal(got[b], O.features(imgs[b])), (case, b)
