"""Parity of every HIP entry point against the CPU oracle, through the C ABI, bit for bit.
All ops here are integer/byte results; float stages (grayscale, linear, Gaussian table,
Sobel) are also required bit-exact -- the +-1 LSB allowance of the float filters is not
used."""
import numpy as np
import pytest
import torch

import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu

SIZES = [(64, 48), (768, 72), (37, 23), (16, 5), (48, 3), (5, 7), (1, 1), (130, 9)]
KINDS = ["U", "C", "K"]


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from imageanalysis_amd import ops as _ops
    return _ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


def batch(kind, W, H, B=3):
    return synth.gray_batch(kind, W, H, B)


def each(fn_gpu, fn_ref, imgs):
    got = host(fn_gpu(dev(imgs)))
    for b in range(imgs.shape[0]):
        want = fn_ref(imgs[b])
        assert got[b].shape == want.shape
        if not np.array_equal(got[b], want):
            bad = np.argwhere(got[b] != want)
            raise AssertionError(f"image {b}: {len(bad)} mismatches, first at {bad[0]}: got "
                                 f"{got[b][tuple(bad[0])]} want {want[tuple(bad[0])]}")


@pytest.mark.parametrize("W,H", SIZES)
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("method", [0, 1])
def test_gray(ops, W, H, kind, method):
    rgb = synth.rgb_batch(kind, W, H, 3)
    got = host(ops.convertRGBToGrayscale(dev(rgb), method))
    for b in range(3):
        assert np.array_equal(got[b], O.gray(rgb[b], method))


def test_gray_all_triples_sample(ops):
    # every (R,G,B) with R,G on a stride-3 lattice and all B: covers the exact-multiple-of-100
    # cases where the double sum lands just below an integer (white -> 254)
    r, g, b = np.meshgrid(np.arange(0, 256, 3), np.arange(0, 256, 3), np.arange(256), indexing="ij")
    n = r.size
    W = 256
    H = (n + W - 1) // W
    rgb = np.zeros((3, H * W), np.uint8)
    rgb[0, :n], rgb[1, :n], rgb[2, :n] = r.ravel(), g.ravel(), b.ravel()
    rgb = rgb.reshape(3, H, W)
    got = host(ops.convertRGBToGrayscale(dev(rgb), 0))
    assert np.array_equal(got, O.gray(rgb, 0))
    assert int(host(ops.convertRGBToGrayscale(dev(np.full((3, 1, 1), 255, np.uint8)), 0))[0, 0]) == 254


def test_gray_bad_method(ops):
    from imageanalysis_amd.lib import IakError
    with pytest.raises(IakError):
        ops.convertRGBToGrayscale(dev(np.zeros((3, 4, 4), np.uint8)), 2)


@pytest.mark.parametrize("W,H", SIZES)
@pytest.mark.parametrize("kind", KINDS)
def test_gray_hist_fused(ops, W, H, kind):
    rgb = synth.rgb_batch(kind, W, H, 3)
    gray, hist = ops.convertRGBToGrayscale(dev(rgb), 0, with_hist=True)
    gray, hist = host(gray), host(hist)
    for b in range(3):
        want = O.gray(rgb[b], 0)
        assert np.array_equal(gray[b], want)
        assert np.array_equal(hist[b], O.hist(want))


@pytest.mark.parametrize("W,H", SIZES + [(768, 568)])
@pytest.mark.parametrize("kind", KINDS)
def test_histogram(ops, W, H, kind):
    imgs = batch(kind, W, H, 2)
    got = host(ops.calculateHistogram(dev(imgs)))
    for b in range(2):
        assert np.array_equal(got[b], O.hist(imgs[b]))
    assert (got[:, 0] == 0).all()


def test_histogram_all_zero_and_all_255(ops):
    z = np.zeros((1, 48, 64), np.uint8)
    assert host(ops.calculateHistogram(dev(z))).sum() == 0
    f = np.full((1, 48, 64), 255, np.uint8)
    h = host(ops.calculateHistogram(dev(f)))[0]
    assert h[255] == 48 * 64 and h.sum() == 48 * 64


@pytest.mark.parametrize("W,H", SIZES)
@pytest.mark.parametrize("kind", KINDS)
def test_equalization(ops, W, H, kind):
    imgs = batch(kind, W, H)
    h = ops.calculateHistogram(dev(imgs))
    m = host(ops.equalizeHistogram(h, W * H))
    for b in range(imgs.shape[0]):
        assert np.array_equal(m[b], O.eq_map(O.hist(imgs[b]), W * H))
    each(ops.histogramEqualization, O.equalize, imgs)


def test_eq_map_kat(ops):
    h = np.zeros((1, 256), np.int32)
    h[0, 10], h[0, 20], h[0, 200] = 4, 4, 8
    m = host(ops.equalizeHistogram(dev(h), 16))[0]
    assert m[9] == 0 and (m[10:20] == 63).all() and (m[20:200] == 127).all() and (m[200:] == 255).all()


LEVEL_SETS = [
    [[10, 20, 15], [30, 250, 77]],                                   # input.json:24-36
    [[0, 50, 25], [50, 100, 75], [100, 150, 125], [150, 200, 175], [200, 250, 225], [40, 256, 300]],
    [],
    [[0, 256, 7]],
]


@pytest.mark.parametrize("levels", LEVEL_SETS)
@pytest.mark.parametrize("W,H", [(64, 48), (37, 23)])
def test_quantization(ops, W, H, levels):
    imgs = batch("U", W, H)
    each(lambda x: ops.imageQuantization(x, levels), lambda x: O.quantize(x, levels), imgs)


@pytest.mark.parametrize("t", [-1, 0, 40, 127, 254, 255, 300])
@pytest.mark.parametrize("W,H", [(64, 48), (37, 23)])
def test_threshold(ops, W, H, t):
    imgs = batch("U", W, H)
    each(lambda x: ops.thresholdImage(x, t), lambda x: O.threshold(x, t), imgs)


@pytest.mark.parametrize("W,H", SIZES)
@pytest.mark.parametrize("kind", KINDS)
def test_otsu(ops, W, H, kind):
    imgs = batch(kind, W, H)
    out, thr = ops.otsuThresholdImage(dev(imgs), return_threshold=True)
    out, thr = host(out), host(thr)
    for b in range(imgs.shape[0]):
        want, wthr = O.otsu(imgs[b])
        assert thr[b] == wthr
        assert np.array_equal(out[b], want)


@pytest.mark.parametrize("W,H", SIZES)
@pytest.mark.parametrize("kind", KINDS)
def test_mean_std(ops, W, H, kind):
    imgs = batch(kind, W, H)
    ms = host(ops.calculateMeanAndStdDev(dev(imgs)))
    for b in range(imgs.shape[0]):
        m, s = O.mean_std(imgs[b])
        assert ms[b, 0].tobytes() == m.tobytes()
        assert ms[b, 1].tobytes() == s.tobytes() or (np.isnan(ms[b, 1]) and np.isnan(s))


def test_rng_planes(ops):
    n = 70000  # crosses bit 16 of the subsequence number
    rng = ops.Randomness(n)
    x0, x1 = O.xorwow_planes(n)
    assert np.array_equal(host(rng.x0).view(np.uint32), x0)
    assert np.array_equal(host(rng.x1).view(np.uint32), x1)


def test_rng_planes_high_subsequence(ops):
    # spot-check subsequence numbers up to 2^24 (4096x4096 images) without the full oracle plane
    n = 1 << 24
    rng = ops.Randomness(n)
    x0 = host(rng.x0).view(np.uint32)
    x1 = host(rng.x1).view(np.uint32)
    for tid in [0, 1, 65535, 65536, 436223, 1 << 20, (1 << 24) - 1, 0xABCDEF]:
        s = O.xorwow_stream(1234, tid, 2)
        assert x0[tid] == s[0] and x1[tid] == s[1]


@pytest.mark.parametrize("W,H", [(64, 48), (37, 23), (768, 72)])
@pytest.mark.parametrize("level", [50, 0, 1, 7, 255])
def test_salt_pepper(ops, W, H, level):
    imgs = batch("C", W, H)
    rng = ops.Randomness(W * H)
    x0, x1 = O.xorwow_planes(W * H)
    each(lambda x: ops.saltAndPepperNoise(x, rng, level), lambda x: O.salt_pepper(x, x0, x1, level), imgs)


@pytest.mark.parametrize("W,H", [(64, 48), (37, 23), (768, 72)])
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("mean,std", [(-1.0, -1.0), (128.0, 20.0), (-1.0, 5.0), (100.0, 0.0)])
def test_gauss_noise(ops, W, H, kind, mean, std):
    imgs = batch(kind, W, H)
    rng = ops.Randomness(W * H)
    x0, _ = O.xorwow_planes(W * H)
    each(lambda x: ops.addGaussianNoise(x, rng, mean, std), lambda x: O.gauss_noise(x, x0, mean, std), imgs)


KERNELS3 = {
    "input.json": [1, 1, 1, 10, 1, 1, 1, 1, 1],
    "linear-filter.json": [-1, 0, -1, -2, 0, -2, -1, 0, -1],
    "box9": [1.0 / 9] * 9,
    "gauss16": [1 / 16, 2 / 16, 1 / 16, 2 / 16, 4 / 16, 2 / 16, 1 / 16, 2 / 16, 1 / 16],
    "sharpen": [0, -1, 0, -1, 5, -1, 0, -1, 0],
    "odd": [0.3, -0.7, 1.9, 0.01, 2.5, -1.3, 0.77, 0.2, -0.45],
    "dyadic": [0.5, 0.25, 0, 0.125, 1, 2, 0.0625, 0, 0.5],
    "dyadic-big": [3.5, 0.25, 7, 0.125, 1, 2, 0.0625, 9, 0.5],
    "fine": [1 / 4096, 0, 0, 0, 1 / 2, 0, 0, 0, 1 / 8192],
    "box16": [1 / 16] * 9,
    "plus": [0, 1 / 8, 0, 1 / 8, 1 / 2, 1 / 8, 0, 1 / 8, 0],
    "identity": [0, 0, 0, 0, 1, 0, 0, 0, 0],
    "corners": [1 / 4, 0, 1 / 4, 0, 0, 0, 1 / 4, 0, 1 / 4],
    "too-heavy": [1 / 4, 1 / 4, 1 / 4, 1 / 4, 1 / 4, 1 / 4, 1 / 4, 1 / 4, 1 / 4],
    "tiny": [1 / 128, 1 / 64, 1 / 128, 1 / 64, 1 / 2, 1 / 64, 1 / 128, 1 / 64, 1 / 128],
}


@pytest.mark.parametrize("name", list(KERNELS3))
@pytest.mark.parametrize("W,H", SIZES)
@pytest.mark.parametrize("avg", [False, True])
def test_linear3(ops, name, W, H, avg):
    k = KERNELS3[name]
    imgs = batch("U", W, H)
    if avg:
        each(lambda x: ops.averageFilter(x, k, 3, 3), lambda x: O.average(x, k, 3, 3), imgs)
    else:
        each(lambda x: ops.linearFilter(x, k, 3, 3), lambda x: O.linear(x, k, 3, 3), imgs)


@pytest.mark.parametrize("kW,kH", [(5, 5), (11, 11), (3, 5), (7, 1), (1, 1), (2, 2), (4, 3)])
@pytest.mark.parametrize("W,H", [(64, 48), (37, 23), (16, 5)])
def test_linear_generic(ops, kW, kH, W, H):
    rs = np.random.default_rng(kW * 100 + kH)
    k = (rs.random(kW * kH).astype(np.float32) / (kW * kH) * 1.7)
    imgs = batch("C", W, H, 2)
    each(lambda x: ops.linearFilter(x, k, kW, kH), lambda x: O.linear(x, k, kW, kH), imgs)
    each(lambda x: ops.averageFilter(x, k * 9, kW, kH), lambda x: O.average(x, k * 9, kW, kH), imgs)


@pytest.mark.parametrize("kW,kH", [(5, 5), (11, 11), (25, 25), (13, 3), (1, 9), (9, 1), (15, 7), (6, 5)])
@pytest.mark.parametrize("W,H,kind", [(200, 70, "C"), (131, 45, "U"), (64, 33, "K"), (9, 40, "U")])
def test_linear_tiled(ops, kW, kH, W, H, kind):
    """Tiles that are cut by the right and bottom edges, windows wider than the image, the
    wrap-around taps at the row ends, negative and large weights (signed sums, clamping, the
    average's wrap-around), and weights too large for exact float sums (one-pixel-per-lane path)."""
    rs = np.random.default_rng(kW * 131 + kH)
    imgs = batch(kind, W, H, 2)
    k = (rs.random(kW * kH).astype(np.float32) / (kW * kH) * 1.7)
    each(lambda x: ops.linearFilter(x, k, kW, kH), lambda x: O.linear(x, k, kW, kH), imgs)
    each(lambda x: ops.averageFilter(x, k * 9, kW, kH), lambda x: O.average(x, k * 9, kW, kH), imgs)
    box = np.full(kW * kH, 1.0 / (kW * kH), np.float32)
    each(lambda x: ops.linearFilter(x, box, kW, kH), lambda x: O.linear(x, box, kW, kH), imgs)
    signed = ((rs.random(kW * kH) - 0.5) * 4).astype(np.float32)
    each(lambda x: ops.linearFilter(x, signed, kW, kH), lambda x: O.linear(x, signed, kW, kH), imgs)
    each(lambda x: ops.averageFilter(x, signed * 50, kW, kH), lambda x: O.average(x, signed * 50, kW, kH), imgs)
    huge = (rs.random(kW * kH) * 3000).astype(np.float32)
    each(lambda x: ops.averageFilter(x, huge, kW, kH), lambda x: O.average(x, huge, kW, kH), imgs)


@pytest.mark.parametrize("w", [1.0 / 121, 1.0 / 25, 1.0, 0.5, 0.999, 1.0 / 3, 2.5, 3.7, 0.0078125, 1e-3])
@pytest.mark.parametrize("kW,kH", [(11, 11), (5, 5), (25, 25), (7, 3), (1, 9), (4, 6), (7, 7), (9, 9), (8, 8)])
@pytest.mark.parametrize("W,H,kind", [(200, 70, "C"), (131, 150, "U"), (64, 64, "K"), (9, 40, "U"), (16, 5, "U"), (8, 19, "C")])
def test_box_filters(ops, w, kW, kH, W, H, kind):
    # every tap the same weight: integer box sums where the per-tap truncation provably commutes
    # with the sum, the tap-by-tap kernels otherwise (weights near an integer from below, big sums)
    imgs = batch(kind, W, H, 2)
    k = np.full(kW * kH, w, np.float32)
    each(lambda x: ops.linearFilter(x, k, kW, kH), lambda x: O.linear(x, k, kW, kH), imgs)
    each(lambda x: ops.averageFilter(x, k, kW, kH), lambda x: O.average(x, k, kW, kH), imgs)


@pytest.mark.parametrize("kW,kH,w", [(11, 11, 1.0 / 121), (7, 7, 1.0 / 49), (13, 13, 1.0 / 169), (11, 5, 1.0 / 55),
                                       (3, 15, 1.0 / 45), (25, 9, 1.0 / 121), (11, 11, 1.0 / 242), (11, 11, 3.0 / 242),
                                       (9, 9, 1.0 / 81), (15, 15, 1.0 / 225), (5, 5, 1.0 / 100), (7, 7, 3.0 / 242)])
@pytest.mark.parametrize("W,H", [(200, 70), (64, 64), (131, 45), (9, 40), (2080, 40), (4100, 35), (256, 100), (32, 9), (4, 30)])
def test_box_filters_whose_taps_depend_on_the_running_sum(ops, kW, kH, w, W, H):
    """Weights like fl(1/121): 121 * w lies one float step below 1, so that tap adds 0 to an empty
    running sum and 1 to any other (imageAnalysis.cu:517-524 truncates after every tap) -- the
    result depends on the order of the taps.  Images made of the critical pixel values and their
    neighbours, in dark / bright arrangements that put the first contributing tap anywhere in a
    window, against the oracle's tap-by-tap loop."""
    rs = np.random.default_rng(kW * 1000 + kH * 10 + W)
    q = int(round(1.0 / w)) if w < 0.5 else 1
    critical = sorted({v for m in (1, 2, 3) for v in (m * q - 1, m * q, m * q + 1) if 0 <= v <= 255} | {0, 255})
    imgs = []
    # (1) only critical values; (2) mostly dark with a few of them; (3) dark top half, bright bottom; (4) bright left columns
    a = rs.choice(critical, size=(H, W)).astype(np.uint8)
    imgs.append(a)
    b = rs.integers(0, min(q, 255), size=(H, W)).astype(np.uint8)
    m = rs.random((H, W)) < 0.03
    b[m] = rs.choice(critical, size=int(m.sum()))
    imgs.append(b)
    c = b.copy()
    c[H // 2:] = rs.choice(critical + [200, 201], size=(H - H // 2, W))
    imgs.append(c)
    d = b.copy()
    d[:, : max(1, W // 3)] = rs.choice(critical, size=(H, max(1, W // 3)))
    imgs.append(d)
    imgs = np.stack(imgs)
    k = np.full(kW * kH, w, np.float32)
    each(lambda x: ops.linearFilter(x, k, kW, kH), lambda x: O.linear(x, k, kW, kH), imgs)
    each(lambda x: ops.averageFilter(x, k, kW, kH), lambda x: O.average(x, k, kW, kH), imgs)


def test_box_filter_with_sum_dependent_taps_in_a_strided_batch(ops):
    """The bit-sliced box kernel on images inside a larger allocation whose gaps hold 255s: rows past an image's
    end must read as zeros, not as the gap or the next image, and nothing is written between the images."""
    import ctypes as C
    from imageanalysis_amd import lib as L
    W, H, B = 132, 41, 3
    rs = np.random.default_rng(5)
    imgs = rs.choice([0, 120, 121, 122, 241, 242, 243, 255], size=(B, H, W)).astype(np.uint8)
    stride = W * H + 76
    buf = torch.full((B * stride + 64,), 255, dtype=torch.uint8, device="cuda")
    buf[: B * stride].view(B, stride)[:, : W * H] = torch.from_numpy(imgs.reshape(B, -1)).cuda()
    out = torch.zeros_like(buf)
    k = np.full(121, 1.0 / 121, np.float32)
    rc = L.load().iak_linear(buf.data_ptr(), out.data_ptr(), W, H, B, stride, stride, k.ctypes.data_as(C.c_void_p), 11, 11, 0, None)
    assert rc == 0
    torch.cuda.synchronize()
    got = out[: B * stride].view(B, stride)
    for b in range(B):
        assert np.array_equal(got[b, : W * H].cpu().numpy().reshape(H, W), O.linear(imgs[b], k, 11, 11)), b
    assert int(got[:, W * H:].max()) == 0


@pytest.mark.parametrize("kW,kH", [(5, 5), (7, 7), (15, 15), (3, 9), (9, 3), (1, 5), (5, 1), (15, 17)])
@pytest.mark.parametrize("W,H,kind", [(200, 70, "C"), (131, 150, "U"), (64, 33, "K"), (300, 20, "C"), (5, 9, "U")])
def test_median_sliding_histogram(ops, kW, kH, W, H, kind):
    """All-ones medians larger than 3x3: strips cut by the bottom edge, more than one strip per
    column (H > 64), more than one workgroup per row (W > 128), windows taller than the image,
    17x15 = 255 pixels (the counters' limit) on a constant image."""
    imgs = batch(kind, W, H, 2)
    if kind == "C":
        imgs[1, : H // 2] = 0
    w = [1] * (kW * kH)
    each(lambda x: ops.medianFilter(x, w, kW, kH), lambda x: O.median(x, w, kW, kH), imgs)


@pytest.mark.parametrize("kW,kH", [(5, 5), (7, 7), (5, 7), (7, 5), (3, 5), (5, 3), (3, 7), (7, 3)])
@pytest.mark.parametrize("W,H", [(8, 11), (16, 3), (24, 2), (768, 40), (64, 64), (8, 1), (40, 5), (4, 1), (4, 9), (36, 33),
                                 (260, 35), (516, 70), (252, 31), (1028, 3)])
@pytest.mark.parametrize("kind", ["U", "C", "K", "binary", "ramp"])
def test_median_bit_sliced(ops, kW, kH, W, H, kind):
    """All-ones medians of up to 49 taps on widths that are multiples of 4 (the bit-plane kernel's cases): every
    window shape it is built for; images of one word, one row, fewer rows or columns than the window, one column of
    tiles plus a few pixels (260, 516, 1028), one pixel short of a tile (252 x 31); pixel values at both ends of
    the range and ties everywhere; two images, so the second starts where the first one's halo loads reach."""
    rs = np.random.default_rng(W * 100 + H)
    if kind == "binary":
        imgs = (rs.integers(0, 2, (2, H, W)) * 255).astype(np.uint8)
    elif kind == "ramp":
        imgs = ((np.arange(2 * H * W).reshape(2, H, W) * 7) % 256).astype(np.uint8)
    else:
        imgs = batch(kind, W, H, 2)
    w = [1] * (kW * kH)
    each(lambda x: ops.medianFilter(x, w, kW, kH), lambda x: O.median(x, w, kW, kH), imgs)


def test_median_bit_sliced_in_a_strided_batch(ops):
    """Images inside a larger allocation: rows of the next image's slot must not leak into the last rows' windows,
    and an image stride that is no multiple of the tile keeps the 4-byte alignment the kernel asks for."""
    W, H, B = 132, 37, 3
    imgs = batch("U", W, H, B)
    stride = W * H + 52
    buf = torch.full((B * stride + 64,), 255, dtype=torch.uint8, device="cuda")
    view = buf[: B * stride].view(B, stride)
    view[:, : W * H] = torch.from_numpy(imgs.reshape(B, -1)).cuda()
    out = torch.zeros_like(buf)
    from imageanalysis_amd import lib as L
    h = L.load()
    import ctypes as C
    for kW, kH in ((7, 7), (5, 3)):
        w = (C.c_int * (kW * kH))(*[1] * (kW * kH))
        rc = h.iak_median(buf.data_ptr(), out.data_ptr(), W, H, B, stride, stride, w, kW, kH, None)
        assert rc == 0
        torch.cuda.synchronize()
        got = out[: B * stride].view(B, stride)[:, : W * H].cpu().numpy().reshape(B, H, W)
        for b in range(B):
            assert np.array_equal(got[b], O.median(imgs[b], [1] * (kW * kH), kW, kH)), (kW, kH, b)
        assert int(out[: B * stride].view(B, stride)[:, W * H:].max()) == 0  # nothing written between the images


@pytest.mark.parametrize("kW,kH", [(51, 5), (255, 1), (1, 255), (27, 9)])
def test_median_wide_windows(ops, kW, kH):
    # at most 255 taps but wider than the sliding-histogram kernel's 25 columns: the bisection kernel takes them
    imgs = batch("U", 300, 40, 1) if kH < 100 else batch("U", 20, 300, 1)
    w = [1] * (kW * kH)
    each(lambda x: ops.medianFilter(x, w, kW, kH), lambda x: O.median(x, w, kW, kH), imgs)


def test_median_above_255_taps_uses_bisection(ops):
    imgs = batch("U", 70, 40, 1)
    w = [1] * (17 * 17)
    each(lambda x: ops.medianFilter(x, w, 17, 17), lambda x: O.median(x, w, 17, 17), imgs)


@pytest.mark.parametrize("W,H", SIZES + [(768, 568)])
@pytest.mark.parametrize("kind", KINDS)
def test_median3(ops, W, H, kind):
    imgs = batch(kind, W, H, 2)
    w = [1] * 9
    each(lambda x: ops.medianFilter(x, w, 3, 3), lambda x: O.median(x, w, 3, 3), imgs)


@pytest.mark.parametrize("kW,kH,weights", [
    (5, 5, [1] * 25), (7, 7, [1] * 49), (3, 3, [1, 2, 1, 2, 3, 2, 1, 2, 1]), (3, 1, [1, 1, 1]),
    (1, 1, [1]), (3, 3, [0, 1, 0, 1, 1, 1, 0, 1, 0]), (5, 3, list(range(15)))])
@pytest.mark.parametrize("W,H", [(64, 48), (37, 23)])
def test_median_generic(ops, kW, kH, weights, W, H):
    imgs = batch("U", W, H, 2)
    each(lambda x: ops.medianFilter(x, weights, kW, kH), lambda x: O.median(x, weights, kW, kH), imgs)


def test_median_rejects(ops):
    from imageanalysis_amd.lib import IakError
    x = dev(batch("U", 16, 8, 1))
    with pytest.raises(IakError):
        ops.medianFilter(x, [1] * 4, 2, 2)
    with pytest.raises(IakError):
        ops.medianFilter(x, [1, -1, 1, 1, 1, 1, 1, 1, 1], 3, 3)
    assert O.median(batch("U", 16, 8, 1)[0], [1] * 4, 2, 2) is None


@pytest.mark.parametrize("W,H", SIZES + [(768, 568)])
@pytest.mark.parametrize("kind", KINDS)
def test_sobel_compass(ops, W, H, kind):
    imgs = batch(kind, W, H, 2)
    each(ops.sobelFilter, O.sobel, imgs)
    each(ops.compassFilter, O.compass, imgs)


def test_sobel_every_magnitude(ops):
    """Every pair of clamped responses (gx, gy) a 3x3 window can produce -- equal parity, or one of them
    clamped -- each in an isolated 3x3 cell on a 4-pixel pitch, so the square root is checked on every
    sum of squares it can meet (imageAnalysis.cu:552)."""
    cells = []  # (gx, gy, 3x3 values)
    for gx in range(256):
        for gy in range(gx & 1, 256, 2):
            p = gx & 1
            cells.append((gx, gy, [0, 0, 0, 0, 0, (gx - p) // 2, 0, (gy - p) // 2, p]))
    for g in range(256):
        z, y = g & 1, g >> 1
        cells.append((255, g, [0, 0, 0, 0, 0, 200, 0, y, z]))      # gx clamped from above
        cells.append((0, g, [0, 0, 0, 255, 0, 0, 0, y, z]))        # gx clamped from below
        cells.append((g, 255, [0, 0, 0, 0, 0, y, 0, 200, z]))      # gy clamped from above
        cells.append((g, 0, [0, 255, 0, 0, 0, y, 0, 0, z]))        # gy clamped from below
    per_row = 256
    rows = (len(cells) + per_row - 1) // per_row
    img = np.zeros((4 * rows + 1, 4 * per_row), np.uint8)
    want_centre = np.zeros((rows, per_row), np.uint8)
    for i, (gx, gy, v) in enumerate(cells):
        r, c = divmod(i, per_row)
        img[4 * r + 1:4 * r + 4, 4 * c + 1:4 * c + 4] = np.array(v, np.uint8).reshape(3, 3)
        want_centre[r, c] = int(np.floor(np.sqrt(float(gx * gx + gy * gy)))) & 255
    got = ops.sobelFilter(dev(img)).cpu().numpy()
    assert np.array_equal(got, O.sobel(img))
    centre = got[2::4, 2::4][:rows]
    flat = np.arange(rows * per_row).reshape(rows, per_row) < len(cells)
    assert np.array_equal(centre[flat], want_centre[flat])


SES = {
    "ones3": (3, 3, [1] * 9),
    "cross3": (3, 3, [0, 1, 0, 1, 1, 1, 0, 1, 0]),
    "asym3": (3, 3, [1, 0, 0, 0, 0, 1, 0, 7, 0]),
    "zero3": (3, 3, [0] * 9),
    "ones5": (5, 5, [1] * 25),
    "bar": (5, 1, [1, 0, 1, 1, 0]),
    "col": (1, 3, [1, 1, 1]),
    "even": (2, 2, [1, 1, 1, 1]),
    "even43": (4, 3, [1, 0, 1, 1, 0, 1, 1, 0, 1, 1, 0, 0]),
    "rand7x3": (7, 3, [1, 0, 0, 1, 0, 1, 1, 0, 1, 0, 0, 0, 1, 0, 1, 0, 1, 1, 0, 0, 1]),
    "sparse9": (9, 9, [1 if (i * 7) % 11 == 0 else 0 for i in range(81)]),
    "ones7": (7, 7, [1] * 49),
    "box9x3": (9, 3, [1] * 27),
    "line3x9": (3, 9, [1, 0, 1] * 9),
    "disk5": (5, 5, [0, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 0]),
    "ones25": (25, 25, [1] * 625),
    "corners25": (25, 25, [1 if i in (0, 24, 600, 624, 312) else 0 for i in range(625)]),
}


@pytest.mark.parametrize("se", list(SES))
@pytest.mark.parametrize("W,H", SIZES + [(200, 70), (131, 45), (8, 11), (24, 2)])
def test_morphology(ops, se, W, H):
    kW, kH, v = SES[se]
    rs = np.random.default_rng(5)
    imgs = batch("U", W, H, 2)
    imgs[rs.random(imgs.shape) < 0.6] = 0  # sparse foreground so erosion/dilation are non-trivial
    each(lambda x: ops.imageErosion(x, v, kW, kH), lambda x: O.erode(x, v, kW, kH), imgs)
    each(lambda x: ops.imageDilation(x, v, kW, kH), lambda x: O.dilate(x, v, kW, kH), imgs)
    dense = batch("C", W, H, 1)
    each(lambda x: ops.imageErosion(x, v, kW, kH), lambda x: O.erode(x, v, kW, kH), dense)
    each(lambda x: ops.imageDilation(x, v, kW, kH), lambda x: O.dilate(x, v, kW, kH), dense)


def test_strided_batch_views(ops):
    # images embedded in a larger allocation: stride > W*H, 16-byte aligned and not
    W, H = 64, 48
    for pad in (16, 5):
        buf = torch.zeros((3, H * W + pad), dtype=torch.uint8, device="cuda")
        imgs = batch("U", W, H, 3)
        view = buf[:, :H * W].view(3, H, W)
        view.copy_(dev(imgs))
        got = host(ops.medianFilter(view, [1] * 9, 3, 3))
        for b in range(3):
            assert np.array_equal(got[b], O.median(imgs[b], [1] * 9, 3, 3))
        got = host(ops.histogramEqualization(view))
        for b in range(3):
            assert np.array_equal(got[b], O.equalize(imgs[b]))


@pytest.mark.parametrize("seed,n", [(1234, 768 * 568), (1234, 70000), (987654321012345, 5000), (0, 300)])
def test_random_planes_match_rocrand_xorwow(ops, seed, n, tmp_path):
    """A second implementation of the generator: rocRAND's XORWOW device API (shipped with ROCm)
    called the way the reference calls cuRAND -- init(seed, subsequence = pixel index, offset 0),
    then two draws.  rocRAND is the same generator as cuRAND's except for the constants that
    scramble the seed into the start state, so the start state is computed here with rocRAND's
    constants and handed to iak_rng_planes_from_state: the 2^67-step subsequence jump, the step
    and the output function must then match rocRAND bit for bit."""
    import ctypes as C
    import os
    import subprocess
    from imageanalysis_amd import lib as L
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "imageanalysis_amd", "lib", "rocrand_check")
    out = tmp_path / "draws.bin"
    r = subprocess.run([exe, str(n), str(seed), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    want = np.fromfile(out, dtype=np.uint32).reshape(n, 2)
    M = 0xffffffff
    s0 = (seed & M) ^ 0x2c7f967f                      # rocrand_xorwow.h: xorwow_engine constructor
    s1 = ((seed >> 32) & M) ^ 0xa03697cb
    t0 = (1228688033 * s0) & M
    t1 = (2073658381 * s1) & M
    state = np.array([(123456789 + t0) & M, 362436069 ^ t0, (521288629 + t1) & M, 88675123 ^ t1, (5783321 + t0) & M,
                      (6615241 + t1 + t0) & M], np.uint32)
    x0 = torch.empty(n, dtype=torch.int32, device="cuda")
    x1 = torch.empty(n, dtype=torch.int32, device="cuda")
    L.check(L.load().iak_rng_planes_from_state(x0.data_ptr(), x1.data_ptr(), n, state.ctypes.data,
                                               torch.cuda.current_stream().cuda_stream), "iak_rng_planes_from_state")
    assert np.array_equal(x0.cpu().numpy().view(np.uint32), want[:, 0])
    assert np.array_equal(x1.cpu().numpy().view(np.uint32), want[:, 1])


@pytest.mark.parametrize("weights", [
    [1, 2, 1, 2, 3, 2, 1, 2, 1], [0, 1, 0, 1, 1, 1, 0, 1, 0], [0, 0, 0, 0, 1, 0, 0, 0, 0], [255] * 9,
    [255, 0, 7, 1, 1, 1, 30, 2, 0], [1, 1, 1, 1, 1, 1, 1, 1, 2], [256, 1, 1, 1, 1, 1, 1, 1, 1], [5, 0, 0, 0, 0, 0, 0, 0, 5]])
@pytest.mark.parametrize("W,H,kind", [(64, 48, "U"), (37, 23, "U"), (768, 24, "C")])
def test_median3_weighted(ops, weights, W, H, kind):
    """Weighted 3x3 medians: the sorting-network path (weights up to 255, zeros included) and the
    bisection path behind it (a weight of 256, widths that are not a multiple of 8)."""
    imgs = batch(kind, W, H, 2)
    each(lambda x: ops.medianFilter(x, weights, 3, 3), lambda x: O.median(x, weights, 3, 3), imgs)


@pytest.mark.parametrize("W", list(range(8, 42)) + [67, 250, 770])
def test_stencils3_any_width_and_alignment(ops, W):
    """The 3x3 operations on widths that are no multiple of 8, with images and outputs at odd byte
    offsets and strides: results as the oracle's, and not one byte written outside the images."""
    from imageanalysis_amd import lib as L
    lib = L.load()
    rs = np.random.default_rng(W)
    stream = torch.cuda.current_stream().cuda_stream
    k_f = np.array([0.1, 0.2, 0.05, 0.3, 0.15, 0.02, 0.08, 0.07, 0.03], np.float32)
    k_w = np.array([1, 2, 1, 2, 3, 2, 1, 2, 1], np.int32)
    k_1 = np.ones(9, np.int32)
    k_se = np.array([0, 1, 0, 1, 1, 1, 0, 1, 1], np.int32)
    for H in (1, 2, 9, 26):
        B, N = 2, W * H
        imgs = batch("U", W, H, B)
        imgs[rs.random(imgs.shape) < 0.3] = 0
        in_off, out_off = int(rs.integers(0, 8)), int(rs.integers(0, 8))
        in_stride, out_stride = N + int(rs.integers(0, 6)), N + int(rs.integers(0, 6))
        src = torch.zeros(in_off + in_stride * B + 16, dtype=torch.uint8, device="cuda")
        x = torch.as_strided(src[in_off:], (B, H, W), (in_stride, W, 1))
        x.copy_(dev(imgs))
        cases = [
            ("sobel", lambda i, o: lib.iak_sobel(i, o, W, H, B, in_stride, out_stride, stream), O.sobel),
            ("compass", lambda i, o: lib.iak_compass(i, o, W, H, B, in_stride, out_stride, stream), O.compass),
            ("median", lambda i, o: lib.iak_median(i, o, W, H, B, in_stride, out_stride, k_1.ctypes.data, 3, 3, stream),
             lambda a: O.median(a, k_1, 3, 3)),
            ("wmedian", lambda i, o: lib.iak_median(i, o, W, H, B, in_stride, out_stride, k_w.ctypes.data, 3, 3, stream),
             lambda a: O.median(a, k_w, 3, 3)),
            ("linear", lambda i, o: lib.iak_linear(i, o, W, H, B, in_stride, out_stride, k_f.ctypes.data, 3, 3,
                                                   L.IAK_LINEAR_CLAMP, stream), lambda a: O.linear(a, k_f, 3, 3)),
            ("average", lambda i, o: lib.iak_linear(i, o, W, H, B, in_stride, out_stride, k_f.ctypes.data, 3, 3,
                                                    L.IAK_LINEAR_AVERAGE, stream), lambda a: O.average(a, k_f, 3, 3)),
            ("erode", lambda i, o: lib.iak_erode(i, o, W, H, B, in_stride, out_stride, k_se.ctypes.data, 3, 3, stream),
             lambda a: O.erode(a, k_se, 3, 3)),
            ("dilate", lambda i, o: lib.iak_dilate(i, o, W, H, B, in_stride, out_stride, k_se.ctypes.data, 3, 3, stream),
             lambda a: O.dilate(a, k_se, 3, 3)),
        ]
        for name, run, ref in cases:
            dst = torch.full((out_off + out_stride * B + 16,), 0xA5, dtype=torch.uint8, device="cuda")
            L.check(run(x.data_ptr(), dst.data_ptr() + out_off), name)
            got = host(dst)
            expect = np.full(got.shape, 0xA5, np.uint8)
            for b in range(B):
                expect[out_off + b * out_stride: out_off + b * out_stride + N] = ref(imgs[b]).reshape(-1)
            bad = np.flatnonzero(got != expect)
            assert bad.size == 0, (f"{name} {W}x{H} in+{in_off}/{in_stride} out+{out_off}/{out_stride}: {bad.size} bytes "
                                   f"differ, first at {bad[0]} (got {got[bad[0]]}, want {expect[bad[0]]})")
