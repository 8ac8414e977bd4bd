"""Frozen vectors (tests/golden/*.npz, written by tests/golden/make_golden.py from the oracle):
the oracle must keep reproducing them (CPU), and the HIP kernels must hit them (GPU)."""
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
ONES, CROSS = [1] * 9, [0, 1, 0, 1, 1, 1, 0, 1, 0]
K_INPUT = [1, 1, 1, 10, 1, 1, 1, 1, 1]
K_LINEAR = [-1, 0, -1, -2, 0, -2, -1, 0, -1]
K_G16 = [1 / 16, 2 / 16, 1 / 16, 2 / 16, 4 / 16, 2 / 16, 1 / 16, 2 / 16, 1 / 16]
K_BOX9 = [1.0 / 9] * 9
LEVELS = [10, 20, 15, 30, 250, 77]
FIXTURES = ["ops_C_64x48.npz", "ops_U_37x23.npz"]


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_reproduces_golden(name):
    d = load(name)
    rgb, g = d["rgb"], d["gray"]
    H, W = g.shape
    assert np.array_equal(O.gray(rgb, 0), g)
    assert np.array_equal(O.gray(rgb, 1), d["gray_avg"])
    assert np.array_equal(O.hist(g), d["hist"])
    assert np.array_equal(O.eq_map(d["hist"], W * H), d["eq_map"])
    assert np.array_equal(O.equalize(g), d["equalize"])
    assert np.array_equal(O.quantize(g, LEVELS), d["quantize"])
    assert np.array_equal(O.threshold(g, 40), d["threshold40"])
    otsu, thr = O.otsu(g)
    assert np.array_equal(otsu, d["otsu"]) and thr == int(d["otsu_thr"])
    x0, x1 = O.xorwow_planes(W * H)
    assert np.array_equal(x0, d["x0"]) and np.array_equal(x1, d["x1"])
    assert np.array_equal(O.salt_pepper(g, x0, x1, 50), d["salt_pepper50"])
    assert np.array_equal(O.gauss_noise(g, x0, -1, -1), d["gauss_auto"])
    assert np.array_equal(O.gauss_noise(g, x0, 100.0, 25.0), d["gauss_100_25"])
    assert np.array_equal(O.linear(g, K_INPUT, 3, 3), d["linear_input"])
    assert np.array_equal(O.linear(g, K_LINEAR, 3, 3), d["linear_neg"])
    assert np.array_equal(O.linear(g, K_G16, 3, 3), d["linear_g16"])
    assert np.array_equal(O.linear(g, K_BOX9, 3, 3), d["linear_box9"])
    assert np.array_equal(O.average(g, K_INPUT, 3, 3), d["average_input"])
    assert np.array_equal(O.linear(g, [1.0 / 121] * 121, 11, 11), d["linear_box11"])
    assert np.array_equal(O.median(g, ONES, 3, 3), d["median3"])
    assert np.array_equal(O.median(g, [1] * 49, 7, 7), d["median7"])
    assert np.array_equal(O.sobel(g), d["sobel"])
    assert np.array_equal(O.compass(g), d["compass"])
    assert np.array_equal(O.dilate(otsu, CROSS, 3, 3), d["dilate_cross"])
    assert np.array_equal(O.erode(otsu, CROSS, 3, 3), d["erode_cross"])
    assert np.array_equal(O.dilate(otsu, ONES, 3, 3), d["dilate_ones"])
    assert np.array_equal(O.erode(g, ONES, 3, 3), d["erode_ones_gray"])
    assert np.array_equal(O.kmeans2(g, 215, 101)[0], d["kmeans_215_101"])
    assert np.array_equal(O.features(g), d["features"])


def test_oracle_reproduces_xorwow_golden():
    d = load("xorwow_seed1234.npz")
    for s, want in zip(d["subsequences"], d["outputs"]):
        assert np.array_equal(O.xorwow_stream(1234, int(s), 4), want)


@pytest.mark.gpu
@pytest.mark.parametrize("name", FIXTURES)
def test_gpu_hits_golden(name):
    import torch
    from imageanalysis_amd import ops
    d = load(name)
    g = torch.from_numpy(d["gray"]).cuda()
    rgb = torch.from_numpy(d["rgb"]).cuda()
    H, W = d["gray"].shape

    def eq(t, key):
        assert np.array_equal(t.cpu().numpy(), d[key]), key

    gray, hist = ops.convertRGBToGrayscale(rgb, 0, with_hist=True)
    eq(gray, "gray")
    eq(hist, "hist")
    eq(ops.convertRGBToGrayscale(rgb, 1), "gray_avg")
    eq(ops.equalizeHistogram(hist, W * H), "eq_map")
    eq(ops.histogramEqualization(g), "equalize")
    eq(ops.imageQuantization(g, LEVELS), "quantize")
    eq(ops.thresholdImage(g, 40), "threshold40")
    otsu = ops.otsuThresholdImage(g)
    eq(otsu, "otsu")
    rng = ops.Randomness(W * H)
    assert np.array_equal(rng.x0.cpu().numpy().view(np.uint32), d["x0"])
    assert np.array_equal(rng.x1.cpu().numpy().view(np.uint32), d["x1"])
    eq(ops.saltAndPepperNoise(g, rng, 50), "salt_pepper50")
    eq(ops.addGaussianNoise(g, rng, -1.0, -1.0), "gauss_auto")
    eq(ops.addGaussianNoise(g, rng, 100.0, 25.0), "gauss_100_25")
    eq(ops.linearFilter(g, K_INPUT, 3, 3), "linear_input")
    eq(ops.linearFilter(g, K_LINEAR, 3, 3), "linear_neg")
    eq(ops.linearFilter(g, K_G16, 3, 3), "linear_g16")
    eq(ops.linearFilter(g, K_BOX9, 3, 3), "linear_box9")
    eq(ops.averageFilter(g, K_INPUT, 3, 3), "average_input")
    eq(ops.linearFilter(g, [1.0 / 121] * 121, 11, 11), "linear_box11")
    eq(ops.medianFilter(g, ONES, 3, 3), "median3")
    eq(ops.medianFilter(g, [1] * 49, 7, 7), "median7")
    eq(ops.sobelFilter(g), "sobel")
    eq(ops.compassFilter(g), "compass")
    eq(ops.imageDilation(otsu, CROSS, 3, 3), "dilate_cross")
    eq(ops.imageErosion(otsu, CROSS, 3, 3), "erode_cross")
    eq(ops.imageDilation(otsu, ONES, 3, 3), "dilate_ones")
    eq(ops.imageErosion(g, ONES, 3, 3), "erode_ones_gray")
    cent = torch.tensor([[215, 101]], dtype=torch.int32, device="cuda")  # the first two draws of an unseeded rand()
    eq(ops.kMeansThresholding(g, cent), "kmeans_215_101")
    assert tuple(ops.LibcRand().centroids(1).cpu().numpy().reshape(-1)) == (215, 101)
    eq(ops.featureExtraction(g), "features")


@pytest.mark.gpu
def test_gpu_pipeline_hits_golden():
    import torch
    from imageanalysis_amd.pipeline import Pipeline
    d = load("pipeline_PA_64x48.npz")
    ops = [{"type": O.OP_EQUALIZE}, {"type": O.OP_GAUSS, "mean": -1.0, "stddev": -1.0},
           {"type": O.OP_SALT_PEPPER, "iparam": 50}, {"type": O.OP_LINEAR, "kW": 3, "kH": 3, "fkernel": K_G16},
           {"type": O.OP_MEDIAN, "kW": 3, "kH": 3, "ikernel": ONES}, {"type": O.OP_SOBEL}]
    for fused in (True, False):
        p = Pipeline(64, 48, 2, ops, fused=fused)
        hs = torch.zeros(256, dtype=torch.int64, device="cuda")
        out = p.run(torch.from_numpy(d["rgb"]).cuda(), hist_sum=hs)
        assert np.array_equal(out.cpu().numpy(), d["out"])
        assert np.array_equal(hs.cpu().numpy(), d["hist_sum"])
