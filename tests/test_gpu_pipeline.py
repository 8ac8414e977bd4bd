"""The batched executor (stage-wise and fused) against the oracle's pipeline, bit for bit,
on small batches; and the configured pipelines of the reference's JSON configs."""
import numpy as np
import pytest
import torch

import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu

GAUSS16 = [1 / 16, 2 / 16, 1 / 16, 2 / 16, 4 / 16, 2 / 16, 1 / 16, 2 / 16, 1 / 16]
ONES9 = [1] * 9
CROSS = [0, 1, 0, 1, 1, 1, 0, 1, 0]

# op dicts are shared by the executor and the oracle (same field names)
P_A = [
    {"type": O.OP_EQUALIZE},
    {"type": O.OP_GAUSS, "mean": -1.0, "stddev": -1.0},
    {"type": O.OP_SALT_PEPPER, "iparam": 50},
    {"type": O.OP_LINEAR, "kW": 3, "kH": 3, "fkernel": GAUSS16},
    {"type": O.OP_MEDIAN, "kW": 3, "kH": 3, "ikernel": ONES9},
    {"type": O.OP_SOBEL},
]
P_C = [
    {"type": O.OP_OTSU},
    {"type": O.OP_SOBEL},
    {"type": O.OP_DILATE, "kW": 3, "kH": 3, "ikernel": ONES9},
    {"type": O.OP_ERODE, "kW": 3, "kH": 3, "ikernel": CROSS},
]
INPUT_JSON = [  # test/configs/input.json:11-145 without kMeans / feature-extract
    {"type": O.OP_GAUSS, "mean": -1.0, "stddev": -1.0},
    {"type": O.OP_EQUALIZE},
    {"type": O.OP_QUANTIZE, "ikernel": [10, 20, 15, 30, 250, 77], "iparam": 2},
    {"type": O.OP_LINEAR, "kW": 3, "kH": 3, "fkernel": [1, 1, 1, 10, 1, 1, 1, 1, 1]},
    {"type": O.OP_AVERAGE, "kW": 3, "kH": 3, "fkernel": [1, 1, 1, 10, 1, 1, 1, 1, 1]},
    {"type": O.OP_SALT_PEPPER, "iparam": 50},
    {"type": O.OP_MEDIAN, "kW": 3, "kH": 3, "ikernel": ONES9},
    {"type": O.OP_SOBEL},
    {"type": O.OP_COMPASS},
    {"type": O.OP_THRESHOLD, "iparam": 40},
    {"type": O.OP_DILATE, "kW": 3, "kH": 3, "ikernel": CROSS},
    {"type": O.OP_ERODE, "kW": 3, "kH": 3, "ikernel": CROSS},
    {"type": O.OP_OTSU},
]
CHAINS = {
    "P-A": P_A,
    "P-C": P_C,
    "input.json": INPUT_JSON,
    "empty": [],
    "lut-run": [{"type": O.OP_EQUALIZE}, {"type": O.OP_QUANTIZE, "ikernel": [0, 100, 50, 100, 256, 200], "iparam": 2},
                {"type": O.OP_EQUALIZE}, {"type": O.OP_OTSU}],
    "noise-then-stats": [{"type": O.OP_SALT_PEPPER, "iparam": 7}, {"type": O.OP_EQUALIZE},
                         {"type": O.OP_GAUSS, "mean": -1.0, "stddev": 4.0}, {"type": O.OP_GAUSS, "mean": -1.0, "stddev": -1.0},
                         {"type": O.OP_OTSU}],
    "fixed-gauss": [{"type": O.OP_GAUSS, "mean": 120.0, "stddev": 30.0}, {"type": O.OP_THRESHOLD, "iparam": 100},
                    {"type": O.OP_THRESHOLD, "iparam": 10}, {"type": O.OP_THRESHOLD, "iparam": 200},
                    {"type": O.OP_SALT_PEPPER, "iparam": 3}, {"type": O.OP_SALT_PEPPER, "iparam": 50}],
    # a per-pixel run feeding each kind of 3x3 stencil (the executor stages the run's output in LDS)
    "sp+median": [{"type": O.OP_SALT_PEPPER, "iparam": 20}, {"type": O.OP_MEDIAN, "kW": 3, "kH": 3, "ikernel": ONES9}],
    "eq+sobel": [{"type": O.OP_EQUALIZE}, {"type": O.OP_SOBEL}],
    "gauss+compass": [{"type": O.OP_GAUSS, "mean": 10.0, "stddev": 25.0}, {"type": O.OP_COMPASS}],
    "otsu+dilate+erode": [{"type": O.OP_OTSU}, {"type": O.OP_DILATE, "kW": 3, "kH": 3, "ikernel": CROSS},
                          {"type": O.OP_THRESHOLD, "iparam": 100}, {"type": O.OP_ERODE, "kW": 3, "kH": 3, "ikernel": ONES9}],
    "quant+average": [{"type": O.OP_QUANTIZE, "ikernel": [0, 128, 40, 128, 256, 220], "iparam": 2},
                      {"type": O.OP_AVERAGE, "kW": 3, "kH": 3, "fkernel": [1.0] * 9},
                      {"type": O.OP_THRESHOLD, "iparam": 90},
                      {"type": O.OP_LINEAR, "kW": 3, "kH": 3, "fkernel": [0.3, -0.2, 0.1, 0.7, 1.1, -0.4, 0.25, 0.05, 0.6]}],
    # runs of 3x3 stencils
    "median+sobel": [{"type": O.OP_MEDIAN, "kW": 3, "kH": 3, "ikernel": ONES9}, {"type": O.OP_SOBEL}],
    "dilate+erode+compass+median+sobel": [
        {"type": O.OP_DILATE, "kW": 3, "kH": 3, "ikernel": ONES9}, {"type": O.OP_ERODE, "kW": 3, "kH": 3, "ikernel": CROSS},
        {"type": O.OP_COMPASS}, {"type": O.OP_MEDIAN, "kW": 3, "kH": 3, "ikernel": ONES9}, {"type": O.OP_SOBEL}],
    "erode+dilate+dilate": [
        {"type": O.OP_THRESHOLD, "iparam": 120}, {"type": O.OP_ERODE, "kW": 3, "kH": 3, "ikernel": ONES9},
        {"type": O.OP_DILATE, "kW": 3, "kH": 3, "ikernel": [1, 0, 0, 0, 1, 0, 0, 0, 1]},
        {"type": O.OP_DILATE, "kW": 3, "kH": 3, "ikernel": [0, 0, 1, 1, 1, 1, 1, 0, 0]}],
    # runs of 3x3 erosions / dilations (one bit-mask pass for up to three of them)
    "erode3x": [{"type": O.OP_ERODE, "kW": 3, "kH": 3, "ikernel": CROSS},
                {"type": O.OP_ERODE, "kW": 3, "kH": 3, "ikernel": [1, 0, 0, 0, 1, 1, 0, 0, 0]},
                {"type": O.OP_ERODE, "kW": 3, "kH": 3, "ikernel": [0, 0, 0, 0, 0, 0, 0, 1, 1]}],
    # found by fuzzing: pixel (1, 1), as the halo of row 0's last chunk, takes a tap from pixel (0, 0)
    "erode-dilate-erode": [{"type": O.OP_ERODE, "kW": 3, "kH": 3, "ikernel": [1, 0, 0, 0, 0, 0, 0, 1, 1]},
                           {"type": O.OP_DILATE, "kW": 3, "kH": 3, "ikernel": [0, 0, 0, 1, 0, 1, 1, 0, 0]},
                           {"type": O.OP_ERODE, "kW": 3, "kH": 3, "ikernel": [0, 0, 1, 1, 0, 0, 1, 0, 1]}],
    "morph-mix": [{"type": O.OP_THRESHOLD, "iparam": 150}, {"type": O.OP_SOBEL},
                  {"type": O.OP_DILATE, "kW": 3, "kH": 3, "ikernel": [1, 1, 0, 0, 0, 0, 0, 0, 1]},
                  {"type": O.OP_DILATE, "kW": 3, "kH": 3, "ikernel": [0, 0, 1, 0, 0, 0, 1, 0, 0]},
                  {"type": O.OP_ERODE, "kW": 3, "kH": 3, "ikernel": ONES9},
                  {"type": O.OP_ERODE, "kW": 3, "kH": 3, "ikernel": CROSS},
                  {"type": O.OP_DILATE, "kW": 3, "kH": 3, "ikernel": ONES9}],
    "open-close": [{"type": O.OP_MEDIAN, "kW": 3, "kH": 3, "ikernel": ONES9},
                   {"type": O.OP_ERODE, "kW": 3, "kH": 3, "ikernel": [0, 1, 0, 0, 1, 0, 0, 0, 0]},
                   {"type": O.OP_DILATE, "kW": 3, "kH": 3, "ikernel": [0, 1, 0, 0, 1, 0, 0, 0, 0]}],
    # one threshold in front of each border-zeroing 3x3 operation (done by table on the mask bits)
    "thresholded-stencils": [
        {"type": O.OP_THRESHOLD, "iparam": 100},
        {"type": O.OP_LINEAR, "kW": 3, "kH": 3, "fkernel": [0.3, -0.2, 0.1, 0.7, 0.11, -0.4, 0.25, 0.05, 0.06]},
        {"type": O.OP_THRESHOLD, "iparam": 50}, {"type": O.OP_AVERAGE, "kW": 3, "kH": 3, "fkernel": [0.5, 0.25, 0.5, 1, 2, 1, 0.5, 0.25, 0.5]},
        {"type": O.OP_OTSU}, {"type": O.OP_MEDIAN, "kW": 3, "kH": 3, "ikernel": ONES9},
        {"type": O.OP_THRESHOLD, "iparam": 0}, {"type": O.OP_COMPASS},
        {"type": O.OP_THRESHOLD, "iparam": 254}, {"type": O.OP_ERODE, "kW": 3, "kH": 3, "ikernel": CROSS},
        {"type": O.OP_THRESHOLD, "iparam": 255}, {"type": O.OP_SOBEL},
        {"type": O.OP_THRESHOLD, "iparam": -1}, {"type": O.OP_SOBEL}],
    "linears": [{"type": O.OP_LINEAR, "kW": 3, "kH": 3, "fkernel": [1, 1, 1, 10, 1, 1, 1, 1, 1]},
                {"type": O.OP_AVERAGE, "kW": 3, "kH": 3, "fkernel": [0.5, 0.25, 0.5, 1, 2, 1, 0.5, 0.25, 0.5]},
                {"type": O.OP_LINEAR, "kW": 3, "kH": 3, "fkernel": [0.3, -0.2, 0.1, 0.7, 1.1, -0.4, 0.25, 0.05, 0.6]},
                {"type": O.OP_AVERAGE, "kW": 3, "kH": 3, "fkernel": [1.0] * 9},
                {"type": O.OP_LINEAR, "kW": 3, "kH": 3, "fkernel": GAUSS16},
                {"type": O.OP_AVERAGE, "kW": 3, "kH": 3, "fkernel": [0.3, -0.2, 0.1, 0.7, 1.1, -0.4, 0.25, 0.05, 0.6]}],
    # a Sobel between a threshold and a dilation: only its being non-zero is looked at, on bit masks
    "thr+sobel+dilate": [{"type": O.OP_THRESHOLD, "iparam": 120}, {"type": O.OP_SOBEL},
                         {"type": O.OP_DILATE, "kW": 3, "kH": 3, "ikernel": CROSS}],
    "otsu+dilate+sobel+dilate": [{"type": O.OP_OTSU}, {"type": O.OP_DILATE, "kW": 3, "kH": 3, "ikernel": [1, 0, 0, 0, 1, 0, 0, 1, 1]},
                                 {"type": O.OP_SOBEL}, {"type": O.OP_DILATE, "kW": 3, "kH": 3, "ikernel": ONES9},
                                 {"type": O.OP_MEDIAN, "kW": 3, "kH": 3, "ikernel": ONES9}],
    "otsu+sobel+erode+dilate+sobel": [{"type": O.OP_OTSU}, {"type": O.OP_SOBEL}, {"type": O.OP_ERODE, "kW": 3, "kH": 3, "ikernel": CROSS},
                                      {"type": O.OP_DILATE, "kW": 3, "kH": 3, "ikernel": [0, 1, 1, 1, 1, 0, 0, 0, 1]}, {"type": O.OP_SOBEL}],
    "otsu+sobel+erode": [{"type": O.OP_OTSU}, {"type": O.OP_SOBEL}, {"type": O.OP_ERODE, "kW": 3, "kH": 3, "ikernel": ONES9}],
    "big-stencils": [{"type": O.OP_LINEAR, "kW": 11, "kH": 11, "fkernel": [1.0 / 121] * 121},
                     {"type": O.OP_MEDIAN, "kW": 7, "kH": 7, "ikernel": [1] * 49}],
}


@pytest.fixture(scope="module")
def Pipeline():
    assert torch.cuda.is_available()
    from imageanalysis_amd.pipeline import Pipeline as P
    return P


@pytest.mark.parametrize("name", list(CHAINS))
@pytest.mark.parametrize("fused", [False, "pointwise", True])
@pytest.mark.parametrize("W,H,kind", [(64, 48, "C"), (37, 23, "U"), (768, 40, "C"), (48, 203, "U"), (1056, 21, "U"),
                                      (16, 9, "U"), (1024, 43, "U"), (2048, 5, "C"), (8, 11, "U"), (40, 8, "C")])
def test_pipeline_matches_oracle(Pipeline, name, fused, W, H, kind):
    ops = CHAINS[name]
    B = 3
    rgb = synth.rgb_batch(kind, W, H, B)
    x0, x1 = O.xorwow_planes(W * H)
    want, want_hist = O.pipeline(rgb, ops, x0, x1)
    p = Pipeline(W, H, B + 1, ops, fused=fused)
    hist_sum = torch.zeros(256, dtype=torch.int64, device="cuda")
    gray_hist = torch.empty((B, 256), dtype=torch.int32, device="cuda")
    got = p.run(torch.from_numpy(rgb).cuda(), hist_sum=hist_sum, gray_hist=gray_hist).cpu().numpy()
    for b in range(B):
        if not np.array_equal(got[b], want[b]):
            bad = np.argwhere(got[b] != want[b])
            raise AssertionError(f"{name} fused={fused} image {b}: {len(bad)} mismatches, first {bad[0]}")
    assert np.array_equal(hist_sum.cpu().numpy(), want_hist)
    for b in range(B):
        assert np.array_equal(gray_hist[b].cpu().numpy(), O.hist(O.gray(rgb[b], 0)))
    # a second run accumulates the dataset histogram and reproduces the images
    got2 = p.run(torch.from_numpy(rgb).cuda(), hist_sum=hist_sum).cpu().numpy()
    assert np.array_equal(got2, got)
    assert np.array_equal(hist_sum.cpu().numpy(), 2 * want_hist)
    p.close()


@pytest.mark.parametrize("name", ["P-A", "P-C", "input.json", "sp+median", "big-stencils", "thr+sobel+dilate", "otsu+dilate+erode"])
@pytest.mark.parametrize("W,H,kind", [(768, 40, "C"), (64, 48, "U"), (1024, 43, "U")])
def test_pipeline_with_the_specialised_kernels_switched_off(Pipeline, monkeypatch, name, W, H, kind):
    """IAK_NO_LGS (read when a pipeline is planned) sends look-up / Gaussian / salt-pepper runs to the generic fused
    kernel, IAK_NO_BITSLICE_MEDIAN (read per call) sends small all-ones medians to the column-histogram kernel,
    IAK_NO_BITSLICE_BOX (read per call) sends box filters with sum-dependent taps to the integer-sum kernel,
    IAK_NO_MORPH_BITS (read per call and when a pipeline is planned) sends runs of erosions / dilations to the
    8-pixel-chunk kernel and keeps a Sobel out of them: the same images either way."""
    ops = CHAINS[name]
    rgb = synth.rgb_batch(kind, W, H, 2)
    x0, x1 = O.xorwow_planes(W * H)
    want, _ = O.pipeline(rgb, ops, x0, x1)
    dev = torch.from_numpy(rgb).cuda()
    p = Pipeline(W, H, 2, ops, fused=True)
    assert np.array_equal(p.run(dev).cpu().numpy(), want)
    p.close()
    monkeypatch.setenv("IAK_NO_LGS", "1")
    monkeypatch.setenv("IAK_NO_BITSLICE_MEDIAN", "1")
    monkeypatch.setenv("IAK_NO_BITSLICE_BOX", "1")
    monkeypatch.setenv("IAK_NO_MORPH_BITS", "1")
    p = Pipeline(W, H, 2, ops, fused=True)
    assert np.array_equal(p.run(dev).cpu().numpy(), want)
    p.close()


def test_pipeline_smaller_batch_than_planned(Pipeline):
    W, H = 64, 48
    rgb = synth.rgb_batch("C", W, H, 5)
    x0, x1 = O.xorwow_planes(W * H)
    p = Pipeline(W, H, 5, P_A, fused=True)
    full = p.run(torch.from_numpy(rgb).cuda()).cpu().numpy()
    part = p.run(torch.from_numpy(rgb[:2]).cuda()).cpu().numpy()
    want, _ = O.pipeline(rgb, P_A, x0, x1)
    assert np.array_equal(full, want) and np.array_equal(part, want[:2])


@pytest.mark.parametrize("lanes", [1, 2])
@pytest.mark.parametrize("name", ["P-A", "P-C", "lut-run", "noise-then-stats", "otsu+dilate+erode", "eq+sobel"])
def test_pipeline_histogram_tables_take_turns(Pipeline, lanes, name):
    """The executor's own histogram tables (grayscale image, intermediates) are cleared several at
    a time and handed out run by run: 21 runs (more than two rounds) on alternating batches, two
    of them on another stream, one with the caller's own grayscale histogram array; every result
    and the accumulated data-set histogram against the oracle."""
    W, H, B = 64, 48, 3
    ops = CHAINS[name]
    batches = [synth.rgb_batch("C", W, H, B), synth.rgb_batch("U", W, H, B, first_index=7)]
    x0, x1 = O.xorwow_planes(W * H)
    want = [O.pipeline(r, ops, x0, x1) for r in batches]
    dev = [torch.from_numpy(r).cuda() for r in batches]
    gh = torch.empty((B, 256), dtype=torch.int32, device="cuda")
    p = Pipeline(W, H, B, ops, fused=True)
    p.set_lanes(lanes)
    hs = torch.zeros(256, dtype=torch.int64, device="cuda")
    total = np.zeros(256, np.int64)
    side = torch.cuda.Stream()
    for i in range(21):
        k = (i * 5 // 3) & 1
        if i in (9, 10):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                got = p.run(dev[k], hist_sum=hs)
            torch.cuda.current_stream().wait_stream(side)
        elif i == 5:
            got = p.run(dev[k], hist_sum=hs, gray_hist=gh)
            for b in range(B):
                assert np.array_equal(gh[b].cpu().numpy(), O.hist(O.gray(batches[k][b], 0)))
        else:
            got = p.run(dev[k], hist_sum=hs)
        total += want[k][1]
        assert np.array_equal(got.cpu().numpy(), want[k][0]), f"run {i}"
    assert np.array_equal(hs.cpu().numpy(), total)
    p.close()


@pytest.mark.parametrize("lanes", [2, 3, 8])
@pytest.mark.parametrize("name", ["P-A", "input.json"])
def test_pipeline_lanes_do_not_change_results(Pipeline, lanes, name):
    W, H, B = 64, 48, 5
    ops = CHAINS[name]
    rgb = synth.rgb_batch("C", W, H, B)
    x0, x1 = O.xorwow_planes(W * H)
    want, want_hist = O.pipeline(rgb, ops, x0, x1)
    p = Pipeline(W, H, B, ops, fused=True)
    p.set_lanes(lanes)
    hs = torch.zeros(256, dtype=torch.int64, device="cuda")
    gh = torch.empty((B, 256), dtype=torch.int32, device="cuda")
    for _ in range(3):
        got = p.run(torch.from_numpy(rgb).cuda(), hist_sum=hs, gray_hist=gh)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(hs.cpu().numpy(), 3 * want_hist)
    for b in range(B):
        assert np.array_equal(gh[b].cpu().numpy(), O.hist(O.gray(rgb[b], 0)))


def test_pipeline_timings(Pipeline):
    W, H = 64, 48
    rgb = torch.from_numpy(synth.rgb_batch("C", W, H, 2)).cuda()
    p = Pipeline(W, H, 2, P_A, fused=True)
    p.set_timing(2)
    p.run(rgb)
    p.run(rgb)
    t = p.timings(1)
    names = [r[0] for r in t]
    assert names == ["gray+hist", "equalize+gauss+salt_pepper+linear", "median", "sobel"]
    assert [n for n, _ in Pipeline(W, H, 2, P_A, fused="pointwise").groups()][1:3] == ["equalize+gauss+salt_pepper", "linear"]
    # a width the fused chain + stencil pass does not take (not a multiple of 16) keeps one pass per stencil
    assert [n for n, _ in Pipeline(40, H, 2, P_A, fused=True).groups()][1:] == [
        "equalize+gauss+salt_pepper", "linear", "median", "sobel"]
    assert sum(r[2] for r in t) == 19  # stage-wise compulsory bytes per pixel of this pipeline
    assert all(r[1] >= 0 and 0 <= r[3] <= r[1] + 1e-3 for r in t)
    # events on every third run only: five runs leave two timed ones
    from imageanalysis_amd.lib import IakError
    p.set_timing(4, every=3)
    for _ in range(5):
        p.run(rgb)
    assert len(p.timings(1)) == len(t)
    with pytest.raises(IakError):
        p.timings(2)


P_C_FEATURES = P_C + [{"type": O.OP_FEATURES}]
KM_CHAINS = {
    "P-C+features": P_C_FEATURES,
    "kmeans": [{"type": O.OP_KMEANS}],
    "kmeans-mix": [{"type": O.OP_EQUALIZE}, {"type": O.OP_KMEANS}, {"type": O.OP_FEATURES},
                   {"type": O.OP_MEDIAN, "kW": 3, "kH": 3, "ikernel": ONES9}, {"type": O.OP_GAUSS, "mean": 5.0, "stddev": 40.0},
                   {"type": O.OP_KMEANS}, {"type": O.OP_COMPASS}, {"type": O.OP_FEATURES}],
    "features-first": [{"type": O.OP_FEATURES}, {"type": O.OP_SOBEL}],
}


@pytest.mark.parametrize("name", list(KM_CHAINS))
@pytest.mark.parametrize("fused", [False, "pointwise", True])
@pytest.mark.parametrize("W,H,kind,lanes", [(64, 48, "C", 1), (37, 23, "U", 1), (768, 40, "C", 3)])
def test_pipeline_kmeans_and_features(Pipeline, name, fused, W, H, kind, lanes):
    """k-means draws its centroids from the libc rand() stream, two per image per k-means
    operation in image order, continuing from one run to the next; feature-extract leaves the
    image alone and reports the seven features of what it sees."""
    ops = KM_CHAINS[name]
    B = 5
    nk = sum(1 for o in ops if o["type"] == O.OP_KMEANS)
    rgb = synth.rgb_batch(kind, W, H, B)
    x0, x1 = O.xorwow_planes(W * H)
    p = Pipeline(W, H, B, ops, fused=fused)
    p.set_lanes(lanes)
    feats = torch.full((B, 7), -7.0, dtype=torch.float32, device="cuda")
    p.bind_features(feats)
    for run in range(2):
        cent = O.libc_random_0_256(2 * nk * B, skip=run * 2 * nk * B).reshape(B, max(nk, 1), 2) if nk else None
        want, want_hist, want_feats = O.pipeline_ex(rgb, ops, x0, x1, centroids=cent)
        got = p.run(torch.from_numpy(rgb).cuda()).cpu().numpy()
        for b in range(B):
            if not np.array_equal(got[b], want[b]):
                bad = np.argwhere(got[b] != want[b])
                raise AssertionError(f"{name} fused={fused} run {run} image {b}: {len(bad)} mismatches, first {bad[0]}")
        if any(o["type"] == O.OP_FEATURES for o in ops):
            assert np.array_equal(feats.cpu().numpy(), want_feats)
    p.close()


def test_pipeline_features_need_a_bound_array(Pipeline):
    from imageanalysis_amd.lib import IakError
    p = Pipeline(64, 48, 2, [{"type": O.OP_FEATURES}])
    with pytest.raises(IakError):
        p.run(torch.zeros((2, 3, 48, 64), dtype=torch.uint8, device="cuda"))


@pytest.mark.parametrize("what,code", [("average", 1), ("red", 2), ("green", 3), ("blue", 4)])
@pytest.mark.parametrize("W,H", [(64, 48), (37, 23)])
def test_pipeline_input_modes(Pipeline, what, code, W, H):
    """The driver's extract_channel: a colour plane (or the averaging grayscale) instead of the
    luminance grayscale as the pipeline's input; the histograms follow."""
    B = 3
    rgb = synth.rgb_batch("U", W, H, B)
    x0, x1 = O.xorwow_planes(W * H)
    want, want_hist, _ = O.pipeline_ex(rgb, P_A, x0, x1, input=code)
    p = Pipeline(W, H, B, P_A)
    p.set_input(what)
    hs = torch.zeros(256, dtype=torch.int64, device="cuda")
    gh = torch.empty((B, 256), dtype=torch.int32, device="cuda")
    got = p.run(torch.from_numpy(rgb).cuda(), hist_sum=hs, gray_hist=gh).cpu().numpy()
    assert np.array_equal(got, want)
    assert np.array_equal(hs.cpu().numpy(), want_hist)
    first = rgb[:, code - 2] if code >= 2 else None
    if first is not None:
        for b in range(B):
            assert np.array_equal(gh[b].cpu().numpy(), O.hist(first[b]))


@pytest.mark.parametrize("name", ["P-A", "P-C", "sp+median", "otsu+dilate+erode", "thresholded-stencils", "erode3x", "empty"])
@pytest.mark.parametrize("off_out,off_rgb", [(1, 0), (7, 3), (8, 0), (0, 5)])
def test_pipeline_takes_buffers_of_any_alignment(Pipeline, name, off_out, off_rgb):
    """The fused passes want 16-byte aligned images; a caller's buffers that are not (slices of a
    larger allocation) must still give the same results: the executor stages what it has to."""
    ops = CHAINS[name]
    W, H, B = 64, 48, 3
    rgb = synth.rgb_batch("C", W, H, B)
    x0, x1 = O.xorwow_planes(W * H)
    want, want_hist = O.pipeline(rgb, ops, x0, x1)
    big_in = torch.zeros(rgb.size + 32, dtype=torch.uint8, device="cuda")
    big_out = torch.full((B * H * W + 32,), 0xAB, dtype=torch.uint8, device="cuda")
    src = big_in[off_rgb:off_rgb + rgb.size].view(B, 3, H, W)
    src.copy_(torch.from_numpy(rgb).cuda())
    dst = big_out[off_out:off_out + B * H * W].view(B, H, W)
    assert src.data_ptr() % 16 == off_rgb % 16 and dst.data_ptr() % 16 == off_out % 16
    for fused in (True, "pointwise", False):
        p = Pipeline(W, H, B, ops, fused=fused)
        hist_sum = torch.zeros(256, dtype=torch.int64, device="cuda")
        p.run(src, out=dst, hist_sum=hist_sum)
        torch.cuda.synchronize()
        assert np.array_equal(dst.cpu().numpy(), want), (name, fused)
        assert np.array_equal(hist_sum.cpu().numpy(), want_hist)
        # nothing outside the output range was written
        assert int((big_out[:off_out] != 0xAB).sum()) == 0 and int((big_out[off_out + B * H * W:] != 0xAB).sum()) == 0
        p.close()
