"""BASELINE.json's full sizes.  A few images of each configuration are compared with the oracle
directly; the whole batch is checked through size-independent properties (histogram totals,
lookup-table structure, order relations between morphology results, idempotence)."""
import numpy as np
import pytest
import torch

import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu

GAUSS16 = [1 / 16, 2 / 16, 1 / 16, 2 / 16, 4 / 16, 2 / 16, 1 / 16, 2 / 16, 1 / 16]
ONES9 = [1] * 9
P_A = [
    {"type": O.OP_EQUALIZE}, {"type": O.OP_GAUSS, "mean": -1.0, "stddev": -1.0}, {"type": O.OP_SALT_PEPPER, "iparam": 50},
    {"type": O.OP_LINEAR, "kW": 3, "kH": 3, "fkernel": GAUSS16}, {"type": O.OP_MEDIAN, "kW": 3, "kH": 3, "ikernel": ONES9},
    {"type": O.OP_SOBEL},
]


@pytest.fixture(scope="module")
def mods():
    assert torch.cuda.is_available()
    from imageanalysis_amd import ops
    from imageanalysis_amd.pipeline import Pipeline
    return ops, Pipeline


def test_config2_batch_500_of_768x568(mods):
    """Config 2: the bench's batch shape.  Images 0, 249 and 499 against the oracle, the rest
    through the dataset histogram (every pixel of every image is counted exactly once)."""
    ops, Pipeline = mods
    import bench
    W, H, B = 768, 568, 500
    bench.W, bench.H = W, H
    rgb = bench.make_cell_batch(B, torch.device("cuda"), seed=77)
    p = Pipeline(W, H, B, P_A)
    hist_sum = torch.zeros(256, dtype=torch.int64, device="cuda")
    gray_hist = torch.empty((B, 256), dtype=torch.int32, device="cuda")
    out = p.run(rgb, hist_sum=hist_sum, gray_hist=gray_hist)
    torch.cuda.synchronize()
    x0, x1 = O.xorwow_planes(W * H)
    pick = [0, 249, 499]
    want, _ = O.pipeline(rgb[pick].cpu().numpy(), P_A, x0, x1, threads=3)
    assert np.array_equal(out[pick].cpu().numpy(), want)
    # histogram of the grayscale images: per image it misses only the zeros, and the batch sum is the sum of the rows
    gh = gray_hist.cpu().numpy().astype(np.int64)
    assert np.array_equal(gh.sum(axis=0), hist_sum.cpu().numpy())
    gray = ops.convertRGBToGrayscale(rgb)
    zeros = (gray == 0).sum(dim=(1, 2)).cpu().numpy()
    assert np.array_equal(gh.sum(axis=1) + zeros, np.full(B, W * H))
    assert np.all(gh[:, 0] == 0)
    # Sobel output: top row and left column are border zeros in every image
    assert int(out[:, 0, :].max()) == 0 and int(out[:, :, 0].max()) == 0


def test_config2_properties_on_the_whole_batch(mods):
    ops, _ = mods
    import bench
    W, H, B = 768, 568, 500
    bench.W, bench.H = W, H
    rgb = bench.make_cell_batch(B, torch.device("cuda"), seed=78)
    gray = ops.convertRGBToGrayscale(rgb)
    # equalisation maps are non-decreasing and the equalised image is the map applied
    hist = ops.calculateHistogram(gray)
    m = ops.equalizeHistogram(hist, W * H)
    assert bool((m[:, 1:] >= m[:, :-1]).all()) and int(m.max()) <= 255 and int(m.min()) >= 0
    eq = ops.equalizeImageWithHist(gray, m)
    b = 123
    assert torch.equal(eq[b], m[b].to(torch.uint8)[gray[b].long()])
    # thresholding is idempotent; Otsu's mask is a threshold mask of the same image
    mask, thr = ops.otsuThresholdImage(gray, return_threshold=True)
    assert torch.equal(ops.thresholdImage(mask, 0), mask)
    assert torch.equal(mask[b], ((gray[b].int() > int(thr[b])) * 255).to(torch.uint8))
    # morphology order: erosion keeps or clears pixels, dilation of a mask covers the interior of the mask
    er = ops.imageErosion(mask, ONES9, 3, 3)
    assert bool(((er == 0) | (er == mask)).all())
    di = ops.imageDilation(mask, ONES9, 3, 3)
    assert bool((di[:, 1:-1, 1:-1] >= mask[:, 1:-1, 1:-1]).all())
    # the 3x3 median of a two-valued image stays two-valued; of a constant interior it is the constant
    med = ops.medianFilter(mask, ONES9, 3, 3)
    assert bool(((med == 0) | (med == 255)).all())
    const = torch.full((4, H, W), 77, dtype=torch.uint8, device="cuda")
    assert int(ops.medianFilter(const, ONES9, 3, 3)[:, 1:, 1:].min()) == 77
    # noise is the same field for every image: where salt-and-pepper changed image 0 it changed image b the same way
    rng = ops.Randomness(W * H)
    sp = ops.saltAndPepperNoise(gray, rng, 50)
    changed = (sp[0] != gray[0]) | (sp[b] != gray[b])
    hit = ((sp[0] == 0) | (sp[0] == 255)) & ((sp[b] == 0) | (sp[b] == 255))
    assert bool((hit | ~changed).all())
    frac = float(((sp == 0) | (sp == 255)).float().mean())
    assert 0.015 < frac < 0.03          # 1 / 50.999999 of the pixels, plus the few that were 0 / 255 already


def test_config4_4096_square_large_windows(mods):
    """Config 4: one 4096x4096 image against the oracle (11x11 box linear filter, 7x7 median);
    the median's idempotence on its own two-valued output for the rest of the batch."""
    ops, Pipeline = mods
    W = H = 4096
    rgb = synth.rgb_batch("C", W, H, 1)
    ops_pb = [{"type": O.OP_LINEAR, "kW": 11, "kH": 11, "fkernel": [1.0 / 121] * 121},
              {"type": O.OP_MEDIAN, "kW": 7, "kH": 7, "ikernel": [1] * 49}]
    p = Pipeline(W, H, 2, ops_pb)
    both = torch.from_numpy(np.concatenate([rgb, rgb[:, :, ::-1, :].copy()])).cuda()
    got = p.run(both).cpu().numpy()
    want, _ = O.pipeline(rgb, ops_pb, threads=1)
    assert np.array_equal(got[0], want[0])
    # a vertically flipped image is a different input: its result must differ from the first, and
    # re-running reproduces it
    assert not np.array_equal(got[1], got[0])
    assert np.array_equal(p.run(both).cpu().numpy(), got)


def test_config4_batch_8_of_4096_square(mods):
    """Config 4 at the bench's size: 8 images of 4096x4096 through gray -> 11x11 box (1/121) -> 7x7 median.
    The first and the last image against the oracle; the whole batch again in reverse order, which must give
    the same images in reverse order (no image's result depends on its place in the batch)."""
    ops, Pipeline = mods
    import bench
    W = H = 4096
    B = 8
    old = bench.W, bench.H
    bench.W, bench.H = W, H
    try:
        rgb = bench.make_cell_batch(B, torch.device("cuda"), seed=404)
    finally:
        bench.W, bench.H = old
    p = Pipeline(W, H, B, bench.P_B)
    got = p.run(rgb).clone()
    pick = [0, B - 1]
    want, _ = O.pipeline(rgb[pick].cpu().numpy(), bench.P_B, threads=2)
    assert np.array_equal(got[pick].cpu().numpy(), want)
    back = p.run(rgb.flip(0).contiguous())
    assert torch.equal(back.flip(0), got)
    # the filters' borders: 5 rows / columns of zeros from the box filter's rule, of which the median keeps 3 and
    # may fill the rest with zeros or small values; the first three rows and columns are zero in every image
    assert int(got[:, :3, :].max()) == 0 and int(got[:, :, :3].max()) == 0
    p.close()


def test_config5_batch_500_segmentation_path_with_features(mods):
    """Config 5 at the bench's size: 500 images of 768x568 through gray -> Otsu -> Sobel -> 3x3 dilation ->
    3x3 erosion (cross) -> 7 features per image, with the per-class statistics the multi-GPU job reduces.
    Every image, every feature row, the data-set histogram and the packed class sums against the oracle
    (imageMachineLearning.cu:393-422, test/test.cu:541-550)."""
    import os
    ops, Pipeline = mods
    import bench
    from imageanalysis_amd.dist import PackedDatasetStats
    W, H, B, C = 768, 568, 500, bench.N_CLASSES
    bench.W, bench.H = W, H
    rgb = bench.make_cell_batch(B, torch.device("cuda"), seed=505)
    p = Pipeline(W, H, B, bench.P_C)
    feats = torch.zeros((B, 7), dtype=torch.float32, device="cuda")
    p.bind_features(feats)
    hist = torch.zeros(256, dtype=torch.int64, device="cuda")
    out = p.run(rgb, hist_sum=hist)
    labels = (torch.arange(B, device="cuda", dtype=torch.int64) % C).to(torch.int32)
    packed = PackedDatasetStats(C, torch.device("cuda"))
    packed.fill(hist, feats, labels)
    torch.cuda.synchronize()
    threads = min(16, os.cpu_count() or 1)
    x0, x1 = O.xorwow_planes(W * H)
    image_ops = [o for o in bench.P_C if o["type"] != bench.OP_FEATURES]
    want_img, want_hist = O.pipeline(rgb.cpu().numpy(), image_ops, x0, x1, threads=threads)
    got_img = out.cpu().numpy()
    assert np.array_equal(got_img, want_img)
    assert np.array_equal(packed.histogram().cpu().numpy(), want_hist)
    pick = sorted(set([0, 249, 499] + list(range(7, B, 31))))
    want = np.stack([O.features(want_img[b]) for b in pick]).astype(np.float64)
    got = feats.cpu().numpy().astype(np.float64)
    assert np.array_equal(got[pick], want)
    # the class sums are sums of the feature rows the GPU wrote (all 500 of them, oracle-checked on `pick`)
    lab = labels.cpu().numpy()
    count, mean, std = (t.cpu().numpy() for t in packed.class_stats())
    for c in range(C):
        sel = got[lab == c]
        assert count[c] == len(sel)
        assert np.allclose(mean[c], sel.mean(axis=0), rtol=1e-12, atol=1e-9)
        assert np.allclose(std[c], sel.std(axis=0), rtol=1e-6, atol=1e-4)
    # a binary image comes out: erosion of a dilated mask
    assert set(np.unique(got_img[::50]).tolist()) <= {0, 255}
    p.close()


def test_grayscale_pass_when_lanes_walk_more_than_two_chunks(mods):
    """A launch is capped at 16K workgroups; past that every lane of the grayscale pass walks further
    than the two chunks it is sized for.  3000 images of 768x284 (lanes take up to three chunks, the
    last pixels of each image fall to the scalar tail: 768*284 + 5 per image would not be a whole image,
    so the odd tail comes from a second call on 37x29 images): every pixel against the same float64
    expression evaluated by torch on the device (one rounding per operation, as the reference's), the
    per-image histograms against a count of the result."""
    ops, _ = mods
    from imageanalysis_amd import lib as L
    lib = L.load()
    g = torch.Generator(device="cuda")
    g.manual_seed(99)
    for (W, H, B) in [(768, 284, 3000), (37, 29, 2000)]:
        N = W * H
        rgb = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, device="cuda", generator=g)
        gray = torch.empty((B, H, W), dtype=torch.uint8, device="cuda")
        hist = torch.empty((B, 256), dtype=torch.int32, device="cuda")
        L.check(lib.iak_gray_hist(rgb.data_ptr(), gray.data_ptr(), hist.data_ptr(), W, H, B, 3 * N, N, 0,
                                  torch.cuda.current_stream().cuda_stream), "iak_gray_hist")
        want = torch.empty_like(gray)
        for s in range(0, B, 500):
            r, gg, b = (rgb[s:s + 500, c].to(torch.float64) for c in range(3))
            want[s:s + 500] = ((0.21 * r + 0.72 * gg) + 0.07 * b).to(torch.int32).to(torch.uint8)
        assert torch.equal(gray, want)
        # histogram of image b = counts of its non-zero values
        flat = want.view(B, N).to(torch.int64) + 256 * torch.arange(B, device="cuda").view(B, 1)
        counts = torch.bincount(flat.view(-1), minlength=256 * B).view(B, 256)
        counts[:, 0] = 0
        assert torch.equal(hist.to(torch.int64), counts)
        # and the oracle on three of the images
        for i in (0, B // 2, B - 1):
            assert np.array_equal(gray[i].cpu().numpy(), O.gray(rgb[i].cpu().numpy(), 0))
        del rgb, gray, hist, want, flat, counts
        torch.cuda.empty_cache()
