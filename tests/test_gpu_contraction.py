"""The HIP path in contraction mode (iak_set_contraction) against the oracle in the same mode: the grayscale
conversion on every (R, G, B) triple, the filters that accumulate in float, and a whole pipeline.  The count of
differing triples is the figure quoted for how far the two possible nvcc builds of the reference are apart."""
import numpy as np
import pytest
import torch

import oracle_lib as O
import synth
from test_contraction import all_triples_image

pytestmark = pytest.mark.gpu

FRAC3 = [0.3, -0.2, 0.1, 0.7, 1.1, -0.4, 0.25, 0.05, 0.6]


@pytest.fixture()
def fused_mode():
    from imageanalysis_amd import lib as L
    h = L.load()
    assert h.iak_get_contraction() == 0
    h.iak_set_contraction(1)
    O.lib().iao_set_contraction(1)
    yield h
    h.iak_set_contraction(0)
    O.lib().iao_set_contraction(0)


def test_grayscale_on_every_triple(fused_mode):
    from imageanalysis_amd import ops
    rgb = all_triples_image(1)  # 16.7 M pixels
    dev = torch.from_numpy(rgb).cuda()
    fused = ops.convertRGBToGrayscale(dev).cpu().numpy()
    assert np.array_equal(fused, O.gray(rgb, 0))
    fused_mode.iak_set_contraction(0)
    O.lib().iao_set_contraction(0)
    plain = ops.convertRGBToGrayscale(dev).cpu().numpy()
    assert np.array_equal(plain, O.gray(rgb, 0))
    differ = plain != fused
    frac = float(differ.mean())
    assert 0.0005 < frac < 0.005, frac  # about 0.17 % of all triples
    assert np.all(np.abs(plain[differ].astype(int) - fused[differ].astype(int)) == 1)
    fused_mode.iak_set_contraction(1)  # the fixture switches both off again
    O.lib().iao_set_contraction(1)


@pytest.mark.parametrize("W,H", [(64, 48), (37, 23), (768, 40)])
def test_float_filters(fused_mode, W, H):
    from imageanalysis_amd import ops
    img = synth.gray_batch("U", W, H, 2)
    dev = torch.from_numpy(img).cuda()
    rng = np.random.default_rng(4)
    k5 = (rng.random(25) * 0.2 - 0.05).astype(np.float32).tolist()
    k11 = [1.0 / 121] * 121
    for kernel, kw, kh in ((FRAC3, 3, 3), (k5, 5, 5), (k11, 11, 11)):
        got = ops.linearFilter(dev, kernel, kw, kh).cpu().numpy()
        want = np.stack([O.linear(img[b], kernel, kw, kh) for b in range(2)])
        assert np.array_equal(got, want), (kw, kh)
        got = ops.averageFilter(dev, kernel, kw, kh).cpu().numpy()
        want = np.stack([O.average(img[b], kernel, kw, kh) for b in range(2)])
        assert np.array_equal(got, want), (kw, kh)


@pytest.mark.parametrize("fused_plan", [True, "pointwise", False])
def test_pipeline_keeps_the_setting_of_its_creation(fused_mode, fused_plan):
    from imageanalysis_amd.pipeline import Pipeline
    W, H, B = 128, 48, 3
    ops_list = [{"type": O.OP_EQUALIZE}, {"type": O.OP_GAUSS, "mean": -1.0, "stddev": -1.0}, {"type": O.OP_SALT_PEPPER, "iparam": 50},
                {"type": O.OP_LINEAR, "kW": 3, "kH": 3, "fkernel": FRAC3}, {"type": O.OP_AVERAGE, "kW": 5, "kH": 5, "fkernel": [0.11] * 25},
                {"type": O.OP_SOBEL}]
    rgb = synth.rgb_batch("C", W, H, B)
    x0, x1 = O.xorwow_planes(W * H)
    want, want_hist = O.pipeline(rgb, ops_list, x0, x1)
    p = Pipeline(W, H, B, ops_list, fused=fused_plan)
    fused_mode.iak_set_contraction(0)  # a run uses what the pipeline was created under
    hist = torch.zeros(256, dtype=torch.int64, device="cuda")
    got = p.run(torch.from_numpy(rgb).cuda(), hist_sum=hist).cpu().numpy()
    fused_mode.iak_set_contraction(1)
    assert np.array_equal(got, want)
    assert np.array_equal(hist.cpu().numpy(), want_hist)
    p.close()
