"""The HIP path against vectors made by the reference's own host-only functions
(tests/golden/ref_host.npz, see tests/test_ref_host_golden.py): the equalisation map, the
random centroid draws and the histogram features."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_host.npz"))


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from imageanalysis_amd import ops as o
    return o


def test_eq_map_kernel_matches_the_reference_function(ops):
    # iak_eq_map takes one numPixels per call: group the vectors by it
    hs, ns, want = G["eq_hist"], G["eq_num_pixels"], G["eq_map"]
    for n in np.unique(ns):
        sel = np.nonzero(ns == n)[0]
        got = ops.equalizeHistogram(torch.from_numpy(hs[sel]).cuda(), int(n)).cpu().numpy()
        assert np.array_equal(got, want[sel]), n


def test_centroid_draws_match_the_reference_function(ops):
    want = G["random_0_256_unseeded"]
    got = ops.LibcRand().centroids(len(want) // 2, 1).cpu().numpy().reshape(-1)
    assert np.array_equal(got, want)


def test_histogram_features_match_the_reference_functions(ops):
    """Images built to have the stored histograms: mode, rangeMin and rangeMax of iak_features are
    calcMode's and calcMinAndMaxOfRange's answers (imageMachineLearning.cu:341-351,379-391)."""
    done = 0
    for i, (h, n) in enumerate(zip(G["feat_hist"], G["feat_num_pixels"])):
        h = h.astype(np.int64)
        n = int(n)
        zeros = n - int(h.sum())
        if zeros < 0 or n == 0 or n % 16 or n > (1 << 22):
            continue
        vals = np.concatenate([np.zeros(zeros, np.uint8), np.repeat(np.arange(256, dtype=np.uint8), h)])
        np.random.default_rng(i).shuffle(vals)
        img = torch.from_numpy(vals.reshape(-1, 16)).cuda()
        f = ops.featureExtraction(img).cpu().numpy()
        assert (f[4], f[5], f[6]) == (G["feat_mode"][i], G["feat_range_min"][i], G["feat_range_max"][i]), i
        done += 1
    assert done >= 8
