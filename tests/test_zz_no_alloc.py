"""iak_features_static (the entry point that allocates nothing: its scratch is a fixed array
inside the library) against iak_features, through the C ABI: two calls in a row on one GPU,
the second with a smaller input, so that what the first call left in the shared scratch
must not leak into the second."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu


def _features_ws(L, x, hist):
    B, H, W = x.shape
    nbytes = L.iak_features_workspace_bytes(W, H, B)
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device="cuda")
    off = (-ws.data_ptr()) % 256
    out = torch.full((B, 7), -7.0, dtype=torch.float32, device="cuda")
    rc = L.iak_features(x.data_ptr(), W, H, B, W * H, hist.data_ptr() if hist is not None else None, out.data_ptr(),
                        ws.data_ptr() + off, None)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _features_static(L, x, hist):
    B, H, W = x.shape
    out = torch.full((B, 7), -9.0, dtype=torch.float32, device="cuda")
    rc = L.iak_features_static(x.data_ptr(), W, H, B, W * H, hist.data_ptr() if hist is not None else None, out.data_ptr(), None)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_static_features_match_workspace_features_twice():
    from imageanalysis_amd import lib as iak_lib
    from imageanalysis_amd import ops
    L = iak_lib.load()
    assert L.iak_features_static_max_images() == 1024
    # first call: a larger batch of larger images; second call: fewer, smaller images
    for (W, H, B, kind) in [(768, 40, 9, "C"), (64, 48, 4, "U")]:
        imgs = synth.gray_batch(kind, W, H, B)
        imgs[1] = 0
        x = torch.from_numpy(imgs).cuda()
        for with_hist in (False, True):
            hist = ops.calculateHistogram(x) if with_hist else None
            want = _features_ws(L, x, hist)
            got = _features_static(L, x, hist)
            assert np.array_equal(got, want), (W, H, B, with_hist)
        for b in range(B):
            assert np.array_equal(got[b], O.features(imgs[b]))


def test_static_features_limits():
    from imageanalysis_amd import lib as iak_lib
    L = iak_lib.load()
    x = torch.zeros((2, 16, 32), dtype=torch.uint8, device="cuda")
    out = torch.zeros((1025, 7), dtype=torch.float32, device="cuda")
    # more images than the fixed scratch holds
    assert L.iak_features_static(x.data_ptr(), 32, 16, 1025, 0, None, out.data_ptr(), None) == iak_lib.IAK_E_ARG  # stride < N
    big = torch.zeros((1025, 16, 32), dtype=torch.uint8, device="cuda")
    assert L.iak_features_static(big.data_ptr(), 32, 16, 1025, 512, None, out.data_ptr(), None) == iak_lib.IAK_E_TOO_LARGE
    # a width the one-pass form does not take
    y = torch.zeros((2, 16, 37), dtype=torch.uint8, device="cuda")
    assert L.iak_features_static(y.data_ptr(), 37, 16, 2, 37 * 16, None, out.data_ptr(), None) == iak_lib.IAK_E_UNSUPPORTED
    # the largest batch it takes
    got = _features_static(L, big[:1024], None)
    assert np.array_equal(got[0], O.features(np.zeros((16, 32), np.uint8)))
    assert np.array_equal(got[1023], got[0])
