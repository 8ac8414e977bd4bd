"""The oracle against vectors made by the REFERENCE'S OWN code: tests/golden/ref_host.npz holds
inputs and the outputs of the reference's host-only functions, compiled unmodified from its
sources (oracle/ref_host/, tests/golden/make_ref_host_golden.py).  These rows of the oracle are
reference-pinned; everything that lives in a CUDA kernel stays a restatement."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_host.npz"))
i32p = np.ctypeslib.ndpointer(np.int32, flags="C")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C")


@pytest.fixture(scope="module")
def L():
    L = O.lib()
    L.iao_centroids_changed.argtypes = [u8p, C.c_int]
    L.iao_random_from_rand.argtypes = [C.c_int, C.c_int, C.c_int]
    L.iao_array_sum.argtypes = [i32p, C.c_int]
    L.iao_hist_mode.argtypes = [i32p]
    L.iao_hist_area.argtypes = [i32p, C.c_int]
    L.iao_hist_perimeter.argtypes = [i32p]
    L.iao_hist_range.argtypes = [i32p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.iao_confusion_metrics.argtypes = [i32p, f32p, C.c_int, C.c_int, i32p, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                        C.POINTER(C.c_float)]
    return L


def test_equalisation_map_matches_the_reference_function():
    # equalizeHistogram, imageAnalysis.cu:155-176: the float32 running sum, bit for bit
    for h, n, want in zip(G["eq_hist"], G["eq_num_pixels"], G["eq_map"]):
        assert np.array_equal(O.eq_map(h, int(n)), want)
    # the hand-derived known answer is among them
    kat = G["eq_map"][-1]
    assert kat[9] == 0 and kat[10] == 63 and kat[199] == 127 and kat[255] == 255


def test_random_matches_the_reference_function(L):
    # random(), imageAnalysis.cu:39-41, on the C library's rand() stream
    want = G["random_0_256_unseeded"]
    assert np.array_equal(O.libc_random_0_256(len(want)), want)
    assert tuple(want[:6]) == (215, 101, 201, 205, 234, 50)
    libc = C.CDLL("libc.so.6")
    libc.rand.restype = C.c_int
    libc.srand(20261111)
    got = [L.iao_random_from_rand(3, 17, libc.rand()) for _ in range(len(G["random_3_17_seed_20261111"]))]
    assert np.array_equal(got, G["random_3_17_seed_20261111"])


def test_centroid_update_matches_the_reference_function(L):
    for cin, cout, chg, k in zip(G["cent_in"], G["cent_out"], G["cent_changed"], G["cent_k"]):
        work = cin.copy()
        assert L.iao_centroids_changed(work, int(k)) == chg
        assert np.array_equal(work, cout)
    assert 0 < int(G["cent_changed"].sum()) < len(G["cent_changed"])  # both outcomes are in the file


def test_array_sum_matches_the_reference_function(L):
    for a, n, want in zip(G["sum_in"], G["sum_len"], G["sum_out"]):
        assert L.iao_array_sum(np.ascontiguousarray(a[:n]), int(n)) == want


def test_histogram_features_match_the_reference_functions(L):
    # calcMode, calculateCellArea, calculateCellPerimeter, calcMinAndMaxOfRange (imageMachineLearning.cu:341-391)
    for i, (h, n) in enumerate(zip(G["feat_hist"], G["feat_num_pixels"])):
        h = np.ascontiguousarray(h)
        lo, hi = C.c_int(), C.c_int()
        L.iao_hist_range(h, C.byref(lo), C.byref(hi))
        assert L.iao_hist_mode(h) == G["feat_mode"][i]
        assert L.iao_hist_area(h, int(n)) == G["feat_area"][i]
        assert L.iao_hist_perimeter(h) == G["feat_perimeter"][i]
        assert (lo.value, hi.value) == (G["feat_range_min"][i], G["feat_range_max"][i])
    assert -1 in G["feat_mode"] and -1 in G["feat_range_min"]  # the all-zero image is among them


def test_whole_image_features_agree_with_the_reference_functions():
    """iao_features on an image built to have a given histogram: mode and range are the reference
    functions' answers for that histogram."""
    for i in (0, 3, 12, 20, 21, 22):
        h = G["feat_hist"][i].astype(np.int64)
        n = int(G["feat_num_pixels"][i])
        zeros = n - int(h.sum())
        if zeros < 0 or n == 0:
            continue
        vals = np.concatenate([np.zeros(zeros, np.uint8), np.repeat(np.arange(256, dtype=np.uint8), h)])
        W = 16
        pad = (-len(vals)) % W
        if pad:
            continue
        img = vals.reshape(-1, W)
        f = O.features(img)
        assert f[4] == G["feat_mode"][i] and f[5] == G["feat_range_min"][i] and f[6] == G["feat_range_max"][i]


def test_cross_validation_metrics_match_the_reference_functions(L):
    assert int(G["cm_num_classes"][0]) == 6
    for i, n in enumerate(G["cm_n"]):
        n = int(n)
        cm = np.zeros(36, np.int32)
        acc, prec, rec = C.c_float(), C.c_float(), C.c_float()
        L.iao_confusion_metrics(np.ascontiguousarray(G["cm_pred"][i][:n]), np.ascontiguousarray(G["cm_rows"][i][:n]), 8, n, cm,
                                C.byref(acc), C.byref(prec), C.byref(rec))
        assert np.array_equal(cm, G["cm_confusion"][i])
        assert np.float32(acc.value) == G["cm_accuracy"][i]
        assert np.float32(prec.value) == G["cm_precision"][i]
        assert np.float32(rec.value) == G["cm_recall"][i]


def test_cpp_api_host_equalize_histogram_matches_the_reference_function():
    """equalizeHistogram of the C++ API layer is host code here as in the reference
    (imageAnalysis.cu:155-176); no GPU is touched by calling it."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "imageanalysis_amd", "lib", "libimageAnalysis.so")
    lib = C.CDLL(so)
    fn = getattr(lib, "_Z17equalizeHistogramPiS_i")
    fn.argtypes = [i32p, i32p, C.c_int]
    fn.restype = None
    for h, n, want in zip(G["eq_hist"], G["eq_num_pixels"], G["eq_map"]):
        m = np.zeros(256, np.int32)
        fn(np.ascontiguousarray(h), m, int(n))
        assert np.array_equal(m, want)
