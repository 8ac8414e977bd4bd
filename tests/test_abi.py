"""The C-ABI library loads and exports exactly what include/iak.h declares (no compute:
this runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "iak.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(iak_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_expected_surface():
    names = declared_symbols()
    for must in ("iak_gray", "iak_gray_hist", "iak_hist", "iak_eq_map", "iak_apply_map", "iak_quantize",
                 "iak_rng_planes", "iak_salt_pepper", "iak_gauss_noise", "iak_linear", "iak_median", "iak_sobel",
                 "iak_dilate", "iak_erode", "iak_pipeline_create", "iak_pipeline_run"):
        assert must in names


def test_library_exports_every_declared_symbol():
    path = os.path.join(ROOT, "imageanalysis_amd", "lib", "libiak.so")
    assert os.path.exists(path), "libiak.so is not built (run __graft_entry__.build())"
    lib = ctypes.CDLL(path)
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, f"declared in iak.h but not exported: {missing}"


def test_python_binding_covers_the_header():
    from imageanalysis_amd import lib as L
    assert sorted(L.SIGNATURES) == declared_symbols()
    handle = L.load()
    assert handle.iak_version() >= 100
    assert handle.iak_error_string(L.IAK_E_UNSUPPORTED).decode().startswith("iak:")
    assert handle.iak_error_string(0) == b"success"


def test_argument_errors_need_no_gpu():
    from imageanalysis_amd import lib as L
    h = L.load()
    assert h.iak_hist(None, None, 0, 0, 0, None) == L.IAK_E_ARG
    assert h.iak_gray(None, None, 4, 4, 1, 48, 16, 0, None) == L.IAK_E_ARG
    assert h.iak_median(None, None, 4, 4, 1, 16, 16, None, 3, 3, None) == L.IAK_E_ARG
    assert h.iak_pipeline_create(None, 4, 4, 1, None, 0, 1234, 1) == L.IAK_E_ARG
    assert h.iak_features(None, 16, 16, 1, 256, None, None, None, None) == L.IAK_E_ARG
    assert h.iak_features_workspace_bytes(0, 16, 1) == L.IAK_E_ARG
    assert h.iak_features_workspace_bytes(768, 568, 500) < h.iak_features_workspace_bytes(769, 568, 500)
    assert h.iak_kmeans2_map(None, 16, 1, None, None, None, None) == L.IAK_E_ARG
    assert h.iak_knn_predict(None, None, 1, 1, 8, 3, None, None) == L.IAK_E_ARG
    assert h.iak_channel_hist(None, None, None, 4, 4, 1, 48, 16, 3, None) == L.IAK_E_ARG
    assert h.iak_rng_planes_from_state(None, None, 16, None, None) == L.IAK_E_ARG


def test_cpp_api_library_exports_reference_functions():
    path = os.path.join(ROOT, "imageanalysis_amd", "lib", "libimageAnalysis.so")
    assert os.path.exists(path)
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", "-C", path], capture_output=True, text=True).stdout
    for fn in ("copyHostImageToDevice", "copyDeviceRGBImageToHost", "convertRGBToGrayscale", "copyHostRGBImageToDevice",
               "copyDeviceImageToHost", "calculateHistogram", "equalizeHistogram", "equalizeImageWithHist",
               "linearFilter", "averageFilter", "medianFilter", "sobelFilter", "compassFilter", "saltAndPepperNoise",
               "addGaussianNoise", "cleanupEdgeDetection", "setupEdgeDetection", "cleanupRandomness", "imageDilation",
               "imageErosion", "thresholdImage", "otsuThresholdImage", "kMeansThresholding(Image*, Image*, int)",
               "kMeansThresholding(Image*, Image*)", "extractSingleColorChannel", "setupRandomness",
               "imageQuantization", "calcMSQE", "calculateMeanAndStdDev", "featureExtraction", "extractAllPixelsAsFeatures",
               "knn(int, int, float*, float*, int, int*, int)", "knnTenfoldCrossVal"):
        assert fn in out, fn


def test_product_package_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "imageanalysis_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.lower(), os.path.join(dirpath, f)
