"""The C++ imageAnalysis.h API layer (libimageAnalysis.so), driven by a small C++ program the
way the reference's driver uses it, against the oracle."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "imageanalysis_amd", "lib", "api_check")


def _reference_centroids():
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)
    rmax = 2147483647
    return [(0 + libc.rand() // (rmax // (256 - 0 + 1) + 1)) & 0xff for _ in range(2)]


@pytest.mark.parametrize("W,H,kind", [(768, 568, "C"), (37, 23, "U"), (64, 48, "K")])
def test_cpp_api(tmp_path, W, H, kind):
    assert os.path.exists(EXE), "build first: make -C imageanalysis_amd/csrc"
    rgb = synth.rgb_image(kind, W, H, 3)
    raw = tmp_path / "rgb.raw"
    rgb.tofile(raw)
    out = tmp_path / "out"
    out.mkdir()
    r = subprocess.run([EXE, str(raw), str(W), str(H), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr

    def got(name):
        return np.fromfile(out / f"{name}.u8", np.uint8).reshape(H, W)

    g = O.gray(rgb, 0)
    x0, x1 = O.xorwow_planes(W * H)
    ones, cross = [1] * 9, [0, 1, 0, 1, 1, 1, 0, 1, 0]
    k1 = [1, 1, 1, 10, 1, 1, 1, 1, 1]
    k2 = [1 / 16, 2 / 16, 1 / 16, 2 / 16, 4 / 16, 2 / 16, 1 / 16, 2 / 16, 1 / 16]
    otsu, _ = O.otsu(g)
    c0, c1 = _reference_centroids()
    want = {
        "gray": g, "gray_avg": O.gray(rgb, 1), "green": rgb[1],
        "equalize": O.equalize(g), "quantize": O.quantize(g, [10, 20, 15, 30, 250, 77]),
        "gauss": O.gauss_noise(g, x0, -1, -1), "gauss_fixed": O.gauss_noise(g, x0, 100.0, 25.0),
        "salt_pepper": O.salt_pepper(g, x0, x1, 50), "salt_pepper5": O.salt_pepper(g, x0, x1, 5),
        "linear_int": O.linear(g, k1, 3, 3), "linear_g16": O.linear(g, k2, 3, 3), "average": O.average(g, k1, 3, 3),
        "median": O.median(g, ones, 3, 3), "sobel": O.sobel(g), "compass": O.compass(g),
        "threshold": O.threshold(g, 40), "otsu": otsu, "dilate": O.dilate(otsu, cross, 3, 3),
        "erode": O.erode(otsu, cross, 3, 3), "kmeans": O.kmeans2(g, c0, c1)[0],
    }
    for name, w in want.items():
        assert np.array_equal(got(name), w), name
    scal = {}
    for line in open(out / "scalars.txt"):
        k, *v = line.split()
        scal[k] = v
    assert np.array_equal(np.array(scal["hist"], np.int64), O.hist(g))
    m, s = O.mean_std(g)
    assert np.float32(scal["mean_std"][0]) == m and np.float32(scal["mean_std"][1]) == s
    f = O.features(g)
    gotf = np.array(scal["features"], np.float64).astype(np.float32)
    assert np.array_equal(gotf, f) or (np.isnan(gotf) == np.isnan(f)).all()
    # calcMSQE: bytes of round(d^2 * hist[in]/255) summed modulo 256
    q = want["quantize"].astype(np.int64)
    d = g.astype(np.int64) - q
    h = O.hist(g).astype(np.float64)
    v = np.floor(np.abs(d * d * (h[g] / 255.0)) + 0.5).astype(np.int64)
    assert int(scal["msqe"][0]) == int(v.astype(np.uint8).astype(np.int64).sum() % 256)
