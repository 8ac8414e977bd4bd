"""Batched feature extraction, k-means thresholding and the libc rand() stream against the
oracle (and against the C library itself for rand)."""
import numpy as np
import pytest
import torch

import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from imageanalysis_amd import ops as o
    return o


def test_libc_rand_stream_matches_the_c_library(ops):
    want = O.libc_random_0_256(2000)
    r = ops.LibcRand()
    a = r.centroids(300, 1).cpu().numpy().reshape(-1)           # draws 0..599
    b = r.centroids(100, 7).cpu().numpy()                       # draws 600..1999, operation-major
    assert np.array_equal(a, want[:600])
    assert tuple(a[:6]) == (215, 101, 201, 205, 234, 50)        # first centroids of an unseeded run
    assert np.array_equal(b.transpose(1, 0, 2).reshape(-1), want[600:2000])


def test_libc_rand_other_seed(ops):
    import ctypes as C
    libc = C.CDLL("libc.so.6")
    libc.rand.restype = C.c_int
    libc.srand(20261111)
    want = [libc.rand() // (2147483647 // 257 + 1) for _ in range(64)]
    got = ops.LibcRand(seed=20261111).centroids(32).cpu().numpy().reshape(-1)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("W,H,kind", [(64, 48, "C"), (37, 23, "U"), (768, 24, "C"), (48, 64, "K")])
def test_kmeans_matches_oracle(ops, W, H, kind):
    B = 6
    imgs = synth.gray_batch(kind, W, H, B)
    cent = np.array([[215, 101], [201, 205], [234, 50], [0, 255], [7, 7], [256, 3]], np.int32)  # 256 -> (uchar) 0
    got, it = ops.kMeansThresholding(torch.from_numpy(imgs).cuda(), torch.from_numpy(cent).cuda(), return_iterations=True)
    got, it = got.cpu().numpy(), it.cpu().numpy()
    for b in range(B):
        want, n = O.kmeans2(imgs[b], int(cent[b, 0]), int(cent[b, 1]))
        assert np.array_equal(got[b], want), (b, cent[b])
        assert it[b] == n


def test_kmeans_degenerate_images(ops):
    W, H = 32, 16
    imgs = np.stack([np.zeros((H, W), np.uint8), np.full((H, W), 255, np.uint8), np.full((H, W), 9, np.uint8)])
    cent = np.array([[10, 20], [20, 10], [9, 9]], np.int32)
    got = ops.kMeansThresholding(torch.from_numpy(imgs).cuda(), torch.from_numpy(cent).cuda()).cpu().numpy()
    for b in range(3):
        want, _ = O.kmeans2(imgs[b], int(cent[b, 0]), int(cent[b, 1]))
        assert np.array_equal(got[b], want)


@pytest.mark.parametrize("W,H,kind", [(64, 48, "C"), (37, 23, "U"), (768, 40, "C"), (48, 203, "U"), (1056, 21, "C"),
                                      (32, 16, "K")])
@pytest.mark.parametrize("with_hist", [False, True])
def test_features_match_oracle(ops, W, H, kind, with_hist):
    B = 5
    imgs = synth.gray_batch(kind, W, H, B)
    if kind == "C":
        imgs[1] = 0          # all-zero image: ranges stay -1
        imgs[2, :, : W // 2] = 0
    x = torch.from_numpy(imgs).cuda()
    hist = ops.calculateHistogram(x) if with_hist else None
    got = ops.featureExtraction(x, hist=hist).cpu().numpy()
    for b in range(B):
        want = O.features(imgs[b])
        assert np.array_equal(got[b], want), (b, got[b], want)


@pytest.mark.parametrize("W,H", [(64, 48), (768, 40), (37, 23)])
def test_features_of_images_whose_mask_is_empty_or_full(ops, monkeypatch, W, H):
    """Otsu's threshold on a two-valued image with both values present is the upper value: no pixel lies above it,
    the mask is empty and the edge count has nothing to read (the kernel skips such images).  Mixed in one batch
    with images that do have a mask, an all-255 image (mask = everything) and an all-zero one; the same with the
    shortcut switched off."""
    rs = np.random.default_rng(W + H)
    imgs = synth.gray_batch("C", W, H, 7)
    imgs[1] = (rs.random((H, W)) < 0.4) * np.uint8(255)          # 0 / 255
    imgs[2] = np.where(rs.random((H, W)) < 0.7, 17, 200).astype(np.uint8)  # two grey levels
    imgs[3] = 255
    imgs[4] = 0
    imgs[5] = 90                                                  # one grey level
    x = torch.from_numpy(imgs).cuda()
    want = np.stack([O.features(imgs[b]) for b in range(7)])
    assert want[1][3] == 0 and want[1][2] == 0  # the empty mask: no perimeter, no area
    got = ops.featureExtraction(x).cpu().numpy()
    assert np.array_equal(got, want), (got, want)
    monkeypatch.setenv("IAK_NO_EMPTY_MASK_SKIP", "1")
    got = ops.featureExtraction(x).cpu().numpy()
    assert np.array_equal(got, want), (got, want)


def test_features_single_image_and_strided_batch(ops):
    W, H = 64, 32
    imgs = synth.gray_batch("C", W, H, 4)
    big = torch.zeros((4, H + 3, W), dtype=torch.uint8, device="cuda")
    big[:, :H] = torch.from_numpy(imgs).cuda()
    got = ops.featureExtraction(big[:, :H]).cpu().numpy()      # image stride (H+3)*W
    one = ops.featureExtraction(torch.from_numpy(imgs[2]).cuda()).cpu().numpy()
    for b in range(4):
        assert np.array_equal(got[b], O.features(imgs[b]))
    assert np.array_equal(one, O.features(imgs[2]))


def _dataset(n, A, classes, seed, ties=False):
    rs = np.random.default_rng(seed)
    X = rs.normal(0, 1, (n, A - 1)).astype(np.float32) * 10
    y = rs.integers(0, classes, n)
    X += y[:, None].astype(np.float32) * 6          # separable-ish clusters
    if ties:
        X = np.round(X / 8) * 8                     # many exactly equal distances
    return np.concatenate([X, y[:, None].astype(np.float32)], axis=1).astype(np.float32)


@pytest.mark.parametrize("n_train,n_test,A,k,ties", [(300, 40, 8, 3, False), (50, 7, 8, 1, False), (130, 33, 4, 5, True),
                                                     (5, 3, 8, 9, False), (700, 20, 2, 64, True)])
def test_knn_matches_oracle(ops, n_train, n_test, A, k, ties):
    train = _dataset(n_train, A, 5, 11 + n_train, ties)
    test = _dataset(n_test, A, 5, 12 + n_test, ties)
    if ties:
        test[: min(5, n_test)] = train[:5][: min(5, n_test)]      # zero distances, duplicated rows
        train[7 % n_train] = train[3 % n_train]
    got = ops.knn(torch.from_numpy(train).cuda(), torch.from_numpy(test).cuda(), k).cpu().numpy()
    assert np.array_equal(got, O.knn_predict(train, test, k))


def test_knn_crossval_through_the_driver(tmp_path):
    """The driver's knn stage: reads the CSV, shuffles it with rand() as the reference does,
    runs the ten folds and prints the reference's report line."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data = _dataset(203, 8, 5, 99)
    csv = tmp_path / "features.csv"
    csv.write_text("".join(",".join(repr(float(v)) for v in row[:-1]) + f",{int(row[-1])}\n" for row in data))
    (tmp_path / "in").mkdir()
    cfg = {"image_folder": str(tmp_path / "in"), "input_image_filter": "", "saveIntermediateImages": False,
           "saveFinalImages": False, "calc_MSQE": False, "output_dir": str(tmp_path), "extract_channel": "grey",
           "operations": [], "knn": True, "knnConfig": {"k": 3, "input": str(csv)}}
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    r = subprocess.run([os.path.join(root, "imageanalysis_amd", "lib", "ia_driver"), str(tmp_path / "cfg.json")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    # the same shuffle here: srand(1) state, swap(v[i], v[rand() % (i + 1)])
    import ctypes as C
    libc = C.CDLL("libc.so.6")
    libc.rand.restype = C.c_int
    libc.srand(1)
    rows = [np.array([np.float32(float(c)) for c in l.split(",")], np.float32) for l in csv.read_text().splitlines()]
    for i in range(1, len(rows)):
        j = libc.rand() % (i + 1)
        rows[i], rows[j] = rows[j], rows[i]
    acc, prec, rec, folds = O.knn_crossval(np.stack(rows), 3)
    lines = r.stdout.splitlines()
    fold_lines = [l for l in lines if l.startswith("thread ")]
    assert len(fold_lines) == 10
    for f, l in enumerate(fold_lines):
        assert l.startswith(f"thread {f} had accuracy of {folds[f]:f},"), (l, folds[f])
    assert f"Average accuracy was {np.float64(np.sum(folds.astype(np.float64)) / 10.0):f}" in lines
    last = [l for l in lines if l.startswith("Total time required for a tenfold cross validation knn with k of 3 on 203 instances with 7 features")]
    assert len(last) == 1 and f"achieved accuracy of {np.float32(acc * 100.0):0.4f}%, precision of: {prec:f}, and recall of {rec:f}" in last[0]


def test_features_alignment(ops):
    # one-pass form (W % 16 == 0): 4-byte aligned images are taken, others refused
    from imageanalysis_amd.lib import IakError
    W, H, B = 64, 32, 2
    imgs = synth.gray_batch("C", W, H, B)
    buf = torch.zeros(B * H * W + 16, dtype=torch.uint8, device="cuda")
    x = buf[8:8 + B * H * W].view(B, H, W)          # 8 bytes off a 16-byte boundary
    assert x.data_ptr() % 16 == 8
    x.copy_(torch.from_numpy(imgs).cuda())
    got = ops.featureExtraction(x).cpu().numpy()
    for b in range(B):
        assert np.array_equal(got[b], O.features(imgs[b]))
    y = buf[2:2 + B * H * W].view(B, H, W)          # 2 bytes off
    with pytest.raises(IakError):
        ops.featureExtraction(y)
