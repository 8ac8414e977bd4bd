"""The driver end to end on a folder of BMP files: batched and per-image flows against the
oracle, output naming, the feature CSV and its class labels."""
import json
import os
import subprocess

import numpy as np
import pytest

import bmp_util
import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "imageanalysis_amd", "lib", "ia_driver")
W, H = 64, 48
NAMES = ["cyl01.BMP", "cyl02.BMP", "inter01.BMP", "let01.BMP", "let02.BMP", "mod01.BMP", "para01.BMP"]
GAUSS16 = [1 / 16, 2 / 16, 1 / 16, 2 / 16, 4 / 16, 2 / 16, 1 / 16, 2 / 16, 1 / 16]

OPS_JSON = [
    {"type": "histogram-equalization"},
    {"type": "gaussian-noise", "std_dev": -1, "mean": -1},
    {"type": "salt-and-pepper", "intensity": 50},
    {"type": "linear-filter", "kernel": {"width": 3, "height": 3, "values": GAUSS16}},
    {"type": "median-filter", "kernel": {"width": 3, "height": 3, "values": [1] * 9}},
    {"type": "kMeans-thresh"},
    {"type": "erosion", "structuring-element": {"width": 3, "height": 3, "values": [0, 1, 0, 1, 1, 1, 0, 1, 0]}},
    {"type": "feature-extract", "output": None},
]
OPS = [
    {"type": O.OP_EQUALIZE}, {"type": O.OP_GAUSS, "mean": -1.0, "stddev": -1.0}, {"type": O.OP_SALT_PEPPER, "iparam": 50},
    {"type": O.OP_LINEAR, "kW": 3, "kH": 3, "fkernel": GAUSS16}, {"type": O.OP_MEDIAN, "kW": 3, "kH": 3, "ikernel": [1] * 9},
    {"type": O.OP_KMEANS}, {"type": O.OP_ERODE, "kW": 3, "kH": 3, "ikernel": [0, 1, 0, 1, 1, 1, 0, 1, 0]},
    {"type": O.OP_FEATURES},
]


@pytest.fixture(scope="module")
def folder(tmp_path_factory):
    d = tmp_path_factory.mktemp("drv")
    (d / "in").mkdir()
    rgb = synth.rgb_batch("C", W, H, len(NAMES))
    for i, n in enumerate(NAMES):
        bmp_util.write_bmp24(str(d / "in" / n), rgb[i])
    x0, x1 = O.xorwow_planes(W * H)
    cent = O.libc_random_0_256(2 * len(NAMES)).reshape(len(NAMES), 1, 2)
    want, _, feats = O.pipeline_ex(rgb, OPS, x0, x1, centroids=cent)
    return d, rgb, want, feats


def run_driver(d, mode, tag, extra=None):
    out = d / f"out_{tag}"
    out.mkdir()
    ops = json.loads(json.dumps(OPS_JSON))
    ops[-1]["output"] = str(out / "features.csv")
    cfg = {"image_folder": str(d / "in"), "input_image_filter": "", "saveIntermediateImages": False,
           "saveFinalImages": True, "calc_MSQE": False, "output_dir": str(out), "extract_channel": "grey",
           "operations": ops}
    cfg.update(extra or {})
    (d / f"cfg_{tag}.json").write_text("// driver test config\n" + json.dumps(cfg, indent=1))
    r = subprocess.run([DRIVER, str(d / f"cfg_{tag}.json")] + mode, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    return out, r.stdout


def check_outputs(out, want, feats):
    for i, n in enumerate(NAMES):
        got = bmp_util.read_bmp8(str(out / n))
        assert np.array_equal(got, want[i]), n
    rows = [l.split(",") for l in (out / "features.csv").read_text().splitlines()]
    assert len(rows) == len(NAMES)
    labels = [int(r[-1]) for r in rows]
    assert labels == [0, 0, 1, 2, 2, 3, 4]          # cyl, cyl, inter, let, let, mod, para
    got = np.array([[float(v) for v in r[:-1]] for r in rows], np.float32)
    # the CSV holds 6 significant digits (default ostream precision), as the reference's does
    assert np.allclose(got, feats, rtol=1e-5, atol=0)


@pytest.mark.parametrize("mode,tag", [(["--batch"], "batch"), (["--batch", "--batch-size", "3"], "batch3"),
                                      (["--per-image"], "per")])
def test_driver_matches_oracle(folder, mode, tag):
    d, rgb, want, feats = folder
    out, stdout = run_driver(d, mode, tag)
    check_outputs(out, want, feats)
    assert "Total time spent on the entire batch" in stdout


def test_driver_intermediates_use_the_reference_names(folder):
    d, rgb, want, feats = folder
    out, stdout = run_driver(d, [], "inter", {"saveIntermediateImages": True})
    for n in NAMES:
        for t in ["grey", "histogram-equalization", "gaussian-noise", "salt-and-pepper", "linear-filter", "median-filter",
                  "kMeans-thresh", "erosion", "feature-extract"]:
            assert (out / f"{t}-{n}").exists(), (t, n)
    gray = bmp_util.read_bmp8(str(out / f"grey-{NAMES[2]}"))
    assert np.array_equal(gray, O.gray(rgb[2], 0))
    check_outputs(out, want, feats)
    assert "Total time spent k Means thresholding images" in stdout


@pytest.mark.parametrize("mode", [["--batch"], ["--per-image"]])
def test_driver_colour_channel(folder, mode):
    d, rgb, _, _ = folder
    tag = "green" + mode[0].strip("-")
    out, _ = run_driver(d, mode, tag, {"extract_channel": "green"})
    x0, x1 = O.xorwow_planes(W * H)
    cent = O.libc_random_0_256(2 * len(NAMES)).reshape(len(NAMES), 1, 2)
    want, _, feats = O.pipeline_ex(rgb, OPS, x0, x1, centroids=cent, input=3)
    check_outputs(out, want, feats)


@pytest.mark.parametrize("gpus", [2, 3])
def test_driver_shards_the_folder_over_devices(folder, gpus):
    """--gpus N cuts the file list into N contiguous shares, one host thread each (on this box they
    share the one device).  Images, the CSV with its labels and the k-means draws (a share that
    starts at file i continues the centroid stream from image i) are the one-device results."""
    d, rgb, want, feats = folder
    out, stdout = run_driver(d, ["--batch", "--gpus", str(gpus), "--share-devices", "--dataset-summary"], f"gpus{gpus}")
    check_outputs(out, want, feats)
    line = [l for l in stdout.splitlines() if l.startswith("Dataset histogram over")]
    assert len(line) == 1 and f"over {gpus} device(s)" in line[0]
    # the data-set histogram: every non-zero grayscale pixel of every image, counted once
    nonzero = sum(int((O.gray(rgb[i], 0) != 0).sum()) for i in range(len(NAMES)))
    assert f": {nonzero} non-zero pixels in {len(NAMES)} images" in line[0]
    classes = [l for l in stdout.splitlines() if l.startswith("class ")]
    assert len(classes) == 5 and classes[0].startswith("class 0 'cyl': 2 images")


def test_driver_refuses_more_gpus_than_the_machine_has(folder):
    d, _, _, _ = folder
    out = d / "out_toomany"
    out.mkdir()
    cfg = {"image_folder": str(d / "in"), "input_image_filter": "", "saveIntermediateImages": False, "saveFinalImages": False,
           "calc_MSQE": False, "output_dir": str(out), "extract_channel": "grey", "operations": []}
    (d / "cfg_toomany.json").write_text(json.dumps(cfg))
    r = subprocess.run([DRIVER, str(d / "cfg_toomany.json"), "--gpus", "64"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "this machine has" in r.stderr


def test_knn_after_kmeans_sees_the_same_rand_stream_in_both_flows(folder):
    """kMeans-thresh draws its centroids from rand(); the cross-validation shuffles its rows with
    rand() afterwards (test/test.cu:165,587).  Both flows of the driver must leave the stream at
    the same point, so the folds and the report agree."""
    d, _, _, _ = folder
    rs = np.random.default_rng(5)
    rows = rs.normal(0, 1, (120, 8)).astype(np.float32) * 10
    y = rs.integers(0, 5, 120)
    rows[:, :7] += y[:, None] * 6
    rows[:, 7] = y
    csv = d / "knn_in.csv"
    csv.write_text("".join(",".join(repr(float(v)) for v in r[:-1]) + f",{int(r[-1])}\n" for r in rows))
    reports = {}
    for mode in ("--batch", "--per-image"):
        _, stdout = run_driver(d, [mode], "knn" + mode.strip("-"), {"knn": True, "knnConfig": {"k": 3, "input": str(csv)}})
        folds = [l for l in stdout.splitlines() if l.startswith("thread ")]
        last = [l.split(" ms achieved ")[1] for l in stdout.splitlines() if l.startswith("Total time required for a tenfold")]
        assert len(folds) == 10 and len(last) == 1
        reports[mode] = (sorted(folds), last[0])
    assert reports["--batch"] == reports["--per-image"]
