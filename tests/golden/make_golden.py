#!/usr/bin/env python3
"""Writes tests/golden/*.npz: small input/output vectors produced by the CPU oracle
(oracle/ia_oracle.c).  The reference holds no fixtures of its own and cannot be built in this
image, so these vectors pin the oracle against regressions and give the GPU tests a second,
frozen target; they are data only (inputs and expected outputs)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
import synth  # noqa: E402

ONES, CROSS = [1] * 9, [0, 1, 0, 1, 1, 1, 0, 1, 0]
K_INPUT = [1, 1, 1, 10, 1, 1, 1, 1, 1]                       # test/configs/input.json:43-53
K_LINEAR = [-1, 0, -1, -2, 0, -2, -1, 0, -1]                 # test/configs/linear-filter.json
K_G16 = [1 / 16, 2 / 16, 1 / 16, 2 / 16, 4 / 16, 2 / 16, 1 / 16, 2 / 16, 1 / 16]
K_BOX9 = [1.0 / 9] * 9
LEVELS = [10, 20, 15, 30, 250, 77]                           # test/configs/input.json:24-36


def ops_fixture(kind, W, H):
    rgb = synth.rgb_image(kind, W, H, 11)
    g = O.gray(rgb, 0)
    x0, x1 = O.xorwow_planes(W * H)
    otsu, thr = O.otsu(g)
    m, s = O.mean_std(g)
    d = {
        "rgb": rgb, "gray": g, "gray_avg": O.gray(rgb, 1), "hist": O.hist(g),
        "eq_map": O.eq_map(O.hist(g), W * H), "equalize": O.equalize(g),
        "quantize": O.quantize(g, LEVELS), "threshold40": O.threshold(g, 40),
        "otsu": otsu, "otsu_thr": np.int32(thr), "mean_std": np.array([m, s], np.float32),
        "x0": x0, "x1": x1,
        "salt_pepper50": O.salt_pepper(g, x0, x1, 50), "gauss_auto": O.gauss_noise(g, x0, -1, -1),
        "gauss_100_25": O.gauss_noise(g, x0, 100.0, 25.0),
        "linear_input": O.linear(g, K_INPUT, 3, 3), "linear_neg": O.linear(g, K_LINEAR, 3, 3),
        "linear_g16": O.linear(g, K_G16, 3, 3), "linear_box9": O.linear(g, K_BOX9, 3, 3),
        "average_input": O.average(g, K_INPUT, 3, 3),
        "linear_box11": O.linear(g, [1.0 / 121] * 121, 11, 11),
        "median3": O.median(g, ONES, 3, 3), "median7": O.median(g, [1] * 49, 7, 7),
        "sobel": O.sobel(g), "compass": O.compass(g),
        "dilate_cross": O.dilate(otsu, CROSS, 3, 3), "erode_cross": O.erode(otsu, CROSS, 3, 3),
        "dilate_ones": O.dilate(otsu, ONES, 3, 3), "erode_ones_gray": O.erode(g, ONES, 3, 3),
        "kmeans_215_101": O.kmeans2(g, 215, 101)[0], "features": O.features(g),
    }
    return d


def main():
    for kind, W, H in (("C", 64, 48), ("U", 37, 23)):
        np.savez_compressed(os.path.join(HERE, f"ops_{kind}_{W}x{H}.npz"), **ops_fixture(kind, W, H))
    # the north-star pipeline on a batch of 2
    W, H = 64, 48
    rgb = synth.rgb_batch("C", W, H, 2, first_index=20)
    x0, x1 = O.xorwow_planes(W * H)
    ops = [{"type": O.OP_EQUALIZE}, {"type": O.OP_GAUSS, "mean": -1.0, "stddev": -1.0},
           {"type": O.OP_SALT_PEPPER, "iparam": 50}, {"type": O.OP_LINEAR, "kW": 3, "kH": 3, "fkernel": K_G16},
           {"type": O.OP_MEDIAN, "kW": 3, "kH": 3, "ikernel": ONES}, {"type": O.OP_SOBEL}]
    out, hist = O.pipeline(rgb, ops, x0, x1)
    np.savez_compressed(os.path.join(HERE, "pipeline_PA_64x48.npz"), rgb=rgb, out=out, hist_sum=hist)
    # first outputs of a few subsequences (seed 1234), for the generator alone
    subs = np.array([0, 1, 2, 63, 64, 65535, 65536, 436223, 16777215], np.uint32)
    streams = np.stack([O.xorwow_stream(1234, int(s), 4) for s in subs])
    np.savez_compressed(os.path.join(HERE, "xorwow_seed1234.npz"), subsequences=subs, outputs=streams)
    print("golden written")


if __name__ == "__main__":
    main()
