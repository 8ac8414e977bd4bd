#!/usr/bin/env python3
"""Per-entry-point timing on the bench batch (B x 768x568, cell-like), interleaved rounds in
one process.  Usage: python tools/kbench.py [--batch 500] [--reps 20] [--only name,name]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from imageanalysis_amd import ops  # noqa: E402
from imageanalysis_amd import lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=500)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--kind", default="C")
    ap.add_argument("--size", default="", help="WxH instead of the bench's 768x568")
    args = ap.parse_args()
    if args.size:
        bench.W, bench.H = (int(v) for v in args.size.split("x"))
    B, H, W = args.batch, bench.H, bench.W
    dev = torch.device("cuda")
    rgb = bench.make_cell_batch(B, dev, 1)
    if args.kind == "U":
        rgb = torch.randint(0, 256, rgb.shape, dtype=torch.uint8, device=dev)
    elif args.kind == "K":
        rgb.fill_(128)
    lib = L.load()
    st = lambda: torch.cuda.current_stream().cuda_stream
    N = H * W
    gray = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    out = torch.empty_like(gray)
    hist = torch.empty((B, 256), dtype=torch.int32, device=dev)
    lib.iak_gray(rgb.data_ptr(), gray.data_ptr(), W, H, B, 3 * N, N, 0, st())
    rng = ops.Randomness(N)
    rv, sp = rng.rv(), rng.sp(50)
    m = torch.empty((B, 256), dtype=torch.int32, device=dev)
    lib.iak_hist(gray.data_ptr(), hist.data_ptr(), N, B, N, st())
    lib.iak_eq_map(hist.data_ptr(), m.data_ptr(), N, B, st())
    table = torch.empty((B, 256), dtype=torch.int16, device=dev)
    lib.iak_gauss_table(hist.data_ptr(), N, B, -1.0, -1.0, table.data_ptr(), st())
    g16 = np.array(bench.GAUSS16, np.float32)
    kint = np.array([1, 1, 1, 10, 1, 1, 1, 1, 1], np.float32)
    ones = np.ones(9, np.int32)
    cross = np.array([0, 1, 0, 1, 1, 1, 0, 1, 0], np.int32)
    box11 = np.full(121, 1.0 / 121, np.float32)
    ones25f = np.ones(25, np.float32)
    ones49 = np.ones(49, np.int32)
    b5 = np.array([1, 4, 6, 4, 1], np.float32)
    gauss5 = (np.outer(b5, b5) / 256.0).astype(np.float32).ravel()
    rand7 = (np.random.default_rng(3).random(49) / 25.0).astype(np.float32)
    rand9 = (np.random.default_rng(4).random(81) / 40.0).astype(np.float32)
    ones81 = np.ones(81, np.int32)
    disk9 = np.array([1 if (i // 9 - 4) ** 2 + (i % 9 - 4) ** 2 <= 18 else 0 for i in range(81)], np.int32)
    ones81f = np.ones(81, np.float32)
    box7 = np.full(49, 0.02, np.float32)
    wmed = np.array([1, 2, 1, 2, 3, 2, 1, 2, 1], np.int32)
    g, o = gray.data_ptr(), out.data_ptr()
    feats = torch.empty((B, 7), dtype=torch.float32, device=dev)
    fws = torch.empty(int(lib.iak_features_workspace_bytes(W, H, B)) + 256, dtype=torch.uint8, device=dev)
    fws_ptr = (fws.data_ptr() + 255) & ~255
    cases = {
        "copy(torch)": (lambda: out.copy_(gray), 2),
        "gray": (lambda: lib.iak_gray(rgb.data_ptr(), o, W, H, B, 3 * N, N, 0, st()), 4),
        "gray_avg": (lambda: lib.iak_gray(rgb.data_ptr(), o, W, H, B, 3 * N, N, 1, st()), 4),
        "gray_hist": (lambda: lib.iak_gray_hist(rgb.data_ptr(), o, hist.data_ptr(), W, H, B, 3 * N, N, 0, st()), 5),
        "hist": (lambda: lib.iak_hist(g, hist.data_ptr(), N, B, N, st()), 1),
        "features": (lambda: lib.iak_features(g, W, H, B, N, hist.data_ptr(), feats.data_ptr(), fws_ptr, st()), 1),
        "apply_map": (lambda: lib.iak_apply_map(g, o, N, B, N, N, m.data_ptr(), 256, st()), 2),
        "threshold": (lambda: lib.iak_threshold(g, o, N, B, N, N, 100, st()), 2),
        "salt_pepper": (lambda: lib.iak_salt_pepper(g, o, N, B, N, N, sp.data_ptr(), st()), 3),
        "gauss_noise": (lambda: lib.iak_gauss_noise(g, o, N, B, N, N, rv.data_ptr(), table.data_ptr(), 256, st()), 3),
        "linear3_float": (lambda: lib.iak_linear(g, o, W, H, B, N, N, g16.ctypes.data, 3, 3, 0, st()), 2),
        "linear3_int": (lambda: lib.iak_linear(g, o, W, H, B, N, N, kint.ctypes.data, 3, 3, 0, st()), 2),
        "median3": (lambda: lib.iak_median(g, o, W, H, B, N, N, ones.ctypes.data, 3, 3, st()), 2),
        "sobel": (lambda: lib.iak_sobel(g, o, W, H, B, N, N, st()), 2),
        "compass": (lambda: lib.iak_compass(g, o, W, H, B, N, N, st()), 2),
        "erode3": (lambda: lib.iak_erode(g, o, W, H, B, N, N, cross.ctypes.data, 3, 3, st()), 2),
        "dilate3": (lambda: lib.iak_dilate(g, o, W, H, B, N, N, cross.ctypes.data, 3, 3, st()), 2),
        "median3_weighted": (lambda: lib.iak_median(g, o, W, H, B, N, N, wmed.ctypes.data, 3, 3, st()), 2),
        "erode5": (lambda: lib.iak_erode(g, o, W, H, B, N, N, ones49.ctypes.data, 5, 5, st()), 2),
        "linear5": (lambda: lib.iak_linear(g, o, W, H, B, N, N, ones25f.ctypes.data, 5, 5, 0, st()), 2),
        "linear11_box": (lambda: lib.iak_linear(g, o, W, H, B, N, N, box11.ctypes.data, 11, 11, 0, st()), 2),
        "average5": (lambda: lib.iak_linear(g, o, W, H, B, N, N, ones25f.ctypes.data, 5, 5, 1, st()), 2),
        "median7": (lambda: lib.iak_median(g, o, W, H, B, N, N, ones49.ctypes.data, 7, 7, st()), 2),
        "median5": (lambda: lib.iak_median(g, o, W, H, B, N, N, ones49.ctypes.data, 5, 5, st()), 2),
        "erode5": (lambda: lib.iak_erode(g, o, W, H, B, N, N, ones49.ctypes.data, 5, 5, st()), 2),
        "dilate5": (lambda: lib.iak_dilate(g, o, W, H, B, N, N, ones49.ctypes.data, 5, 5, st()), 2),
        "linear5_gauss": (lambda: lib.iak_linear(g, o, W, H, B, N, N, gauss5.ctypes.data, 5, 5, 0, st()), 2),
        "linear7_rand": (lambda: lib.iak_linear(g, o, W, H, B, N, N, rand7.ctypes.data, 7, 7, 0, st()), 2),
        "linear9_rand": (lambda: lib.iak_linear(g, o, W, H, B, N, N, rand9.ctypes.data, 9, 9, 0, st()), 2),
        "linear9x1": (lambda: lib.iak_linear(g, o, W, H, B, N, N, rand9.ctypes.data, 9, 1, 0, st()), 2),
        "linear1x7": (lambda: lib.iak_linear(g, o, W, H, B, N, N, rand9.ctypes.data, 1, 7, 0, st()), 2),
        "erode9": (lambda: lib.iak_erode(g, o, W, H, B, N, N, ones81.ctypes.data, 9, 9, st()), 2),
        "dilate9_disk": (lambda: lib.iak_dilate(g, o, W, H, B, N, N, disk9.ctypes.data, 9, 9, st()), 2),
        "average9": (lambda: lib.iak_linear(g, o, W, H, B, N, N, ones81f.ctypes.data, 9, 9, 1, st()), 2),
        "box7_frac": (lambda: lib.iak_linear(g, o, W, H, B, N, N, box7.ctypes.data, 7, 7, 0, st()), 2),
    }
    only = [x for x in args.only.split(",") if x]
    names = [n for n in cases if not only or n in only]
    times = {n: [] for n in names}
    for n in names:  # warm
        cases[n][0]()
    torch.cuda.synchronize()
    for _ in range(args.reps):
        for n in names:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            cases[n][0]()
            e1.record()
            e1.synchronize()
            times[n].append(e0.elapsed_time(e1))
    px = B * N
    print(f"batch {B} x {W}x{H} ({px/1e6:.1f} Mpx), kind {args.kind}, median/min over {args.reps} rounds")
    for n in names:
        t = np.array(times[n])
        med, mn = np.median(t), t.min()
        ab = cases[n][1]
        print(f"{n:16s} med {med*1e3:8.1f} us  min {mn*1e3:8.1f} us   {px/med/1e6:8.1f} Gpx/s   alg {ab} B/px -> {ab*px/med/1e6:7.0f} GB/s")


if __name__ == "__main__":
    main()
