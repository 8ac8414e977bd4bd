#!/usr/bin/env python3
"""Differential fuzzing of the windowed entry points (linear / average / median / erosion /
dilation) with windows up to 25 x 25 on images up to 400 x 300, against the oracle.
usage: python tools/fuzz_windows.py [first_case] [n_cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as O, synth
from imageanalysis_amd import ops

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = 0
for case in range(first, first + count):
    rs = np.random.default_rng(770000 + case)
    W = int(rs.integers(1, 400)) if rs.random() < 0.6 else int(rs.choice([64, 128, 200, 256, 384]))
    H = int(rs.integers(1, 300))
    B = int(rs.integers(1, 3))
    kind = str(rs.choice(["C", "U", "K"]))
    imgs = synth.gray_batch(kind, W, H, B, first_index=case)
    if rs.random() < 0.4:
        imgs[rs.random(imgs.shape) < 0.5] = 0
    kW, kH = int(rs.integers(1, 26)), int(rs.integers(1, 26))
    if rs.random() < 0.5:
        kW |= 1; kH |= 1
    which = int(rs.integers(0, 7))
    x = torch.from_numpy(imgs).cuda()
    what = None
    if which == 0:
        if not (kW & 1 and kH & 1) or kW * kH > 255: kW, kH = 7, 5
        w = [1] * (kW * kH); what = f"median ones {kW}x{kH}"
        got, ref = ops.medianFilter(x, w, kW, kH), lambda g: O.median(g, w, kW, kH)
    elif which == 1:
        kW, kH = int(rs.choice([1, 3, 5])), int(rs.choice([1, 3, 5]))
        w = [int(v) for v in rs.integers(0, 5, kW * kH)]; w[0] += 1; what = f"median weighted {kW}x{kH}"
        got, ref = ops.medianFilter(x, w, kW, kH), lambda g: O.median(g, w, kW, kH)
    elif which in (2, 3):
        mode = rs.integers(0, 4)
        if mode == 0: w = np.full(kW * kH, float(rs.choice([1.0 / (kW * kH), 1.0, 0.5, 0.04, 1.0 / 121, 2.0])), np.float32)
        elif mode == 1: w = ((rs.random(kW * kH) - 0.3) * 2.0 / max(1, kW)).astype(np.float32)
        elif mode == 2: w = rs.integers(-2, 4, kW * kH).astype(np.float32)
        else: w = (rs.integers(0, 9, kW * kH) / 64.0).astype(np.float32)
        if which == 2:
            what = f"linear m{mode} {kW}x{kH}"; got, ref = ops.linearFilter(x, w, kW, kH), lambda g: O.linear(g, w, kW, kH)
        else:
            what = f"average m{mode} {kW}x{kH}"; got, ref = ops.averageFilter(x, w, kW, kH), lambda g: O.average(g, w, kW, kH)
    else:
        dens = float(rs.choice([0.1, 0.5, 1.0]))
        se = [int(v) for v in (rs.random(kW * kH) < dens)]
        if which == 4 or which == 6:
            what = f"erode {kW}x{kH}"; got, ref = ops.imageErosion(x, se, kW, kH), lambda g: O.erode(g, se, kW, kH)
        else:
            what = f"dilate {kW}x{kH}"; got, ref = ops.imageDilation(x, se, kW, kH), lambda g: O.dilate(g, se, kW, kH)
    got = got.cpu().numpy()
    for b in range(B):
        want = ref(imgs[b])
        if not np.array_equal(got[b], want):
            bad += 1
            d = np.argwhere(got[b] != want)
            print(f"MISMATCH case {case} {what} {W}x{H} {kind} image {b}: {len(d)} pixels, first {d[:2].tolist()}", flush=True)
            break
    if bad >= 5:
        break
    if (case - first) % 50 == 0:
        print("case", case, what, flush=True)
print("done:", bad, "bad of", count, flush=True)
sys.exit(1 if bad else 0)
