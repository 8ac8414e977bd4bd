#!/usr/bin/env python3
"""Differential fuzzing of box filters whose taps depend on the running sum (weights such as 1/121) on random
image sizes, against the oracle's tap-by-tap loop: the bit-sliced kernel for square windows on widths that are
multiples of 4, the integer-sum kernel otherwise.
usage: python tools/fuzz_box_special.py [first_case] [n_cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as O
from imageanalysis_amd import ops

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
for case in range(first, first + count):
    rs = np.random.default_rng(990000 + case)
    K = int(rs.choice([5, 7, 9, 11, 13, 15]))
    kW, kH = (K, K) if rs.random() < 0.8 else (K, int(rs.choice([3, 5, 7, 9, 11])))
    q = kW * kH * int(rs.choice([1, 1, 1, 2]))
    w = np.float32(int(rs.choice([1, 1, 2, 3])) / q)
    W = 4 * int(rs.integers(1, 80)) if rs.random() < 0.8 else int(rs.integers(1, 300))
    if rs.random() < 0.1:
        W = int(rs.choice([1984, 2016, 2048, 2080, 4096]))
    H = int(rs.integers(1, 120)) if W < 1000 else int(rs.integers(1, 40))
    B = int(rs.integers(1, 4))
    step = max(1, int(round(1.0 / float(w))))
    critical = sorted({v for m in (1, 2, 3) for v in (m * step - 1, m * step, m * step + 1) if 0 <= v <= 255} | {0, 255})
    mode = int(rs.integers(0, 4))
    if mode == 0:
        imgs = rs.choice(critical, size=(B, H, W)).astype(np.uint8)
    elif mode == 1:
        imgs = rs.integers(0, max(2, min(step, 255)), size=(B, H, W)).astype(np.uint8)
        m = rs.random((B, H, W)) < rs.choice([0.003, 0.03, 0.2])
        imgs[m] = rs.choice(critical, size=int(m.sum()))
    elif mode == 2:
        imgs = rs.integers(0, 256, size=(B, H, W)).astype(np.uint8)
    else:
        imgs = np.zeros((B, H, W), np.uint8)
        m = rs.random((B, H, W)) < 0.01
        imgs[m] = rs.choice(critical, size=int(m.sum()))
    k = np.full(kW * kH, w, np.float32)
    got = ops.linearFilter(torch.from_numpy(imgs).cuda(), k, kW, kH).cpu().numpy()
    want = np.stack([O.linear(imgs[b], k, kW, kH) for b in range(B)])
    if not np.array_equal(got, want):
        bad += 1
        d = np.argwhere(got != want)
        print(f"FAIL case {case}: {kW}x{kH} w={w} {W}x{H}x{B} mode {mode}: {len(d)} differ, first {d[0]}", flush=True)
        if bad >= 5:
            break
    if (case - first) % 50 == 0:
        print("case", case, f"{kW}x{kH} w={w} {W}x{H}", flush=True)
print(f"done: {bad} bad of {count}")
sys.exit(1 if bad else 0)
