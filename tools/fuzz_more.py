#!/usr/bin/env python3
"""Extended differential fuzzing of the fused executor against the oracle, biased towards the
threshold / morphology / 3x3-stencil runs that the bit-mask kernels take over.
usage: python tools/fuzz_more.py [first_case] [n_cases] [big]   (big: images up to 1100 x 600)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as O, synth
from test_gpu_fuzz import random_op
from imageanalysis_amd.pipeline import Pipeline

FOCUS = {O.OP_THRESHOLD, O.OP_OTSU, O.OP_DILATE, O.OP_ERODE, O.OP_SOBEL, O.OP_COMPASS, O.OP_MEDIAN, O.OP_LINEAR,
         O.OP_AVERAGE, O.OP_FEATURES, O.OP_EQUALIZE}
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
big = len(sys.argv) > 3
bad_cases = 0
for case in range(first, first + count):
    rs = np.random.default_rng(case)
    W = int(rs.choice([8, 16, 24, 32, 40, 48, 64, 80, 96, 128, 136, 256])) if rs.random() < 0.75 else int(rs.integers(1, 90))
    H = int(rs.integers(1, 40))
    if big:
        W = int(rs.choice([256, 512, 768, 776, 1024, 1040, 1088])) if rs.random() < 0.75 else int(rs.integers(90, 1100))
        H = int(rs.integers(40, 600))
    B = int(rs.integers(1, 4)) if not big else int(rs.integers(1, 3))
    kind = str(rs.choice(["C", "U", "K"]))
    ops = []
    for _ in range(int(rs.integers(1, 8))):
        op = random_op(rs)
        while op["type"] not in FOCUS:
            op = random_op(rs)
        if op["type"] in (O.OP_DILATE, O.OP_ERODE) and rs.random() < 0.8:
            op.update(kW=3, kH=3, ikernel=[int(v) for v in rs.integers(0, 2, 9)])
        ops.append(op)
    rgb = synth.rgb_batch(kind, W, H, B, first_index=case)
    if rs.random() < 0.3:
        rgb[0, :, : H // 2] = 0
    x0, x1 = O.xorwow_planes(W * H)
    want, want_hist, want_feats = O.pipeline_ex(rgb, ops, x0, x1, centroids=None)
    p = Pipeline(W, H, B, ops, fused=True)
    feats = torch.zeros((B, 7), dtype=torch.float32, device="cuda")
    p.bind_features(feats)
    got = p.run(torch.from_numpy(rgb).cuda()).cpu().numpy()
    names = [n for n, _ in p.groups()]
    ok = np.array_equal(got, want)
    if ok and any(o["type"] == O.OP_FEATURES for o in ops):
        ok = np.array_equal(feats.cpu().numpy(), want_feats)
    if not ok:
        bad_cases += 1
        bad = np.argwhere(got != want)
        print(f"MISMATCH case {case} {W}x{H}x{B} {kind} groups {names}: {len(bad)} pixels, first {bad[:1].tolist()}, ops {ops}", flush=True)
        if bad_cases >= 5:
            break
    p.close()
    if (case - first) % (10 if big else 200) == 0:
        print("case", case, "ok so far", flush=True)
print("done:", bad_cases, "bad cases of", count, flush=True)
sys.exit(1 if bad_cases else 0)
