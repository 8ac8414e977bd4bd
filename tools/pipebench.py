#!/usr/bin/env python3
"""Time an operation list through the executor in its three modes on the bench batch.
usage: python tools/pipebench.py [--size WxH] light|pa|input|morph ..."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from imageanalysis_amd.pipeline import Pipeline
G16 = bench.GAUSS16
ONES, CROSS = [1] * 9, [0, 1, 0, 1, 1, 1, 0, 1, 0]
CASES = {
    "light": [{"type": 4, "kW": 3, "kH": 3, "fkernel": G16}, {"type": 10, "kW": 3, "kH": 3, "ikernel": ONES},
              {"type": 10, "kW": 3, "kH": 3, "ikernel": CROSS}],
    "lin2": [{"type": 4, "kW": 3, "kH": 3, "fkernel": G16}, {"type": 4, "kW": 3, "kH": 3, "fkernel": G16}],
    "morph": [{"type": 12}, {"type": 10, "kW": 3, "kH": 3, "ikernel": ONES}, {"type": 10, "kW": 3, "kH": 3, "ikernel": CROSS}],
    "medsob": [{"type": 6, "kW": 3, "kH": 3, "ikernel": ONES}, {"type": 7}],
    "pa": bench.P_A,
    "input": [  # test/configs/input.json
        {"type": 2, "mean": -1.0, "stddev": -1.0}, {"type": 1}, {"type": 13, "ikernel": [10, 20, 15, 30, 250, 77], "iparam": 2},
        {"type": 4, "kW": 3, "kH": 3, "fkernel": [1, 1, 1, 10, 1, 1, 1, 1, 1]},
        {"type": 5, "kW": 3, "kH": 3, "fkernel": [1, 1, 1, 10, 1, 1, 1, 1, 1]}, {"type": 3, "iparam": 50},
        {"type": 6, "kW": 3, "kH": 3, "ikernel": ONES}, {"type": 7}, {"type": 8}, {"type": 11, "iparam": 40},
        {"type": 9, "kW": 3, "kH": 3, "ikernel": CROSS}, {"type": 10, "kW": 3, "kH": 3, "ikernel": CROSS}, {"type": 12},
        {"type": 14}, {"type": 15}],
}
B = 500
args = sys.argv[1:]
if args and args[0] == "--size":
    bench.W, bench.H = (int(v) for v in args[1].lower().split("x"))
    args = args[2:]
rgb = bench.make_cell_batch(B, torch.device("cuda"), 1)
out = torch.empty((B, bench.H, bench.W), dtype=torch.uint8, device="cuda")
for name in args:
    for mode in [False, "pointwise", True]:
        p = Pipeline(bench.W, bench.H, B, CASES[name], fused=mode)
        p.bind_features(torch.zeros((B, 7), dtype=torch.float32, device="cuda"))
        for _ in range(3):
            p.run(rgb, out=out)
        p.set_timing(10)
        for _ in range(10):
            p.run(rgb, out=out)
        acc = {}
        for sl in range(10):
            for n, ms, b, mm in p.timings(sl):
                acc[n] = acc.get(n, 0) + ms / 10
        acc = dict(sorted(acc.items(), key=lambda kv: -kv[1]))
        print(name, mode, " total %.1f us  " % (1e3 * sum(acc.values())), {k: round(v * 1e3, 1) for k, v in acc.items()})
        p.close()
