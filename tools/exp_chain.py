#!/usr/bin/env python3
"""Time variations of P-A's per-pixel run in front of the 3x3 linear filter (fused plan)."""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from imageanalysis_amd.pipeline import Pipeline
LIN = {"type": 4, "kW": 3, "kH": 3, "fkernel": bench.GAUSS16}
EQ, GA, GAF, SP = {"type": 1}, {"type": 2, "mean": -1.0, "stddev": -1.0}, {"type": 2, "mean": 100.0, "stddev": 30.0}, {"type": 3, "iparam": 50}
CASES = {"lin": [LIN], "eq+lin": [EQ, LIN], "gauss+lin": [GA, LIN], "gaussfixed+lin": [GAF, LIN], "sp+lin": [SP, LIN],
         "eq+gauss+lin": [EQ, GA, LIN], "eq+sp+lin": [EQ, SP, LIN], "gauss+sp+lin": [GA, SP, LIN], "eq+gauss+sp+lin": [EQ, GA, SP, LIN],
         "eq+gauss+sp": [EQ, GA, SP], "eq": [EQ], "gauss": [GA], "sp": [SP]}
B = 500
rgb = bench.make_cell_batch(B, torch.device("cuda"), 1)
out = torch.empty((B, bench.H, bench.W), dtype=torch.uint8, device="cuda")
names = sys.argv[1:] or list(CASES)
for rep in range(2):
    for name in names:
        p = Pipeline(bench.W, bench.H, B, CASES[name], fused=True)
        for _ in range(5):
            p.run(rgb, out=out)
        p.set_timing(20)
        for _ in range(20):
            p.run(rgb, out=out)
        acc = {}
        for sl in range(20):
            for n, ms, b, mm in p.timings(sl):
                acc[n] = acc.get(n, [0, 0])
                acc[n][0] += ms / 20
                acc[n][1] += mm / 20
        if rep:
            print(name.ljust(18), {k: (round(v[0] * 1e3, 1), round(v[1] * 1e3, 1)) for k, v in acc.items() if k != "gray+hist"}, flush=True)
        p.close()
