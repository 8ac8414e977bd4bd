import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
import oracle_lib as O, synth
from imageanalysis_amd import ops
for (W, H, kind, kW, kH) in [(200, 70, "C", 5, 5), (200, 70, "U", 5, 5), (131, 150, "U", 7, 7)]:
    imgs = synth.gray_batch(kind, W, H, 1)
    w = [1] * (kW * kH)
    got = ops.medianFilter(torch.from_numpy(imgs).cuda(), w, kW, kH).cpu().numpy()[0]
    want = O.median(imgs[0], w, kW, kH)
    bad = np.argwhere(got != want)
    print(W, H, kind, kW, kH, len(bad), "rows", sorted(set(bad[:, 0]))[:20], "cols", sorted(set(bad[:, 1]))[:20])
    for r, c in bad[:8]:
        print("  ", r, c, got[r, c], want[r, c])
