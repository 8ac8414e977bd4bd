#!/usr/bin/env python3
"""PCIe-inclusive rate of the pipeline: host-resident (pinned) RGB batches in, result batches
out, double-buffered so that the upload of batch n+1, the pipeline on batch n and the download
of batch n-1 overlap.  Prints Mpixels/s and the bus rates."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from imageanalysis_amd.pipeline import Pipeline

B, W, H = 500, bench.W, bench.H
dev = torch.device("cuda")
src = bench.make_cell_batch(B, dev, 1)
h_in = [src.cpu().pin_memory() for _ in range(2)]
h_out = [torch.empty((B, H, W), dtype=torch.uint8).pin_memory() for _ in range(2)]
d_in = [torch.empty_like(src) for _ in range(2)]
d_out = [torch.empty((B, H, W), dtype=torch.uint8, device=dev) for _ in range(2)]
pipe = Pipeline(W, H, B, bench.P_A)
s_up, s_run, s_down = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
up_done = [torch.cuda.Event() for _ in range(2)]
run_done = [torch.cuda.Event() for _ in range(2)]
down_done = [torch.cuda.Event() for _ in range(2)]


def loop(n):
    for i in range(n):
        k = i & 1
        with torch.cuda.stream(s_up):
            s_up.wait_event(run_done[k])          # the run that last read d_in[k]
            d_in[k].copy_(h_in[k], non_blocking=True)
            up_done[k].record()
        with torch.cuda.stream(s_run):
            s_run.wait_event(up_done[k])
            s_run.wait_event(down_done[k])        # the download that last read d_out[k]
            pipe.run(d_in[k], out=d_out[k])
            run_done[k].record()
        with torch.cuda.stream(s_down):
            s_down.wait_event(run_done[k])
            h_out[k].copy_(d_out[k], non_blocking=True)
            down_done[k].record()
    torch.cuda.synchronize()


for e in run_done + down_done:
    e.record()
loop(4)
n = 20
t0 = time.perf_counter()
loop(n)
dt = time.perf_counter() - t0
px = n * B * W * H
print(f"{px / dt / 1e6:.1f} Mpixels/s PCIe-inclusive; {dt / n * 1e3:.2f} ms per batch of {B}; "
      f"upload {3 * px / dt / 1e9:.1f} GB/s, download {px / dt / 1e9:.1f} GB/s")
