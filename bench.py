#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of the full filter pipeline (grayscale -> histogram ->
equalise -> Gaussian noise -> salt-and-pepper -> 3x3 linear -> 3x3 median -> Sobel) on a
device-resident batch of synthetic 768x568 cell-like images, through the C ABI's batched
executor.  One process per GPU; images shard image-wise (every rank owns a full batch,
weak scaling); the only collective is one all-reduce of the 256-bin dataset histogram per
step.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from imageanalysis_amd import lib as iak_lib  # noqa: E402
from imageanalysis_amd.pipeline import Pipeline  # noqa: E402

W, H = 768, 568
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak

GAUSS16 = [1 / 16, 2 / 16, 1 / 16, 2 / 16, 4 / 16, 2 / 16, 1 / 16, 2 / 16, 1 / 16]
OP_EQUALIZE, OP_GAUSS, OP_SP, OP_LINEAR, OP_MEDIAN, OP_SOBEL = 1, 2, 3, 4, 6, 7
P_A = [
    {"type": OP_EQUALIZE},
    {"type": OP_GAUSS, "mean": -1.0, "stddev": -1.0},
    {"type": OP_SP, "iparam": 50},
    {"type": OP_LINEAR, "kW": 3, "kH": 3, "fkernel": GAUSS16},
    {"type": OP_MEDIAN, "kW": 3, "kH": 3, "ikernel": [1] * 9},
    {"type": OP_SOBEL},
]
# launch group -> the HIP kernel that does its image pass (names as rocprofv3 prints them)
KERNEL_OF_GROUP = {
    "gray+hist": "iak::gray_kernel<0, true>",
    "equalize+gauss+salt_pepper+linear+median+sobel": "iak::strip_kernel",
    "equalize+gauss+salt_pepper+linear": "iak::chain_stencil8_flat_kernel<iak::Lin8Swar>",
    "equalize+gauss+salt_pepper": "iak::chain_apply_kernel<false>",
    "linear": "iak::stencil8_kernel<iak::Lin8Swar, 8, false, false>",
    "median": "iak::stencil8_kernel<iak::Median8, 8, false, false>",
    "sobel": "iak::stencil8_kernel<iak::Sobel8, 8, false, false>",
}
# kernels behind the group names of the other two workloads, where they differ from P-A's
KERNEL_OF_GROUP_BY_WORKLOAD = {
    "P-B": {"linear": "iak::tiled_linear_kernel<11, false, true>", "median": "iak::huang_median_kernel<7>"},
    "P-C": {"otsu": "iak::chain_apply_kernel<false>", "otsu+sobel": "iak::binary_table_stencil_kernel",
            "dilate+erode+hist": "iak::morph_chain_kernel<2, 2, false>", "features": "iak::perimeter_bits_kernel"},
}
# The histogram rides on the grayscale pass (the pixel is still in a register), so that kernel's
# compulsory traffic is the grayscale conversion's alone: 3 bytes read + 1 written per pixel.
KERNEL_BYTES_PER_PX = {"gray+hist": 4}
P_A_BYTES_PER_PX = 19
# The other two pipelines of the measurement plan; parity-test cases first, available here with
# --workload for the record (the default line is P-A).
OP_DILATE, OP_ERODE, OP_OTSU, OP_FEATURES = 9, 10, 12, 15
P_B = [
    {"type": OP_LINEAR, "kW": 11, "kH": 11, "fkernel": [1.0 / 121] * 121},
    {"type": OP_MEDIAN, "kW": 7, "kH": 7, "ikernel": [1] * 49},
]
P_C = [
    {"type": OP_OTSU}, {"type": OP_SOBEL}, {"type": OP_DILATE, "kW": 3, "kH": 3, "ikernel": [1] * 9},
    {"type": OP_ERODE, "kW": 3, "kH": 3, "ikernel": [0, 1, 0, 1, 1, 1, 0, 1, 0]}, {"type": OP_FEATURES},
]
WORKLOADS = {  # name: (ops, W, H, default images per GPU, stage-wise bytes per pixel, description)
    "P-A": (P_A, 768, 568, 500, 19, "full pipeline (gray, histogram, equalize, gaussian noise, salt-and-pepper 50, "
            "3x3 linear, 3x3 median, sobel)"),
    "P-B": (P_B, 4096, 4096, 8, 8, "large-stencil stress (gray, 11x11 linear box, 7x7 median)"),
    "P-C": (P_C, 768, 568, 500, 14, "segmentation path (gray, histogram, otsu, sobel, 3x3 dilation, 3x3 erosion, "
            "7 features per image)"),
}  # gray 4, hist 1, equalise 2, gaussian 3, salt-pepper 3, linear 2, median 2, sobel 2


def make_cell_batch(B, device, seed):
    """Cell-like synthetic images generated on the device: bright background N(200,8), 12
    dark elliptical blobs N(90,20), the three channels equal up to +-3 jitter."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((B, 3, H, W), dtype=torch.uint8, device=device)
    yy = torch.arange(H, device=device, dtype=torch.float32).view(1, H, 1)
    xx = torch.arange(W, device=device, dtype=torch.float32).view(1, 1, W)
    step = 50
    for s in range(0, B, step):
        n = min(step, B - s)
        base = 200.0 + 8.0 * torch.randn((n, H, W), generator=g, device=device)
        for _ in range(12):
            cy = torch.rand((n, 1, 1), generator=g, device=device) * H
            cx = torch.rand((n, 1, 1), generator=g, device=device) * W
            ry = torch.rand((n, 1, 1), generator=g, device=device) * (H / 8 - H / 30) + H / 30 + 1
            rx = torch.rand((n, 1, 1), generator=g, device=device) * (W / 8 - W / 30) + W / 30 + 1
            inside = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
            blob = 90.0 + 20.0 * torch.randn((n, H, W), generator=g, device=device)
            base = torch.where(inside, blob, base)
        base = torch.round(base)
        for c in range(3):
            jit = torch.randint(-3, 4, (n, H, W), generator=g, device=device).to(torch.float32)
            out[s:s + n, c] = torch.clamp(base + jit, 0, 255).to(torch.uint8)
    return out


def cpu_baseline(rgb_dev, n_threads, ops):
    """The CPU oracle (a port of the reference's CUDA kernels, oracle/ia_oracle.c) on a
    bounded sample of the same batch, on this machine's host cores."""
    import oracle_lib as O
    O.lib()
    x0, x1 = O.xorwow_planes(W * H)
    probe = rgb_dev[:n_threads].cpu().numpy()
    t0 = time.perf_counter()
    ops = [o for o in ops if o["type"] != OP_FEATURES]  # the image path; features are checked in the tests
    O.pipeline(probe, ops, x0, x1, threads=n_threads)
    dt = time.perf_counter() - t0
    per_image_cpu_s = dt * n_threads / probe.shape[0]
    # about 15 s of CPU work in total, at least one image per thread, at most what we hold
    n = int(max(n_threads, min(rgb_dev.shape[0], 15.0 / max(per_image_cpu_s, 1e-6))))
    n = max(n_threads, (n // n_threads) * n_threads)
    n = min(n, rgb_dev.shape[0])
    sample = rgb_dev[:n].cpu().numpy()
    t0 = time.perf_counter()
    out, _ = O.pipeline(sample, ops, x0, x1, threads=n_threads)
    dt = time.perf_counter() - t0
    # the same on one core, on about 2 s worth of images, for the per-core rate
    n1 = int(max(1, min(n, 2.0 / max(per_image_cpu_s, 1e-6))))
    t1 = time.perf_counter()
    O.pipeline(sample[:n1], ops, x0, x1, threads=1)
    dt1 = time.perf_counter() - t1
    return {
        "value": round(n * W * H / dt / 1e6, 2), "unit": "Mpixels/s", "cores": n_threads, "kind": "port",
        "sample": f"{n} of the batch's {W}x{H} images, same pipeline, OpenMP over images, {dt:.2f} s wall",
        "value_1_core": round(n1 * W * H / dt1 / 1e6, 2), "host_cpus": os.cpu_count(),
    }, out, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (default: the workload's)")
    ap.add_argument("--workload", default="P-A", choices=sorted(WORKLOADS))
    ap.add_argument("--stagewise", action="store_true", help="one pass per operation (no fusion)")
    ap.add_argument("--pointwise", action="store_true", help="fuse per-pixel runs only; stencils stay separate passes")
    ap.add_argument("--strips", action="store_true", help="also fuse runs of 3x3 stencils through LDS strips")
    ap.add_argument("--lanes", type=int, default=1, help="concurrent sub-batches per GPU (executor streams)")
    ap.add_argument("--settle-ms", type=float, default=100.0,
                    help="run the pipeline untimed for this long before the warm-up steps (clock ramp after the idle "
                         "gap of input generation); 0 disables")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the all-reduce path even with one rank (self-test of the multi-GPU path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="do not record per-kernel events in the timed steps (no roofline)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: there is no CPU fallback for the HIP path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        # RCCL writes its version banner (NCCL_DEBUG=VERSION) to stdout when the communicator comes
        # up; stdout is reserved for the one JSON line, so file descriptor 1 points at stderr until
        # the first collective has run.
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if world == 1:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29517")
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            else:
                dist.init_process_group("nccl", device_id=dev)
            probe = torch.zeros(1, dtype=torch.int64, device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    iak_lib.load()

    global W, H
    ops, W, H, default_B, bytes_per_px, what = WORKLOADS[args.workload]
    B = args.batch or default_B
    rgb = make_cell_batch(B, dev, seed=0x1A2B3C4D + rank)
    out = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    pipe = Pipeline(W, H, B, ops, fused=False if args.stagewise else ("pointwise" if args.pointwise else ("strips" if args.strips else True)))
    pipe.set_lanes(args.lanes)
    if any(o["type"] == OP_FEATURES for o in ops):
        feats = torch.zeros((B, 7), dtype=torch.float32, device=dev)
        pipe.bind_features(feats)
    # Two histogram buffers: the all-reduce of one batch's histogram runs on RCCL's stream while the
    # next batch is already being processed into the other buffer.
    hist_bufs = [torch.zeros(256, dtype=torch.int64, device=dev) for _ in range(2)]
    pending = [None, None]
    counter = [0]

    def step():
        # the batch's grayscale histogram is added to the buffer by the run; across GPUs the
        # dataset-level histogram is the one exchange of the path (2 KiB, latency-bound)
        k = counter[0] & 1
        counter[0] += 1
        if pending[k] is not None:
            pending[k].wait()  # the reduction issued two steps ago: orders this stream after it, no host wait
            pending[k] = None
        hist_bufs[k].zero_()
        pipe.run(rgb, out=out, hist_sum=hist_bufs[k])
        if dist is not None:
            pending[k] = dist.all_reduce(hist_bufs[k], op=dist.ReduceOp.SUM, async_op=True)

    def drain():
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    if args.settle_ms > 0:
        # The GPU has been idle or running small torch kernels while the inputs were made; its clocks take
        # some tens of milliseconds of load to come back up.  Untimed, like the warm-up steps.
        t_end = time.perf_counter() + args.settle_ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(10):
                step()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    # per-kernel HIP events on a sample of the timed steps (every 8th, at least one): each timed
    # step pays about 20 us for its ten event records
    every = 1 if args.steps < 16 else 8
    nrec = 0 if args.no_events else min((args.steps + every - 1) // every, 256)
    pipe.set_timing(nrec, every=every)
    drain()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-launch-group durations from the HIP events recorded inside the timed steps
    acc = None
    for sl in range(nrec):
        t = pipe.timings(sl)
        if acc is None:
            acc = [[n, 0.0, b, 0.0] for n, _, b, _ in t]
        for i, (_, ms, _, main_ms) in enumerate(t):
            acc[i][1] += ms
            acc[i][3] += main_ms
    groups = [(n, ms / max(nrec, 1), b, mm / max(nrec, 1)) for n, ms, b, mm in acc] if acc else []
    timed_images = pipe.timed_images()  # images in the sub-batch that carried the events

    if rank == 0:
        px_per_step = B * W * H * world
        value = px_per_step * args.steps / dt / 1e6
        # dominant kernel: the image kernel of the slowest group, timed alone (HIP events on the
        # executor's stream around that one launch)
        dom = max(groups, key=lambda g: g[3]) if groups else None
        roofline = None
        if dom:
            alg = KERNEL_BYTES_PER_PX.get(dom[0], dom[2])
            achieved = alg * timed_images * W * H / (dom[3] * 1e-3) / 1e9
            kernel_of = dict(KERNEL_OF_GROUP, **KERNEL_OF_GROUP_BY_WORKLOAD.get(args.workload, {}))
            roofline = {"bound": "hbm", "kernel": kernel_of.get(dom[0], dom[0]), "group": dom[0],
                        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                        "algorithmic_bytes_per_px": alg, "avg_ms": round(dom[3], 4),
                        "bytes_per_launch": alg * timed_images * W * H}
            tr = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tr):
                try:
                    rec = json.load(open(tr)).get(dom[0])
                    if rec and rec.get("batch") == B:
                        roofline["traffic"] = rec["hbm_bytes_per_launch"]
                except Exception:
                    pass
        line = {
            "metric": "Mpixels/sec (whole node) full filter pipeline", "value": round(value, 1), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"batch of {B} {W}x{H} RGB cell-like images per GPU, {what}, device-resident",
                       "images_per_gpu": B, "settle_ms": args.settle_ms, "mode": "stagewise" if args.stagewise else ("fused-pointwise" if args.pointwise else ("fused-strips" if args.strips else "fused")), "lanes": args.lanes,
                       "parallelism": f"image-wise shards x{world}, dataset-histogram all-reduce per step"},
            "pipeline_frac_of_hbm_roofline": round(value * 1e6 * bytes_per_px / world / (HBM_PEAK_GBS * 1e9), 4),
            "groups": [{"name": n, "ms": round(ms, 4), "kernel_ms": round(mm, 4), "images": timed_images,
                        "alg_GBps": round(KERNEL_BYTES_PER_PX.get(n, b) * timed_images * W * H / (mm * 1e-3) / 1e9, 1)
                        if mm > 0 else None}
                       for n, ms, b, mm in groups],
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline belongs to the 1-GPU line only
            threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
            threads = max(1, min(threads, os.cpu_count() or 1, B))
            base, ref_out, n = cpu_baseline(rgb, threads, ops)
            # the sample doubles as a full-size parity check of the timed path
            base["matches_gpu"] = bool(np.array_equal(out[:n].cpu().numpy(), ref_out))
            line["cpu_baseline"] = base
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
