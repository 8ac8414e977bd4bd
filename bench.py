#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of the full filter pipeline (grayscale -> histogram ->
equalise -> Gaussian noise -> salt-and-pepper -> 3x3 linear -> 3x3 median -> Sobel) on a
device-resident batch of synthetic 768x568 cell-like images, through the C ABI's batched
executor.  One process per GPU; images shard image-wise (every rank owns a full batch,
weak scaling); the only collective is the all-reduce of the dataset-level sums of every step
(the 256-bin histogram; with feature extraction also the per-class feature statistics), the
rows of --reduce-every consecutive steps travelling together.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no RANK in the environment the first form starts the second one itself,
as a child process, relays rank 0's line and exits with the child's status.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (default: the workload's)")
    ap.add_argument("--workload", default="P-A", choices=["P-A", "P-B", "P-C"])
    ap.add_argument("--linear-kernel", default="binomial16", choices=["binomial16", "config"],
                    help="P-A's 3x3 linear filter: the binomial [1,2,1;2,4,2;1,2,1]/16 or the reference configuration's "
                         "[1,1,1,10,1,1,1,1,1] (test/configs/input.json:43-53)")
    ap.add_argument("--stagewise", action="store_true", help="one pass per operation (no fusion)")
    ap.add_argument("--pointwise", action="store_true", help="fuse per-pixel runs only; stencils stay separate passes")
    ap.add_argument("--lanes", type=int, default=1, help="concurrent sub-batches per GPU (executor streams)")
    ap.add_argument("--settle-ms", type=float, default=100.0,
                    help="run the pipeline untimed for this long before the warm-up steps (clock ramp after the idle "
                         "gap of input generation); 0 disables")
    ap.add_argument("--reduce-every", type=int, default=8, choices=[1, 2, 4, 8, 16, 32],
                    help="steps whose data-set sums travel in one all-reduce (each step's sums keep a row of their own; "
                         "the collective is latency-bound, so fewer, larger ones cost less per step)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the all-reduce path even with one rank (self-test of the multi-GPU path)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N ranks that all use GPU 0 and reduce over gloo: exercises the launch, sharding and reduction "
                         "code on a one-GPU machine.  The line it prints is marked as a rehearsal, not a measurement")
    ap.add_argument("--rehearse-async", action="store_true",
                    help="with --rehearse-on-one-gpu: issue the gloo all-reduces asynchronously with the bookkeeping of the "
                         "RCCL path (two in flight, waited for when their rows come round again, the partial bucket flushed "
                         "by the drain), on host copies of the rows, instead of reducing each bucket synchronously")
    ap.add_argument("--hang-dump-s", type=float, default=0.0,
                    help="debugging aid: write every thread's Python stack to stderr if the run is still going after "
                         "this many seconds")
    ap.add_argument("--max-seconds", type=float, default=900.0,
                    help="a run that is still going after this long writes every thread's stack and the state of its "
                         "collectives to stderr and exits with status 3 (no rank of a multi-GPU job can hang silently); "
                         "0 disables")
    ap.add_argument("--dataset-out", default="",
                    help="rank 0 writes the data-set level sums the last step's reduction left (the histogram; with feature "
                         "extraction also the per-class count, feature sums and sums of squares) to this JSON file")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="default P-A run only: skip the short P-B / P-C / configuration-kernel measurements")
    ap.add_argument("--no-events", action="store_true", help="do not record per-kernel events in the timed steps (no roofline)")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """Start `--gpus` ranks of this script under torch.distributed.run as a CHILD process (this
    process has not touched the GPU: torch is not even imported yet), pass rank 0's JSON line
    through and return the child's exit status."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        if line.lstrip().startswith("{"):
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    return proc.wait()


if __name__ == "__main__":
    _early = parse_args()
    if _early.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(_early, sys.argv[1:]))

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from imageanalysis_amd import lib as iak_lib  # noqa: E402
from imageanalysis_amd.pipeline import Pipeline  # noqa: E402

W, H = 768, 568
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
# instruction issue: 256 CUs x 4 SIMDs, one wave64 vector instruction per 4 clocks per SIMD
N_SIMDS, CLOCK_HZ = 1024, 2.4e9

GAUSS16 = [1 / 16, 2 / 16, 1 / 16, 2 / 16, 4 / 16, 2 / 16, 1 / 16, 2 / 16, 1 / 16]
CONFIG_KERNEL = [1.0, 1.0, 1.0, 10.0, 1.0, 1.0, 1.0, 1.0, 1.0]  # test/configs/input.json:43-53
OP_EQUALIZE, OP_GAUSS, OP_SP, OP_LINEAR, OP_MEDIAN, OP_SOBEL = 1, 2, 3, 4, 6, 7
OP_DILATE, OP_ERODE, OP_OTSU, OP_FEATURES = 9, 10, 12, 15


def p_a_ops(linear_kernel="binomial16"):
    return [
        {"type": OP_EQUALIZE},
        {"type": OP_GAUSS, "mean": -1.0, "stddev": -1.0},
        {"type": OP_SP, "iparam": 50},
        {"type": OP_LINEAR, "kW": 3, "kH": 3, "fkernel": GAUSS16 if linear_kernel == "binomial16" else CONFIG_KERNEL},
        {"type": OP_MEDIAN, "kW": 3, "kH": 3, "ikernel": [1] * 9},
        {"type": OP_SOBEL},
    ]


P_A = p_a_ops()
# launch group -> the HIP kernel that does its image pass (names as rocprofv3 prints them)
KERNEL_OF_GROUP = {
    "gray+hist": "iak::gray_kernel<0, true>",
    "equalize+gauss+salt_pepper+linear": "iak::lgs_stencil8_flat_kernel<iak::Lin8Swar>",
    "equalize+gauss+salt_pepper": "iak::chain_apply_kernel<false>",
    "linear": "iak::stencil8_kernel<iak::Lin8Swar, 8, false, false>",
    "median": "iak::stencil8_kernel<iak::Median8, 8, false, false>",
    "sobel": "iak::stencil8_kernel<iak::Sobel8, 8, false, false>",
}
# kernels behind the group names of the other workloads, where they differ from P-A's
KERNEL_OF_GROUP_BY_WORKLOAD = {
    "P-A/config": {"equalize+gauss+salt_pepper+linear": "iak::lgs_stencil8_flat_kernel<iak::Lin8Pk16<false> >",
                   "linear": "iak::stencil8_kernel<iak::Lin8Pk16<false>, 8, false, false>"},
    "P-B": {"linear": "iak::bitslice_box_kernel<11, 11>", "median": "iak::bitslice_median_kernel<7, 7>"},
    "P-C": {"otsu": "iak::chain_apply_kernel<false>", "otsu+sobel": "iak::binary_table_stencil_kernel",
            "dilate+erode+hist": "iak::morph_bits_kernel<2, false, true>",
            "otsu+sobel+dilate+erode+hist": "iak::morph_bits_kernel<3, true, true>", "features": "iak::perimeter_bits_kernel"},
}
P_B = [
    {"type": OP_LINEAR, "kW": 11, "kH": 11, "fkernel": [1.0 / 121] * 121},
    {"type": OP_MEDIAN, "kW": 7, "kH": 7, "ikernel": [1] * 49},
]
P_C = [
    {"type": OP_OTSU}, {"type": OP_SOBEL}, {"type": OP_DILATE, "kW": 3, "kH": 3, "ikernel": [1] * 9},
    {"type": OP_ERODE, "kW": 3, "kH": 3, "ikernel": [0, 1, 0, 1, 1, 1, 0, 1, 0]}, {"type": OP_FEATURES},
]
# name: (ops, W, H, default images per GPU, stage-wise bytes per pixel of the reference's one-kernel-per-operation
# flow, description).  Stage-wise: gray 4, hist 1, equalise 2, gaussian 3, salt-pepper 3, linear 2, median 2, sobel 2.
WORKLOADS = {
    "P-A": (P_A, 768, 568, 500, 19, "full pipeline (gray, histogram, equalize, gaussian noise, salt-and-pepper 50, "
            "3x3 linear, 3x3 median, sobel)"),
    "P-B": (P_B, 4096, 4096, 8, 8, "large-stencil stress (gray, 11x11 linear box, 7x7 median)"),
    "P-C": (P_C, 768, 568, 500, 14, "segmentation path (gray, histogram, otsu, sobel, 3x3 dilation, 3x3 erosion, "
            "7 features per image, per-class feature statistics)"),
}
CONFIG3_IMAGES_PER_GPU = 1250  # BASELINE config 3: 10 000 images of 768x568 sharded image-wise over 8 GPUs
LLC_BYTES = 256 << 20  # Infinity Cache: an intermediate batch no larger than this is still on the die for the next pass
HIST_POOL = 64  # int64[256] rows the steps' data-set histograms take turns through (see Measurement)
N_CLASSES = 6  # the data set's class count (imageMachineLearning.cu:212); synthetic label = image index % 6


def make_cell_batch(B, device, seed):
    """Cell-like synthetic images generated on the device: bright background N(200,8), 12
    dark elliptical blobs N(90,20), the three channels equal up to +-3 jitter."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((B, 3, H, W), dtype=torch.uint8, device=device)
    yy = torch.arange(H, device=device, dtype=torch.float32).view(1, H, 1)
    xx = torch.arange(W, device=device, dtype=torch.float32).view(1, 1, W)
    step = max(1, min(50, (50 * 768 * 568) // (W * H)))
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


def cpu_baseline(rgb_dev, n_threads, ops, cpu_seconds=24.0):
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
    per_image_cpu_s = dt * min(n_threads, probe.shape[0]) / probe.shape[0]
    # about cpu_seconds of CPU work in total, at least one image per thread, at most what we hold
    n = int(max(n_threads, min(rgb_dev.shape[0], cpu_seconds / max(per_image_cpu_s, 1e-6))))
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
        "value": float(f"{n * W * H / dt / 1e6:.3g}"), "unit": "Mpixels/s", "cores": n_threads, "kind": "port",
        "sample": f"{n} of the batch's {W}x{H} images, same pipeline, OpenMP over images, {dt:.2f} s wall",
        "value_1_core": float(f"{n1 * W * H / dt1 / 1e6:.3g}"), "host_cpus": os.cpu_count(),
    }, out, n


def _load_profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    try:
        return json.load(open(path))
    except Exception:
        return {}


class Watchdog:
    """Ends the process (status 3) if the run outlives its budget: dumps every thread's Python stack and what
    `state()` reports (which step, which collectives are in flight) first."""

    def __init__(self, seconds, state):
        import threading
        self.deadline = time.monotonic() + seconds
        self.state = state
        self.done = threading.Event()
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        import faulthandler
        while not self.done.wait(1.0):
            if time.monotonic() > self.deadline:
                sys.stderr.write(f"bench.py: still running at its time limit; state: {self.state()}\n")
                faulthandler.dump_traceback(file=sys.stderr)
                sys.stderr.flush()
                os._exit(3)

    def stop(self):
        self.done.set()


def init_dist(args, world, local_rank, dev):
    """RCCL (or gloo for the one-GPU rehearsal) up, first collective done.  RCCL writes its
    version banner to stdout when the communicator comes up; stdout is reserved for the one JSON
    line, so file descriptor 1 points at stderr until the first collective has run."""
    import torch.distributed as dist
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        import datetime
        backend = "gloo" if args.rehearse_on_one_gpu else "nccl"
        # a collective that has not completed after this long aborts the job (gloo raises; for RCCL torch's
        # watchdog thread ends the process) instead of waiting for a peer for ever
        kw = {"timeout": datetime.timedelta(seconds=180)}
        if backend == "nccl":
            kw["device_id"] = dev
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            dist.init_process_group(backend, rank=0, world_size=1, **kw)
        else:
            dist.init_process_group(backend, **kw)
        probe = torch.zeros(1, dtype=torch.int64, device=dev)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    return dist


class _HostCollective:
    """An asynchronous collective on a host copy of device rows; wait() brings the result back."""

    def __init__(self, work, host, rows):
        self.work, self.host, self.rows = work, host, rows

    def wait(self):
        self.work.wait()
        self.rows.copy_(self.host)


class Measurement:
    """One workload on this rank's GPU: pipeline, buffers, the step function with its (optional)
    asynchronous dataset reduction, and the timed loop."""

    def __init__(self, args, name, ops, B, dev, rank, dist, mode):
        global W, H
        _, W, H, _, self.stagewise_bytes, self.what = WORKLOADS[name]
        self.name, self.ops, self.B, self.dev, self.dist = name, ops, B, dev, dist
        self.rgb = make_cell_batch(B, dev, seed=0x1A2B3C4D + rank)
        self.out = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
        self.pipe = Pipeline(W, H, B, ops, fused=mode)
        self.pipe.set_lanes(args.lanes)
        self.feats = None
        self.reduce_every = args.reduce_every
        # The batch's histogram is ADDED to the buffer a step hands over, so every step needs a cleared
        # int64[256] of its own.  A fill per step is a launch per step (4.6 us for 2 KiB): the steps take
        # turns through a pool of rows instead, cleared half a pool at a time just before that half comes
        # round again.  Rows also let ONE collective carry the sums of several steps: the all-reduce of a
        # few KiB is latency-bound (and costs the stream two cross-stream waits), so the rows of
        # `reduce_every` consecutive steps are reduced together, asynchronously, while later batches are
        # processed into other rows.
        self.hist_pool = torch.zeros((HIST_POOL, 256), dtype=torch.int64, device=dev)
        self.hist_rows = [self.hist_pool[r] for r in range(HIST_POOL)]
        if any(o["type"] == OP_FEATURES for o in ops):
            from imageanalysis_amd.dist import PackedDatasetStats
            self.feats = torch.zeros((B, 7), dtype=torch.float32, device=dev)
            self.pipe.bind_features(self.feats)
            self.labels = ((torch.arange(B, device=dev, dtype=torch.int64) + rank * B) % N_CLASSES).to(torch.int32)
            # histogram + per-class count / sum / sum of squares of a step packed into one row
            self.packed_pool = torch.zeros((HIST_POOL, PackedDatasetStats.length(N_CLASSES)), dtype=torch.float64, device=dev)
            self.packed = [PackedDatasetStats(N_CLASSES, dev, buf=self.packed_pool[r]) for r in range(HIST_POOL)]
        self.pending = []       # collectives in flight, oldest first
        self.via_host = bool(args.rehearse_on_one_gpu)
        self.host_async = bool(args.rehearse_async)
        self.issued = self.waited = 0  # collectives handed to the backend / waited for (for the watchdog's report)
        self.phase = "set-up"
        self.bucket_start = 0   # first row of the steps not handed to a collective yet
        self.bucket_rows = 0
        self.counter = 0

    def _reduce_bucket(self):
        """One asynchronous all-reduce over the rows of the steps since the last one."""
        if self.dist is None or self.bucket_rows == 0:
            self.bucket_rows = 0
            return
        pool = self.packed_pool if self.feats is not None else self.hist_pool
        rows = pool[self.bucket_start:self.bucket_start + self.bucket_rows]
        self.bucket_rows = 0
        if self.via_host and not self.host_async:
            # The gloo rehearsal reduces a host copy: gloo's collectives on DEVICE tensors, issued asynchronously
            # between kernel launches, stall for good with two ranks on one GPU even when both ranks issue exactly
            # the same collectives (both ranks end up in drain(), gloo's transport waiting for a receive that never
            # comes: its worker threads and device-side copies, not this bookkeeping -- RCCL's wait() is a
            # stream-side dependency with no worker thread).
            host = rows.cpu()
            self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM)
            rows.copy_(host)
            return
        # At most two collectives in flight (wait() orders this stream after the collective without
        # blocking the host).
        while len(self.pending) >= 2:
            self._wait(self.pending.pop(0))
        if self.via_host:
            # the same issue / wait pattern over gloo, on a host copy that travels back when the collective is waited for
            host = rows.cpu()
            self.pending.append(_HostCollective(self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM, async_op=True), host, rows))
        else:
            self.pending.append(self.dist.all_reduce(rows, op=self.dist.ReduceOp.SUM, async_op=True))
        self.issued += 1

    def _wait(self, work):
        """Orders this stream behind the collective.  RCCL: no host wait.  gloo: blocks the host until its worker
        thread has run the collective, and raises after the process group's time limit."""
        work.wait()
        self.waited += 1

    def state(self):
        return {"phase": self.phase, "step": self.counter, "collectives_issued": self.issued,
                "collectives_waited_for": self.waited, "in_flight": len(self.pending), "bucket_rows": self.bucket_rows}

    def step(self):
        # the batch's grayscale histogram is added to this step's row by the run; across GPUs the
        # dataset-level sums are the one exchange of the path (a few KiB, latency-bound)
        row = self.counter % HIST_POOL
        if self.counter >= HIST_POOL and row % (HIST_POOL // 2) == 0:
            # this half comes round again: its collectives were issued at least half a pool ago
            for w in self.pending[:-1]:
                self._wait(w)
            del self.pending[:-1]
            self.hist_pool[row:row + HIST_POOL // 2].zero_()
        self.counter += 1
        hist = self.hist_rows[row]
        self.pipe.run(self.rgb, out=self.out, hist_sum=hist)
        if self.feats is not None:
            self.packed[row].fill(hist, self.feats, self.labels)
        if self.bucket_rows == 0:
            self.bucket_start = row
        self.bucket_rows += 1
        if (row + 1) % self.reduce_every == 0:  # buckets end on multiples of reduce_every: never across the pool's end
            self._reduce_bucket()

    def drain(self):
        self._reduce_bucket()
        for w in self.pending:
            self._wait(w)
        self.pending = []

    def run(self, steps, warmup, settle_ms, events=True):
        self.phase = "settling"
        if settle_ms > 0 and self.dist is None:
            # The GPU has been idle or running small torch kernels while the inputs were made; its clocks take
            # some tens of milliseconds of load to come back up.  Untimed, like the warm-up steps.
            t_end = time.perf_counter() + settle_ms * 1e-3
            while time.perf_counter() < t_end:
                for _ in range(10):
                    self.step()
                torch.cuda.synchronize()
        elif settle_ms > 0:
            # The same with several ranks, by COUNT: steps issue collectives, so every rank has to run the
            # same number of them -- ten steps are timed and the slowest rank's estimate decides how many follow.
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                self.step()
            torch.cuda.synchronize()
            per_step = max((time.perf_counter() - t0) / 10, 1e-6)
            more = torch.tensor([int(settle_ms * 1e-3 / per_step)], dtype=torch.int64, device=self.dev)
            self.dist.all_reduce(more, op=self.dist.ReduceOp.MIN)
            for _ in range(min(int(more.item()), 2000)):
                self.step()
            torch.cuda.synchronize()
        self.phase = "warm-up"
        for _ in range(warmup):
            self.step()
        # per-kernel HIP events on a sample of the timed steps (every 8th, at least one): each timed
        # step pays about 20 us for its ten event records
        every = 1 if steps < 16 else 8
        nrec = 0 if not events else min((steps + every - 1) // every, 256)
        self.pipe.set_timing(nrec, every=every)
        self.drain()
        dist = self.dist
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        self.phase = "timed steps"
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.phase = "drain"
        self.drain()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        self.phase = "report"
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        # per-launch-group durations from the HIP events recorded inside the timed steps
        acc = None
        for sl in range(nrec):
            t = self.pipe.timings(sl)
            if acc is None:
                acc = [[n, 0.0, b, 0.0] for n, _, b, _ in t]
            for i, (_, ms, _, main_ms) in enumerate(t):
                acc[i][1] += ms
                acc[i][3] += main_ms
        self.groups = [(n, ms / max(nrec, 1), b, mm / max(nrec, 1)) for n, ms, b, mm in acc] if acc else []
        self.pipe.set_timing(0)
        return dt

    def report(self, dt, steps, world, variant=None):
        """Throughput, per-group figures and the dominant kernel's roofline, on the bytes the
        executed plan moves (image read + image written per launch group)."""
        B = self.B
        timed_images = self.pipe.timed_images()
        moved = self.pipe.moved_bytes()
        px = B * W * H
        value = px * world * steps / dt / 1e6
        kernel_of = dict(KERNEL_OF_GROUP, **KERNEL_OF_GROUP_BY_WORKLOAD.get(variant or self.name, {}))
        # Counter figures are not measured in this run: they come from the committed summaries of earlier rocprofv3
        # --pmc passes (tools/profile_round.sh) and are used only where the kernel's name and the batch are the
        # same as in this run; otherwise the field is left out (null in `roofline`).
        traffic = _load_profile_json("traffic.json" if B == WORKLOADS[self.name][3] else f"traffic_B{B}.json")
        valu_all = _load_profile_json("valu.json")
        valu = valu_all.get(variant or self.name, {})
        valu_batch = WORKLOADS[self.name][3]  # the counter passes run the workload's default batch
        tpx = timed_images * W * H
        # an intermediate batch no larger than the Infinity Cache is still on the die when the next pass reads it
        # (the passes store it with the default policy for that reason): such a pass does not read from HBM
        llc = tpx <= LLC_BYTES
        groups = []
        used_traffic = used_valu = False
        # Feature extraction does not read an image whose Otsu mask is empty (its perimeter is 0 without looking:
        # every two-valued image with both values present).  Such an image has area 0 and perimeter 0 in the
        # results of the last step; the group's moved bytes count the others only.
        features_read_frac = 1.0
        if self.feats is not None:
            f = self.feats.cpu()
            features_read_frac = 1.0 - float(((f[:, 2] == 0) & (f[:, 3] == 0)).float().mean())
        moved = [mv * features_read_frac if n == "features" else mv for (n, _, _, _), mv in zip(self.groups, moved)] \
            if self.groups else list(moved)
        for gi, ((n, ms, stage_b, mm), mv) in enumerate(zip(self.groups, moved)):
            g = {"name": n, "kernel": kernel_of.get(n, n), "ms": round(ms, 4), "kernel_ms": round(mm, 4), "images": timed_images,
                 "moved_bytes_per_px": round(mv, 3) if n == "features" else mv, "stagewise_bytes_per_px": stage_b,
                 "input_from": "hbm" if (gi == 0 or not llc) else "infinity-cache"}
            if mm > 0:
                gbps = mv * tpx / (mm * 1e-3) / 1e9
                g["moved_GBps"] = round(gbps, 1)
                # moved bytes / duration over the HBM peak; a true HBM fraction only where input_from is "hbm"
                g["hbm_frac"] = round(gbps / HBM_PEAK_GBS, 4)
                g["bound"] = "hbm"
                v = valu.get(kernel_of.get(n, n))
                if v and B == valu_batch:  # the counter passes run the default batch
                    # share of the launch during which the vector pipes were issuing, from the SQ counter pass
                    issue_s = v["valu_busy_simd_cycles_per_launch"] / N_SIMDS / CLOCK_HZ
                    # (the counter tallies some instructions at more than one issue slot, so a kernel that does nothing
                    # but vector instructions can come out a little above 1: clipped)
                    g["valu_issue_frac"] = min(1.0, round(issue_s / (mm * 1e-3), 4))
                    used_valu = True
                    if g["valu_issue_frac"] > g["hbm_frac"]:
                        g["bound"] = "valu"
                tr = traffic.get(n)
                if tr and tr.get("batch") == B and tr.get("kernel") == g["kernel"]:
                    g["traffic_bytes_per_launch"] = tr["hbm_bytes_per_launch"]
                    used_traffic = True
            if n == "features" and features_read_frac < 1.0:
                g["images_read_frac"] = round(features_read_frac, 4)
            groups.append(g)
        moved_total = sum(moved)
        line = {
            "value": round(value, 1), "ms_per_step": round(dt / steps * 1e3, 4),
            "pipeline_moved_bytes_per_px": round(moved_total, 3) if self.feats is not None else moved_total,
            # fraction of HBM peak on the bytes the executed plan has to move (<= 1 by construction)
            "pipeline_hbm_frac_moved_bytes": round(value * 1e6 / world * moved_total / (HBM_PEAK_GBS * 1e9), 4),
            "pipeline_stagewise_bytes_per_px": self.stagewise_bytes,
            "intermediates_llc_resident": bool(llc),
            "traffic_source": (f"profiles/{'traffic.json' if B == WORKLOADS[self.name][3] else f'traffic_B{B}.json'} "
                               f"(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, batch {B})") if used_traffic else None,
            "valu_source": f"profiles/valu.json (rocprofv3 --pmc SQ_ACTIVE_INST_VALU pass, batch {valu_batch})" if used_valu else None,
            "groups": groups,
        }
        if groups and all("traffic_bytes_per_launch" in g for g in groups):
            tb = sum(g["traffic_bytes_per_launch"] for g in groups)
            # the same with the HBM bytes the counters saw (profiles/traffic.json)
            line["pipeline_hbm_frac_counter_traffic"] = round(tb / (dt / steps) / (HBM_PEAK_GBS * 1e9), 4)
        dom = max(groups, key=lambda g: g["kernel_ms"]) if groups else None
        if dom and dom["kernel_ms"] > 0:
            alg = dom["moved_bytes_per_px"]
            achieved = alg * tpx / (dom["kernel_ms"] * 1e-3) / 1e9
            line["roofline"] = {"bound": dom.get("bound", "hbm"), "kernel": dom["kernel"], "group": dom["name"],
                                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": dom.get("traffic_bytes_per_launch"),
                                "algorithmic_bytes_per_px": alg, "avg_ms": dom["kernel_ms"], "bytes_per_launch": alg * tpx,
                                "traffic_source": line["traffic_source"] if dom.get("traffic_bytes_per_launch") else None}
            if "valu_issue_frac" in dom:
                line["roofline"]["valu_issue_frac"] = dom["valu_issue_frac"]
        return line

    def close(self):
        self.pipe.close()
        self.rgb = self.out = self.feats = None
        torch.cuda.empty_cache()


def _mode_of(args):
    return False if args.stagewise else ("pointwise" if args.pointwise else True)


def _threads(B):
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    return max(1, min(threads, os.cpu_count() or 1, B))


def side_measurement(args, name, ops, B, dev, variant, steps, cpu_seconds):
    """A short run of another workload for the record: ms per step, dominant kernel and its
    fraction, and a full-size comparison of a sample with the CPU oracle."""
    m = Measurement(args, name, ops, B, dev, 0, None, True)
    dt = m.run(steps, 3, 50.0)
    rep = m.report(dt, steps, 1, variant=variant)
    res = {"workload": f"{B} x {W}x{H}: {m.what}", "ms_per_step": rep["ms_per_step"], "value": rep["value"],
           "unit": "Mpixels/s", "steps": steps, "moved_bytes_per_px": rep["pipeline_moved_bytes_per_px"],
           "hbm_frac_moved_bytes": rep["pipeline_hbm_frac_moved_bytes"],
           "groups": [{k: g[k] for k in ("name", "kernel_ms", "moved_bytes_per_px", "hbm_frac", "bound", "valu_issue_frac")
                       if k in g} for g in rep["groups"]]}
    if "roofline" in rep:
        r = rep["roofline"]
        res["dominant"] = {"kernel": r["kernel"], "avg_ms": r["avg_ms"], "frac": r["frac"], "bound": r["bound"]}
    if not args.no_cpu_baseline:
        base, ref_out, n = cpu_baseline(m.rgb, _threads(B), ops, cpu_seconds=cpu_seconds)
        res["cpu_Mpixels_per_s"] = base["value"]
        res["cpu_cores"] = base["cores"]
        res["cpu_sample_images"] = n
        res["matches_cpu"] = bool(np.array_equal(m.out[:n].cpu().numpy(), ref_out))
    m.close()
    return res


def main():
    args = parse_args()
    if args.hang_dump_s > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.hang_dump_s, exit=False, file=sys.stderr)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: there is no CPU fallback for the HIP path")
    n_dev = torch.cuda.device_count()
    if args.rehearse_on_one_gpu:
        local_rank = 0
    elif local_rank >= n_dev:
        raise SystemExit(f"rank {rank}: local rank {local_rank} but this machine has {n_dev} GPU(s)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = init_dist(args, world, local_rank, dev) if (world > 1 or args.force_dist) else None
    iak_lib.load()

    ops = p_a_ops(args.linear_kernel) if args.workload == "P-A" else WORKLOADS[args.workload][0]
    # One GPU: BASELINE config 2's batch (500 images).  Several GPUs: config 3's per-GPU share -- 10 000 images
    # over 8 GPUs = 1250 each, the same share at every GPU count (weak scaling).
    config3 = world > 1 and not args.batch and args.workload != "P-B" and not args.rehearse_on_one_gpu
    B = args.batch or (CONFIG3_IMAGES_PER_GPU if config3 else WORKLOADS[args.workload][3])
    variant = "P-A/config" if (args.workload == "P-A" and args.linear_kernel == "config") else None
    m = Measurement(args, args.workload, ops, B, dev, rank, dist, _mode_of(args))
    dog = Watchdog(args.max_seconds, m.state) if args.max_seconds > 0 else None
    dt = m.run(args.steps, args.warmup, args.settle_ms, events=not args.no_events)
    if dog:
        dog.stop()

    if rank == 0:
        rep = m.report(dt, args.steps, world, variant=variant)
        lin = ("3x3 linear = binomial [1,2,1;2,4,2;1,2,1]/16 (kernel Lin8Swar)" if args.linear_kernel == "binomial16" else
               "3x3 linear = [1,1,1,10,1,1,1,1,1] of test/configs/input.json (kernel Lin8Pk16)")
        what = m.what + ("; " + lin if args.workload == "P-A" else "")
        reduction = "dataset histogram" + (" + per-class feature statistics" if m.feats is not None else "")
        line = {
            "metric": "Mpixels/sec (whole node) full filter pipeline", "value": rep["value"], "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": rep["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": (f"BASELINE config 3 ({CONFIG3_IMAGES_PER_GPU * 8} images sharded image-wise over 8 GPUs): its per-GPU "
                                    f"share, " if B == CONFIG3_IMAGES_PER_GPU and world > 1 else "") +
                                   f"batch of {B} {W}x{H} RGB cell-like images per GPU, {what}, device-resident",
                       "images_per_gpu": B, "settle_ms": args.settle_ms,
                       "mode": "stagewise" if args.stagewise else ("fused-pointwise" if args.pointwise else "fused"),
                       "lanes": args.lanes,
                       "parallelism": (f"image-wise shards x{world}, the {reduction} of every step all-reduced, "
                                       f"{args.reduce_every} steps' rows per collective" if m.dist is not None else
                                       f"one GPU: the {reduction} of every step stays on this GPU (no collective)")},
        }
        # what the last step's reduction left: the dataset-level sums over all ranks
        k_last = (m.counter - 1) % HIST_POOL
        if m.feats is not None:
            count, mean, _ = m.packed[k_last].class_stats()
            line["dataset"] = {"histogram_pixels": int(m.packed[k_last].histogram().sum().item()),
                               "class_counts": [int(c) for c in count.tolist()],
                               "class_mean_area": [round(float(a), 3) for a in mean[:, 2].tolist()]}
        else:
            line["dataset"] = {"histogram_pixels": int(m.hist_rows[(m.counter - 1) % HIST_POOL].sum().item())}
        if args.dataset_out:
            sums = {}
            if m.feats is not None:
                pk = m.packed[k_last]
                sums = {"histogram": [int(v) for v in pk.histogram().tolist()], "class_count": [int(v) for v in pk.count.round().tolist()],
                        "class_sum": pk.sum.tolist(), "class_sumsq": pk.sumsq.tolist()}
            else:
                sums = {"histogram": [int(v) for v in m.hist_rows[k_last].tolist()]}
            with open(args.dataset_out, "w") as f:
                json.dump(sums, f)
        if args.rehearse_on_one_gpu:
            line["rehearsal"] = ("all ranks on GPU 0, gloo reduction of host copies" +
                                 (", issued asynchronously as the RCCL path issues its collectives" if args.rehearse_async else "") +
                                 ": exercises the multi-rank code path; not a measurement")
        for k in ("pipeline_moved_bytes_per_px", "pipeline_hbm_frac_moved_bytes", "pipeline_hbm_frac_counter_traffic",
                  "pipeline_stagewise_bytes_per_px", "intermediates_llc_resident", "traffic_source", "valu_source", "groups",
                  "roofline"):
            if k in rep:
                line[k] = rep[k]
        solo = world == 1 and not args.force_dist
        if not args.no_cpu_baseline and solo:  # the CPU baseline belongs to the 1-GPU line only
            base, ref_out, n = cpu_baseline(m.rgb, _threads(B), ops)
            # the sample doubles as a full-size parity check of the timed path
            base["matches_gpu"] = bool(np.array_equal(m.out[:n].cpu().numpy(), ref_out))
            line["cpu_baseline"] = base
        if solo and args.workload == "P-A" and not variant and not args.no_other_workloads and _mode_of(args) is True \
                and not args.batch:
            m.close()
            other = {}
            other["P-A with the configuration's linear kernel"] = side_measurement(
                args, "P-A", p_a_ops("config"), WORKLOADS["P-A"][3], dev, "P-A/config", 30, 6.0)
            other["P-B"] = side_measurement(args, "P-B", P_B, WORKLOADS["P-B"][3], dev, None, 10, 3.0)
            other["P-C"] = side_measurement(args, "P-C", P_C, WORKLOADS["P-C"][3], dev, None, 30, 6.0)
            line["other_workloads"] = other
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
