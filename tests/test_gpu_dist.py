"""The multi-GPU code path on the one GPU of the test box: config 3's per-GPU share (1250 images)
through the reduction path with one RCCL rank, bench.py starting its own ranks, and the
per-class feature statistics fed by the executor's feature output."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config3_shard_of_1250_images_through_the_reduction_path():
    """10,000 images over 8 GPUs = 1250 per GPU.  One rank, RCCL initialised, the asynchronous
    all-reduce issued every step as in the N-GPU job; images 0, 624 and 1249 and the int64
    dataset histogram against the oracle."""
    import bench
    args = bench.parse_args(["--force-dist", "--batch", "1250", "--steps", "3", "--warmup", "1"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist = bench.init_dist(args, 1, 0, dev)
    try:
        m = bench.Measurement(args, "P-A", bench.P_A, 1250, dev, 0, dist, True)
        dt = m.run(70, 1, 0.0)  # more steps than the pool of histogram rows holds: rows come round again
        assert dt > 0
        got_hist = m.hist_rows[(m.counter - 1) % bench.HIST_POOL].cpu().numpy()
        assert got_hist.dtype == np.int64
        W, H = 768, 568
        x0, x1 = O.xorwow_planes(W * H)
        pick = [0, 624, 1249]
        want, _ = O.pipeline(m.rgb[pick].cpu().numpy(), bench.P_A, x0, x1, threads=3)
        assert np.array_equal(m.out[pick].cpu().numpy(), want)
        # the dataset histogram: the grayscale histograms of all 1250 images, summed
        want_hist = np.zeros(256, np.int64)
        for s in range(0, 1250, 250):
            _, h = O.pipeline(m.rgb[s:s + 250].cpu().numpy(), [], x0, x1, threads=min(16, os.cpu_count() or 1))
            want_hist += h.astype(np.int64)
        assert np.array_equal(got_hist, want_hist)
        assert int(got_hist.sum()) > 2 ** 29  # 545 Mpx in the shard: 32-bit counts would be close to their end at 8 shards
        m.close()
    finally:
        dist.destroy_process_group()


def _bench_line(*flags):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def _rank_batches(n_ranks, B):
    """The batches the ranks of a bench run hold: rank r's generator is seeded with 0x1A2B3C4D + r."""
    import bench
    bench.W, bench.H = 768, 568
    return np.concatenate([bench.make_cell_batch(B, torch.device("cuda"), seed=0x1A2B3C4D + r).cpu().numpy() for r in range(n_ranks)])


def _oracle_dataset_histogram(rgb):
    x0, x1 = O.xorwow_planes(768 * 568)
    _, h = O.pipeline(rgb, [], x0, x1, threads=min(16, os.cpu_count() or 1))
    return h.astype(np.int64)


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: the parent starts two ranks as a child
    process and relays rank 0's line.  Both ranks share this box's one GPU and reduce over gloo
    (a rehearsal of the code path, not a measurement).  The reduced data-set histogram is the oracle's
    over both ranks' batches (test/test.cu:407-411 sums the same rows on one GPU)."""
    sums = str(tmp_path / "sums.json")
    line = _bench_line("--gpus", "2", "--rehearse-on-one-gpu", "--steps", "3", "--warmup", "1", "--batch", "12",
                       "--settle-ms", "0", "--no-cpu-baseline", "--dataset-out", sums)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and "rehearsal" in line
    assert line["config"]["images_per_gpu"] == 12
    want = _oracle_dataset_histogram(_rank_batches(2, 12))
    got = np.array(json.load(open(sums))["histogram"], dtype=np.int64)
    assert np.array_equal(got, want)
    assert line["dataset"]["histogram_pixels"] == int(want.sum())
    assert abs(line["value"] - 2 * 12 * 768 * 568 / (line["ms_per_step"] * 1e-3) / 1e6) / line["value"] < 1e-3


def test_bench_two_ranks_issue_the_same_collectives_over_many_steps(tmp_path):
    """Two ranks, the default settling phase (its length is agreed between the ranks: steps issue
    collectives, so both must run the same number), and more timed steps than the pools of per-step rows
    hold, so rows come round again and several collectives of 8 steps' rows each are issued on the way.
    The collectives are issued with the RCCL path's bookkeeping: asynchronously, two in flight, waited for
    when their rows come round again, the partial bucket flushed by the drain (--rehearse-async; over gloo
    they travel as host copies of the rows)."""
    sums = str(tmp_path / "sums.json")
    line = _bench_line("--gpus", "2", "--rehearse-on-one-gpu", "--rehearse-async", "--steps", "150", "--warmup", "3",
                       "--batch", "12", "--no-cpu-baseline", "--max-seconds", "200", "--dataset-out", sums)
    assert line["n_gpus"] == 2 and line["steps"] == 150
    # the last step's row, summed over both ranks, and nothing else
    want = _oracle_dataset_histogram(_rank_batches(2, 12))
    assert np.array_equal(np.array(json.load(open(sums))["histogram"], dtype=np.int64), want)


def test_bench_two_ranks_reduce_class_statistics_in_one_collective(tmp_path):
    """P-C on two ranks: the packed row (histogram, per-class image count, feature sums and sums of squares,
    test/test.cu:541-550) after the reduction against the oracle's features of both ranks' images."""
    import bench
    sums = str(tmp_path / "sums.json")
    line = _bench_line("--gpus", "2", "--rehearse-on-one-gpu", "--workload", "P-C", "--steps", "2", "--warmup", "1",
                       "--batch", "9", "--settle-ms", "0", "--no-cpu-baseline", "--dataset-out", sums)
    assert line["n_gpus"] == 2
    # labels are (global image index) % 6 over 2 x 9 images
    assert line["dataset"]["class_counts"] == [3, 3, 3, 3, 3, 3]
    assert "per-class feature statistics" in line["config"]["parallelism"]
    got = json.load(open(sums))
    rgb = _rank_batches(2, 9)
    assert np.array_equal(np.array(got["histogram"], dtype=np.int64), _oracle_dataset_histogram(rgb))
    x0, x1 = O.xorwow_planes(768 * 568)
    image_ops = [o for o in bench.P_C if o["type"] != bench.OP_FEATURES]
    img, _ = O.pipeline(rgb, image_ops, x0, x1, threads=min(16, os.cpu_count() or 1))
    feats = np.stack([O.features(img[b]) for b in range(18)]).astype(np.float32).astype(np.float64)
    labels = np.arange(18) % bench.N_CLASSES
    assert got["class_count"] == [int((labels == c).sum()) for c in range(bench.N_CLASSES)]
    for c in range(bench.N_CLASSES):
        sel = feats[labels == c]
        assert np.allclose(np.array(got["class_sum"][c]), sel.sum(axis=0), rtol=1e-12, atol=0)
        assert np.allclose(np.array(got["class_sumsq"][c]), (sel * sel).sum(axis=0), rtol=1e-12, atol=0)


def test_class_statistics_from_the_executor_features_match_the_oracle():
    """IAK_OP_FEATURES output + labels -> packed per-class sums -> mean / std per class, against the
    oracle's features of the same images (test/test.cu:541-550 collects the same rows)."""
    import synth
    from imageanalysis_amd.dist import PackedDatasetStats
    from imageanalysis_amd.pipeline import Pipeline
    import bench
    W, H, B, C = 64, 48, 10, 3
    rgb = synth.rgb_batch("C", W, H, B)
    p = Pipeline(W, H, B, bench.P_C)
    feats = torch.zeros((B, 7), dtype=torch.float32, device="cuda")
    p.bind_features(feats)
    hist = torch.zeros(256, dtype=torch.int64, device="cuda")
    out = p.run(torch.from_numpy(rgb).cuda(), hist_sum=hist)
    labels = torch.arange(B, device="cuda", dtype=torch.int64) % C
    packed = PackedDatasetStats(C, torch.device("cuda"))
    packed.fill(hist, feats, labels)
    packed.all_reduce()  # no process group: stays as it is
    torch.cuda.synchronize()
    x0, x1 = O.xorwow_planes(W * H)
    image_ops = [o for o in bench.P_C if o["type"] != bench.OP_FEATURES]
    want_img, want_hist = O.pipeline(rgb, image_ops, x0, x1)
    assert np.array_equal(out.cpu().numpy(), want_img)
    assert np.array_equal(packed.histogram().cpu().numpy(), want_hist)
    want = np.stack([O.features(want_img[b]) for b in range(B)]).astype(np.float64)
    assert np.array_equal(feats.cpu().numpy().astype(np.float64), want)
    count, mean, std = (t.cpu().numpy() for t in packed.class_stats())
    lab = labels.cpu().numpy()
    for c in range(C):
        sel = want[lab == c]
        assert count[c] == len(sel)
        assert np.allclose(mean[c], sel.mean(axis=0), rtol=1e-12, atol=1e-9)
        assert np.allclose(std[c], sel.std(axis=0), rtol=1e-6, atol=1e-4)
    p.close()
