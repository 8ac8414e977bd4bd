"""The multi-GPU layer on CPU: two gloo ranks shard a batch image-wise, each reduces its shard
(the oracle stands in for the GPU here), and the all-reduced dataset histogram and per-class
statistics equal the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from imageanalysis_amd.dist import (PackedDatasetStats, accumulate_class_stats, allreduce_class_stats,
                                    allreduce_histogram, class_stat_buffers, shard_range)

W, H, B, C = 48, 20, 7, 3


def test_shard_range_covers_everything_once():
    for n in (0, 1, 7, 8, 500, 10000):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                seen += list(range(lo, hi))
            assert seen == list(range(n))
            sizes = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(10000, 7, 8) == (8750, 10000)
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dataset():
    import oracle_lib as O
    import synth
    rgb = synth.rgb_batch("C", W, H, B)
    labels = np.array([b % C for b in range(B)], np.int64)
    grays = [O.gray(rgb[b]) for b in range(B)]
    hists = np.stack([O.hist(g) for g in grays]).astype(np.int64)
    feats = np.stack([O.features(g) for g in grays])
    return hists, feats, labels


def _worker(rank, world, port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hists, feats, labels = _dataset()
    lo, hi = shard_range(B, rank, world)
    h = torch.from_numpy(hists[lo:hi].sum(axis=0)) if hi > lo else torch.zeros(256, dtype=torch.int64)
    allreduce_histogram(h)
    buf = class_stat_buffers(C, "cpu")
    if hi > lo:
        accumulate_class_stats(buf, torch.from_numpy(feats[lo:hi]), torch.from_numpy(labels[lo:hi]))
    count, mean, std = allreduce_class_stats(buf)
    # the same sums through the packed buffer: histogram and class statistics in ONE collective
    packed = PackedDatasetStats(C, "cpu")
    packed.fill(torch.from_numpy(hists[lo:hi].sum(axis=0)) if hi > lo else torch.zeros(256, dtype=torch.int64),
                torch.from_numpy(feats[lo:hi]), torch.from_numpy(labels[lo:hi]))
    packed.all_reduce_async().wait()
    pc, pm, ps = packed.class_stats()
    assert torch.equal(packed.histogram(), h) and torch.equal(pc, count)
    assert torch.allclose(pm, mean, rtol=1e-13, atol=0) and torch.allclose(ps, std, rtol=1e-9, atol=1e-9)
    if rank == 0:
        out_q.put((h.numpy(), count.numpy(), mean.numpy(), std.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_dataset_reduction_matches_single_process(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    h, count, mean, std = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    hists, feats, labels = _dataset()
    assert np.array_equal(h, hists.sum(axis=0))  # integers: exact at any rank count
    for c in range(C):
        sel = feats[labels == c].astype(np.float64)
        assert count[c] == len(sel)
        assert np.allclose(mean[c], sel.mean(axis=0), rtol=1e-12, atol=1e-9)
        assert np.allclose(std[c], sel.std(axis=0), rtol=1e-6, atol=1e-6)
