"""Image-wise sharding over the GPUs of one node and the one exchange the path has: the
dataset-level reduction of the 256-bin histogram and of per-class feature statistics.

The reference processes images strictly serially on one GPU (test/test.cu:266) and only
keeps dataset-level sums on the host: `totalHist` (test/test.cu:212,407-411) and the
per-image feature rows with their class labels (test/test.cu:541-550).  Here every rank
owns a contiguous shard of the images and the sums are all-reduced once per batch
(torch.distributed: RCCL on GPUs, gloo on CPU).  Counts are 64-bit integers: 10,000 images
of 768x568 exceed 2^31 pixels."""
import torch
import torch.distributed as dist

N_FEATURES = 7  # mean, std, area, perimeter, mode, rangeMin, rangeMax (imageMachineLearning.cu:409-415)


def shard_range(n_images, rank, world):
    """Contiguous, balanced shard [lo, hi) of n_images for `rank`: the first n % world ranks
    get one extra image."""
    if world <= 0 or not (0 <= rank < world) or n_images < 0:
        raise ValueError("bad shard arguments")
    base, extra = divmod(n_images, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allreduce_histogram(hist_local):
    """Sum an int64[256] histogram over all ranks, in place.  Integer addition is exactly
    associative, so the result does not depend on the number of ranks."""
    assert hist_local.dtype == torch.int64 and hist_local.numel() == 256
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(hist_local, op=dist.ReduceOp.SUM)
    return hist_local


def class_stat_buffers(n_classes, device):
    """count int64[C], sum float64[C,7], sumsq float64[C,7]."""
    return (torch.zeros(n_classes, dtype=torch.int64, device=device),
            torch.zeros((n_classes, N_FEATURES), dtype=torch.float64, device=device),
            torch.zeros((n_classes, N_FEATURES), dtype=torch.float64, device=device))


def accumulate_class_stats(buffers, features, labels):
    """features: float32/64 [n,7]; labels: int64 [n] in [0,C)."""
    count, s, ss = buffers
    f = features.to(torch.float64)
    count.index_add_(0, labels, torch.ones_like(labels, dtype=torch.int64))
    s.index_add_(0, labels, f)
    ss.index_add_(0, labels, f * f)
    return buffers


def allreduce_class_stats(buffers):
    """One all-reduce per tensor; returns (count, mean, std) per class after the reduction."""
    count, s, ss = buffers
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(count, op=dist.ReduceOp.SUM)
        packed = torch.cat([s.reshape(-1), ss.reshape(-1)])
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        s.copy_(packed[:s.numel()].view_as(s))
        ss.copy_(packed[s.numel():].view_as(ss))
    n = count.clamp(min=1).to(torch.float64).unsqueeze(1)
    mean = s / n
    var = (ss / n - mean * mean).clamp(min=0)
    return count, mean, var.sqrt()
