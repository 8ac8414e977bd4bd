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


class PackedDatasetStats:
    """The dataset-level sums of one batch in ONE buffer, so that one collective carries them
    all: [ 256 histogram bins | C counts | C x 7 feature sums | C x 7 sums of squares ] as
    float64.  Every integer in it stays far below 2^53 (10,000 images of 768x568 are 4.4e9
    pixels), so the integer parts are summed exactly whatever the number of ranks."""

    @staticmethod
    def length(n_classes):
        return 256 + int(n_classes) * (1 + 2 * N_FEATURES)

    def __init__(self, n_classes, device, buf=None):
        """buf: a float64 array of length(n_classes) to use instead of an own one -- e.g. a row of a
        2-D array, so that one collective over several rows carries several batches' sums."""
        self.C = int(n_classes)
        if buf is None:
            buf = torch.zeros(self.length(n_classes), dtype=torch.float64, device=device)
        assert buf.dtype == torch.float64 and buf.is_contiguous() and buf.numel() == self.length(n_classes)
        self.buf = buf
        c, f = self.C, N_FEATURES
        self.hist = self.buf[:256]
        self.count = self.buf[256:256 + c]
        self.sum = self.buf[256 + c:256 + c + c * f].view(c, f)
        self.sumsq = self.buf[256 + c + c * f:].view(c, f)

    def fill(self, hist_int64, features, labels):
        """This batch: the histogram the executor accumulated (int64[256]), the feature rows
        (float32 [n,7], as IAK_OP_FEATURES wrote them) and their class labels (int64 [n])."""
        if self.buf.is_cuda:
            # one kernel of the library (iak_dataset_sums), on the current stream
            from . import lib as _l
            lab = labels if labels.dtype == torch.int32 else labels.to(torch.int32)
            assert features.dtype == torch.float32 and features.is_contiguous() and lab.is_contiguous()
            assert hist_int64.dtype == torch.int64 and hist_int64.is_contiguous()
            _l.check(_l.load().iak_dataset_sums(hist_int64.data_ptr(), features.data_ptr(), lab.data_ptr(),
                                                int(features.shape[0]), self.C, self.buf.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream), "iak_dataset_sums")
            return
        self.buf.zero_()
        self.hist.copy_(hist_int64)
        f = features.to(torch.float64)
        self.count.index_add_(0, labels, torch.ones(labels.shape, dtype=torch.float64, device=labels.device))
        self.sum.index_add_(0, labels, f)
        self.sumsq.index_add_(0, labels, f * f)

    def all_reduce_async(self):
        return dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, async_op=True)

    def all_reduce(self):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM)
        return self

    def histogram(self):
        """int64[256], exact."""
        return self.hist.round().to(torch.int64)

    def class_stats(self):
        """(count int64[C], mean float64[C,7], std float64[C,7]) of the reduced sums."""
        n = self.count.clamp(min=1).unsqueeze(1)
        mean = self.sum / n
        var = (self.sumsq / n - mean * mean).clamp(min=0)
        return self.count.round().to(torch.int64), mean, var.sqrt()
