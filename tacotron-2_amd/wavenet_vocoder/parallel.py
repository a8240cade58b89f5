"""Data-parallel glue: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on ROCm).

The reference replicates towers in one TF graph and averages per-variable gradients with concat + reduce_mean
on one device (wavenet.py:553-581).  Here every rank holds a replica and the mean is ONE all-reduce over the
flat fp32 gradient buffer (54.7 MB for the paper shape): a single large collective suits xGMI's point-to-point
links (ring all-reduce is per-link bound, ~2*(N-1)/N * bytes / 153 GB/s  ~= 0.6 ms at N=8) far better than
~200 per-tensor calls.  Utterances are sharded across ranks by the feeder; no other data-path collective.
"""
import torch


def is_distributed():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def world_size():
    return torch.distributed.get_world_size() if is_distributed() else 1


def rank():
    return torch.distributed.get_rank() if is_distributed() else 0


def allreduce_mean_(flat):
    """In-place mean over ranks of a flat tensor (tower-gradient mean, wavenet.py:564-575)."""
    if not is_distributed() or world_size() == 1:
        return flat
    torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM)
    flat.mul_(1.0 / world_size())
    return flat


def shard_batch(items, rank_=None, world_=None):
    """Rank r takes the r-th contiguous slice of a global batch (== tf.split over towers, wavenet.py:233-239)."""
    r = rank() if rank_ is None else rank_
    w = world_size() if world_ is None else world_
    if len(items) % w != 0:
        raise ValueError('batch of %d is not divisible by %d ranks (feeder.py:267-268)' % (len(items), w))
    per = len(items) // w
    return items[r * per:(r + 1) * per]
