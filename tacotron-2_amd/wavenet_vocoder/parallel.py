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


def _mean_(t):
    """ONE collective per piece: RCCL averages in the reduction itself (ReduceOp.AVG, no second kernel on the communication
    stream); gloo has no AVG, so CPU tensors (the world-size-2 tests) take SUM and a scale."""
    d = torch.distributed
    if t.is_cuda:
        d.all_reduce(t, op=d.ReduceOp.AVG)
    else:
        d.all_reduce(t, op=d.ReduceOp.SUM)
        t.mul_(1.0 / world_size())
    return t


def allreduce_mean_(flat):
    """In-place mean over ranks of a flat tensor (tower-gradient mean, wavenet.py:564-575)."""
    if not is_distributed() or world_size() == 1:
        return flat
    return _mean_(flat)


def allreduce_loss_and_flags(loss, flags):
    """The step's two host-visible scalars in ONE small collective: the reported loss is the mean of the per-tower losses
    (wavenet.py:515-516) and every rank must learn whether ANY rank raised a flag (feeder failure, local abort).  SUM of
    [loss / world, flag_0, flag_1, ...]: element 0 comes back as the mean, the rest as the number of ranks that raised each flag."""
    vec = torch.cat([loss.reshape(1).float() / world_size(), flags.reshape(-1).float()])
    if is_distributed() and world_size() > 1:
        torch.distributed.all_reduce(vec, op=torch.distributed.ReduceOp.SUM)
    return vec


_COMM_STREAMS = {}


def _comm_stream(device):
    """One side stream per device for the bucketed all-reduce (RCCL enqueues behind whatever this stream waits for)."""
    key = (device.type, device.index)
    if key not in _COMM_STREAMS:
        _COMM_STREAMS[key] = torch.cuda.Stream(device=device)
    return _COMM_STREAMS[key]


class ExchangeTimer(object):
    """Optional instrumentation of allreduce_mean_buckets_ (bench.py under N > 1 ranks: the first multi-GPU run must explain itself).
    Per call and bucket: the span on the communication stream from "the bucket's ready event has fired" to "its all-reduce has ended", and
    the time the CALLER's stream stood at the final join (0 when the exchange had already finished under the backward).  GPU: device events
    (read after a synchronise, in ``summary``); CPU tensors (gloo tests): wall clock."""

    def __init__(self):
        self.calls = []         # per call: {'buckets': [(ready, done) ...], 'join': (before, after)}

    def _stamp(self, stream=None):
        if stream is None:
            import time
            return time.time()
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        return ev

    @staticmethod
    def _ms(a, b):
        return (b - a) * 1e3 if isinstance(a, float) else a.elapsed_time(b)

    def summary(self):
        """median over the recorded calls: per bucket [first float, floats, all-reduce span ms], the join wait, and exchange start -> end"""
        import statistics
        if not self.calls:
            return None
        if not isinstance(self.calls[0]['join'][0], float):
            torch.cuda.synchronize()
        nb = len(self.calls[0]['buckets'])
        med = lambda xs: float(statistics.median(xs))
        return {'calls': len(self.calls),
                'bucket_ready_to_allreduce_end_ms': [med([self._ms(*c['buckets'][i]) for c in self.calls]) for i in range(nb)],
                'first_bucket_ready_to_last_allreduce_end_ms': med([self._ms(c['buckets'][0][0], c['buckets'][-1][1]) for c in self.calls]),
                'caller_stream_wait_at_join_ms': med([self._ms(*c['join']) for c in self.calls])}


def allreduce_mean_buckets_(engine, flat, single_rank_ok=False, timer=None):
    """Tower-gradient mean (wavenet.py:564-575) overlapped with the backward pass.

    ``engine.train_bwd(flat)`` completes the flat gradient in ``engine.grad_buckets()`` contiguous pieces, top layers first, on
    ctx-owned streams.  Each piece is all-reduced (ReduceOp.AVG: one RCCL call, no scaling kernel) on a communication stream as soon as its event fires,
    i.e. while the weight gradients of the layers below are still being computed; the caller's stream is ordered after the last
    piece.  xGMI is point-to-point (7 x ~153 GB/s per GPU): the pieces stay large (4-6 of ~10-15 MB for the paper model), never
    one call per tensor.  Must be called right after ``engine.train_bwd(flat)`` on the same stream.  CPU tensors (gloo tests) take
    the same bucket walk without streams.  ``single_rank_ok`` runs the walk on a one-rank group too (GPU test of the stream /
    event ordering against RCCL without a second GPU).  ``timer`` (an ExchangeTimer): record this call's spans.
    """
    if not is_distributed() or (world_size() == 1 and not single_rank_ok):
        return flat
    buckets = engine.grad_buckets()
    covered = sum(n for _, n in buckets)
    rec = {'buckets': [], 'join': None} if timer is not None else None
    if not flat.is_cuda:
        for i, (off, n) in enumerate(buckets):
            engine.wait_bucket(i, None)
            t_ready = timer._stamp() if rec is not None else None
            _mean_(flat[off:off + n])
            if rec is not None:
                rec['buckets'].append((t_ready, timer._stamp()))
        if rec is not None:
            t = timer._stamp(); rec['join'] = (t, t)
    else:
        cur = torch.cuda.current_stream(flat.device)
        comm = _comm_stream(flat.device)
        for i, (off, n) in enumerate(buckets):
            engine.wait_bucket(i, comm)                     # comm stream waits for bucket i only (not for the rest of the backward)
            with torch.cuda.stream(comm):
                t_ready = timer._stamp(comm) if rec is not None else None
                _mean_(flat[off:off + n])
                if rec is not None:
                    rec['buckets'].append((t_ready, timer._stamp(comm)))
        t_before = timer._stamp(cur) if rec is not None else None
        cur.wait_stream(comm)
        if rec is not None:
            rec['join'] = (t_before, timer._stamp(cur))
    if rec is not None:
        timer.calls.append(rec)
    if covered != flat.numel():                             # alignment padding between tensors is inside the buckets; anything else is a bug
        raise RuntimeError('gradient buckets cover %d of %d floats' % (covered, flat.numel()))
    return flat


def shard_batch(items, rank_=None, world_=None):
    """Rank r takes the r-th contiguous slice of a global batch (== tf.split over towers, wavenet.py:233-239)."""
    r = rank() if rank_ is None else rank_
    w = world_size() if world_ is None else world_
    if len(items) % w != 0:
        raise ValueError('batch of %d is not divisible by %d ranks (feeder.py:267-268)' % (len(items), w))
    per = len(items) // w
    return items[r * per:(r + 1) * per]


def param_checksum(flat):
    """Exact (bit-pattern) checksum of a flat fp32 tensor: the int64 sum of its words read as int32.  Replicas that applied the same
    updates to the same all-reduced gradients have EQUAL checksums; any divergence of a single ulp changes it."""
    return flat.detach().contiguous().view(torch.int32).sum(dtype=torch.int64)


def assert_replicas_in_sync(flat, what='parameters'):
    """Replica-drift guard of data-parallel training: every rank owns a full replica that is never re-broadcast after step 0
    (clip / Adam / EMA are replicated, deterministic kernels on the all-reduced gradient), so a cheap all-reduce of a checksum at
    every checkpoint interval proves they are still bit-identical -- and aborts the job on every rank when they are not."""
    if not is_distributed() or world_size() == 1:
        return True
    cs = param_checksum(flat).reshape(1)
    both = torch.cat([cs, -cs])                             # MAX of (cs, -cs) = (max, -min): one collective
    torch.distributed.all_reduce(both, op=torch.distributed.ReduceOp.MAX)
    hi, lo = int(both[0].item()), -int(both[1].item())
    if lo != hi:
        raise RuntimeError('data-parallel replicas diverged: %s differ between ranks (checksum %d on rank %d, range [%d, %d])'
                           % (what, int(cs.item()), rank(), lo, hi))
    return True
