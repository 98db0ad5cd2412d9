"""Data parallelism for the SSD-VGG step: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The reference trains on a single device (no collective anywhere).  Images are
independent units, the loss is reduce_mean over the batch of per-sample normalised losses
(ssdvgg.py:520,559), so with equal shards the global gradient is the MEAN of the per-rank
gradients: all-reduce (sum) of the flat gradient arena, then 1/world folded into the update.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def reserve_hw_queues():
    """A process that creates an RCCL communicator needs more hardware queues than ROCm's default four
    (GPU_MAX_HW_QUEUES): the executor keeps three streams busy (caller's, heads / side, weight gradients), RCCL brings its
    own, and a stream that has to SHARE a queue with another is time-multiplexed.  Measured on one MI355X, bf16 step,
    single-rank group (profiles/r05_g_dp_hw_queues.txt): no group 6.92 ms; group initialised -- even if never used -- 7.15-7.19 ms
    (+3.5 %); the same with 8 queues 6.95-6.99 ms.  The variable is read when the HIP runtime starts, so this must run before
    the first torch.cuda call (train.py and bench.py call it first thing; a value the user set wins).

    Returns True when the reservation can still take effect.  It cannot -- and says so on stderr -- when the HIP runtime of
    this process is already up (torch.cuda initialised, or a handle / stream created through the C ABI: the variable was read
    then) or when the user pinned fewer than 8 queues: the data-parallel step then runs on shared queues, the measured
    +3.5 ... 4.7 % path.  (Set GPU_MAX_HW_QUEUES=8 in the environment of a C consumer: INTEGRATION.md section 5.)  A process
    that never creates a communicator should NOT set it: with one queue per stream the three-stream executor measured the
    same, but a five-stream one 25 % slower (csrc/net.hip, constructor)."""
    import sys
    have = os.environ.get('GPU_MAX_HW_QUEUES')
    late = torch.cuda.is_initialized()
    if have is None and not late:
        os.environ['GPU_MAX_HW_QUEUES'] = '8'
        return True
    ok = True
    try:
        ok = int(have) >= 8 if have is not None else False
    except ValueError:
        ok = False
    if late and have is None:
        print('[ssd_tensorflow_amd.parallel] WARNING: the HIP runtime was initialised before reserve_hw_queues(): '
              'GPU_MAX_HW_QUEUES cannot be raised to 8 any more; the data-parallel step will share hardware queues with '
              'RCCL (+3.5 ... 4.7 % per step, DESIGN.md section 6).  Call parallel.init() / reserve_hw_queues() before the first '
              'torch.cuda call, or export GPU_MAX_HW_QUEUES=8.', file=sys.stderr, flush=True)
    elif not ok:
        print(f'[ssd_tensorflow_amd.parallel] WARNING: GPU_MAX_HW_QUEUES={have!r} (< 8): executor and RCCL streams will share '
              'hardware queues (+3.5 ... 4.7 % per data-parallel step, DESIGN.md section 6).', file=sys.stderr, flush=True)
    return ok and not (late and have is None)


def env():
    """(rank, local_rank, world_size) from the launcher's environment.  SSD_FORCE_DEVICE pins the GPU ordinal of
    every rank (plumbing tests on a one-GPU box: all ranks on GPU 0, backend gloo)."""
    local = int(os.environ.get('SSD_FORCE_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    return int(os.environ.get('RANK', '0')), local, int(os.environ.get('WORLD_SIZE', '1'))


def init(backend=None):
    rank, local, world = env()
    if world > 1:
        reserve_hw_queues()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = os.environ.get('SSD_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        kw = {}
        if backend == 'nccl':
            torch.cuda.set_device(local)
            kw['device_id'] = torch.device('cuda', local)
        dist.init_process_group(backend, **kw)
    return rank, local, world


class ShardSampler:
    """The reference's single feeder shuffles the sample list and cuts it into batches
    (training_data.py:137-139,176-178).  Here every rank draws the SAME permutation (same seed)
    and takes its slice of each global batch: shards are disjoint, their union is the global batch,
    no exchange is needed.  Every rank yields the same number of batches (one per global batch), so
    the ranks issue the same collectives.  The last global batch may be short (the reference never
    pads a batch that reaches the net): it is split as evenly as possible, shards differ by at most
    one sample and a shard may be EMPTY when fewer samples than ranks are left -- such a rank still
    takes part in the step (train_step_dp: null gradients, the same all-reduce)."""

    def __init__(self, num_samples, batch_per_rank, rank=0, world=1, seed=0):
        self.n, self.b, self.rank, self.world, self.seed = int(num_samples), int(batch_per_rank), rank, world, seed

    def num_batches(self):
        g = self.b * self.world
        return (self.n + g - 1) // g

    def batches_with_count(self, epoch):
        """(this rank's sample indices, size of the global batch) for every global batch of the epoch."""
        perm = np.random.default_rng(self.seed + epoch).permutation(self.n)
        g = self.b * self.world
        for k in range(self.num_batches()):
            glob = perm[k * g:(k + 1) * g]
            q, r = divmod(len(glob), self.world)
            start = self.rank * q + min(self.rank, r)
            yield glob[start:start + q + (1 if self.rank < r else 0)], len(glob)

    def batches(self, epoch):
        for idx, _ in self.batches_with_count(epoch):
            yield idx


def allreduce_flat(flat, world, bucket_floats=0):
    """Sum `flat` (the gradient arena, a 1-D tensor) over ranks, in place.  bucket_floats > 0
    splits it into contiguous buckets issued back to back (async) so the first ones overlap the
    tail of the producer stream; the arena is laid out filters-first in forward order."""
    if world <= 1:
        return
    if bucket_floats <= 0 or bucket_floats >= flat.numel():
        dist.all_reduce(flat)
        return
    works = [dist.all_reduce(flat[o:o + bucket_floats], async_op=True) for o in range(0, flat.numel(), bucket_floats)]
    for w in works:
        w.wait()


def mean_scalars(values, world, device=None):
    """Average a short list of floats over ranks (loss logging)."""
    if world <= 1:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t)
    return (t / world).tolist()


class Bf16Message:
    """All-reduce of a gradient range through a bf16 message of half the bytes (allreduce_dtype='bf16').

    The range is packed (round to nearest even) into a bf16 buffer, the buffer is summed over the ranks, the sum is unpacked
    over the range; the update then scales by 1 / world as always.  Masters, momentum and every rank's own arena stay fp32: what
    is rounded is each rank's contribution (2^-9 relative) and the running sums inside the collective (bf16 adds).  Bound
    checked on CPU by tests/test_parallel_cpu.py: the averaged gradient differs from the fp32 all-reduce by < 1 % of its norm
    at world size 2, and the replicas still agree bit for bit (every rank unpacks the same reduced buffer).
    Packing / unpacking are HIP kernels (ssd_grads_to_bf16 / ssd_grads_from_bf16) on the stream the collective is ordered behind;
    CPU tensors (the gloo plumbing tests) take torch's conversion, which rounds the same way."""

    def __init__(self):
        self.buffers = {}

    def _buf(self, flat, off, cnt):
        key = (flat.data_ptr(), off, cnt)
        b = self.buffers.get(key)
        if b is None:
            b = self.buffers[key] = torch.empty(cnt, dtype=torch.bfloat16, device=flat.device)
        return b

    def all_reduce(self, flat, off, cnt, async_op=True):
        """-> a callable that completes the reduction of flat[off:off + cnt] (wait + unpack), to be called in issue order"""
        rng = flat[off:off + cnt]
        msg = self._buf(flat, off, cnt)
        if flat.is_cuda:
            from ._lib import lib, check
            stream = torch.cuda.current_stream(flat.device).cuda_stream
            check(lib.ssd_grads_to_bf16(flat.device.index, rng.data_ptr(), msg.data_ptr(), cnt, stream))
        else:
            msg.copy_(rng)
        work = dist.all_reduce(msg, async_op=async_op)

        def finish():
            if work is not None:
                work.wait()
            if flat.is_cuda:
                from ._lib import lib, check
                stream = torch.cuda.current_stream(flat.device).cuda_stream
                check(lib.ssd_grads_from_bf16(flat.device.index, msg.data_ptr(), rng.data_ptr(), cnt, stream))
            else:
                rng.copy_(msg)
        return finish


def _bf16_messages(net):
    """The handle's own message buffers (round 4 kept ONE process-wide cache keyed by arena address: every handle ever created
    pinned 2 bytes per filter float for the life of the process, and a recycled address would have found a dead handle's buffer).
    SSDVGG.close() drops them with the handle."""
    m = getattr(net, '_bf16_messages', None)
    if m is None:
        m = net._bf16_messages = Bf16Message()
    return m


def train_step_dp(net, x_dev, y_dev, world, bucket_floats=0, global_count=None, force_collectives=False, allreduce_dtype='f32'):
    """One data-parallel step on this rank's shard (device tensors).

    allreduce_dtype 'bf16': the FILTER gradients (all but 0.03 % of the arena) cross the links as bf16 messages (Bf16Message);
    the bias / scale tail is reduced in fp32.

    bucket_floats > 0: backward is driven in stages and every finished range of the filter
    gradients (>= bucket_floats, completed from the end of the arena: heads, conv11 ... conv1) is
    all-reduced asynchronously while the remaining backward kernels run; the tiny bias / scale
    tail goes last.  bucket_floats == 0: one all-reduce after backward.

    global_count: samples in the global batch when the shards may be unequal (the short last batch
    of an epoch).  Every rank then normalises its losses by global_count / world, so the plain mean
    over ranks (sum, then 1/world in the update) is the global-batch mean; a rank whose shard is empty
    (x_dev is None or has no rows) contributes the shared weight-decay gradient only.
    force_collectives: take the collective path even for world == 1 (a single-rank group: hardware test of the
    stream ordering)."""
    b = 0 if x_dev is None else int(x_dev.shape[0])
    if b == 0 and global_count is not None and global_count >= world:
        # a shard is only ever empty in a global batch with fewer samples than ranks: anything else means this rank's feeder
        # disagrees with the others'.  The step still runs (identical collectives on every rank: no hang), but loudly.
        import warnings
        warnings.warn('rank got an EMPTY shard of a global batch of %d samples on %d ranks: the feeders disagree' % (global_count, world),
                      RuntimeWarning, stacklevel=2)
    if world <= 1 and not force_collectives:
        if b == 0:
            return
        net.train_step_dev(x_dev, y_dev)        # forward + backward + update, the optimizer overlapped with backward's tail
        return
    if global_count is not None:
        net.set_loss_normalizer(global_count / world)
    # Every rank must issue the same collectives whatever its shard holds.  The sequence is a function of
    # bucket_floats alone: the ranges backward reports in stages are fixed by the layer sizes (ssd_backward_ranges), so
    # a rank with an empty shard (a short last batch with fewer samples than ranks -- or a feeder that disagrees with
    # the other ranks', which then shows up as wrong losses instead of a hang) reduces the same ranges of its
    # weight-decay-only gradient arena.
    try:
        if allreduce_dtype not in ('f32', 'bf16'):
            raise ValueError("allreduce_dtype must be 'f32' or 'bf16', got %r" % (allreduce_dtype,))
        bf16_msg = allreduce_dtype == 'bf16'
        if bucket_floats <= 0:
            if b == 0:
                net.null_gradients_dev()
            else:
                net.forward_backward_dev(x_dev, y_dev)
            if bf16_msg:
                nf = net.filter_floats
                fin = _bf16_messages(net).all_reduce(net.grads_flat, 0, nf, async_op=False)
                fin()
                dist.all_reduce(net.grads_flat[nf:])
            else:
                dist.all_reduce(net.grads_flat)
        else:
            # Collectives are enqueued behind the weight-gradient stream only: the data gradients on the
            # current stream keep running ahead of them.  The last stage joins the two streams, after which
            # the bias / scale tail is reduced and the current stream waits for everything.
            side = net.use_torch_wgrad_stream()
            if b == 0:
                net.null_gradients_dev()
                side.wait_stream(torch.cuda.current_stream(side.device))
                ranges = net.backward_ranges(bucket_floats)
            else:
                net.forward_dev(x_dev, y_dev)
                ranges = net.backward_staged(y_dev, b, bucket_floats, sync_main=False)
            works = []
            for off, cnt in ranges:
                with torch.cuda.stream(side):
                    if bf16_msg:      # packed behind the weight-gradient stream, unpacked (below) on the current stream
                        works.append(_bf16_messages(net).all_reduce(net.grads_flat, off, cnt))
                    else:
                        works.append(dist.all_reduce(net.grads_flat[off:off + cnt], async_op=True).wait)
            works.append(dist.all_reduce(net.grads_flat[net.filter_floats:], async_op=True).wait)
            for finish in works:
                finish()
    finally:
        if global_count is not None:
            net.set_loss_normalizer(0.0)
    net.apply_gradients_dev(1.0 / world)
