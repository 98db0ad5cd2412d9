"""CPU, world_size 2 over gloo: the N > 1 path's host logic -- shard sampler, the gradient
all-reduce (bucketed and plain) and the 1/world averaging -- with the CPU oracle standing in for
the per-rank GPU step.  Property checked: mean of per-rank gradients on disjoint shards == the
gradient of the global batch (loss is reduce_mean over per-sample normalised losses,
ssdvgg.py:520,559; the weight-decay term is identical on every rank)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ssd_tensorflow_amd import parallel


def test_shard_sampler_partitions_every_global_batch():
    n, b, world = 37, 4, 3
    seen = []
    for epoch in (0, 1):
        per_rank = [list(parallel.ShardSampler(n, b, r, world, seed=5).batches(epoch)) for r in range(world)]
        assert len({len(p) for p in per_rank}) == 1
        glob = []
        for k in range(len(per_rank[0])):
            shards = [per_rank[r][k] for r in range(world)]
            flat = np.concatenate(shards)
            assert len(set(flat.tolist())) == len(flat), 'shards of one global batch must be disjoint'
            assert all(len(s) <= b for s in shards)
            glob.extend(flat.tolist())
        assert sorted(glob) == list(range(n)), 'every sample exactly once per epoch'
        seen.append(glob)
    assert seen[0] != seen[1], 'reshuffled every epoch'
    # single process degenerates to the reference's feeder: ceil(n / b) batches
    s = parallel.ShardSampler(n, b)
    assert s.num_batches() == 10 and sum(len(i) for i in s.batches(0)) == n


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(3)
    r, l, w = parallel.init('gloo')
    assert (r, w) == (rank, world)
    from oracle import boxes as ob, ssdvgg_ref as ref
    preset = ob.get_preset('vgg300')
    params = ref.init_params(preset, 20, seed=3, alive=True)
    rng = np.random.default_rng(11)
    x, y, _ = ref.synth_batch(rng, world, preset)              # the same global batch on every rank
    m = ref.RefModel('vgg300', params=params)
    names = [k for k in m.params]
    # this rank's shard via the sampler (one global batch of `world` samples, 1 per rank)
    idx = list(parallel.ShardSampler(world, 1, rank, world, seed=0).batches(0))[0]
    _, L, g = m.grads(x[idx], y[idx])
    flat = torch.cat([torch.from_numpy(g[k]).reshape(-1) for k in names])
    flat2 = flat.clone()
    parallel.allreduce_flat(flat, world)                       # one buffer
    parallel.allreduce_flat(flat2, world, bucket_floats=5_000_000)   # bucketed, async
    assert torch.equal(flat, flat2)
    mean_loss = parallel.mean_scalars([L['total'], L['confidence']], world)
    if rank == 0:
        _, Lg, gg = m.grads(x, y)                              # the global batch in one process
        want = torch.cat([torch.from_numpy(gg[k]).reshape(-1) for k in names])
        got = flat / world
        err = float((got - want).norm() / want.norm())
        out.put(dict(err=err, loss=(mean_loss[0], Lg['total']), conf=(mean_loss[1], Lg['confidence']), idx=int(idx[0])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_averaging_gloo():
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    assert res['err'] < 1e-5, res
    assert abs(res['loss'][0] - res['loss'][1]) < 1e-4 * abs(res['loss'][1])
    assert abs(res['conf'][0] - res['conf'][1]) < 1e-4 * abs(res['conf'][1])
