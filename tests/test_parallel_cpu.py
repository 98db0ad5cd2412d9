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


def test_shard_sampler_short_last_batch_is_split_evenly_and_every_rank_steps():
    """ranks must take the same number of steps (same collectives); a short last global batch is split as
    evenly as possible, an empty shard is yielded (not skipped) when fewer samples than ranks are left."""
    n, b, world = 70, 8, 8                      # 64 + 6: the last global batch leaves ranks 6, 7 empty
    per_rank = [list(parallel.ShardSampler(n, b, r, world, seed=1).batches_with_count(0)) for r in range(world)]
    assert {len(p) for p in per_rank} == {2}
    last = [per_rank[r][1] for r in range(world)]
    assert [len(i) for i, _ in last] == [1, 1, 1, 1, 1, 1, 0, 0] and {c for _, c in last} == {6}
    assert sorted(np.concatenate([i for i, _ in last]).tolist() + np.concatenate([per_rank[r][0][0] for r in range(world)]).tolist()) == list(range(n))
    n = 37                                      # 3 ranks x 4: last global batch of 1... 37 = 3*12 + 1
    last = [list(parallel.ShardSampler(n, 4, r, 3, seed=2).batches_with_count(0))[-1] for r in range(3)]
    assert [len(i) for i, _ in last] == [1, 0, 0] and all(c == 1 for _, c in last)
    n = 29                                      # 24 + 5 over 3 ranks: 2, 2, 1
    last = [list(parallel.ShardSampler(n, 4, r, 3, seed=2).batches_with_count(0))[-1] for r in range(3)]
    assert [len(i) for i, _ in last] == [2, 2, 1]


def test_loss_summary_is_sample_weighted(tmp_path):
    """utils.py:236-283: sum(loss_b * n_b) / num_samples, reset after the push; tags as in the reference."""
    import json
    from ssd_tensorflow_amd.summaries import SummaryWriter, LossSummary, PrecisionSummary
    w = SummaryWriter(str(tmp_path))
    ls = LossSummary(w, 'training', 10)
    ls.add(dict(total=4.0, localization=1.0, confidence=2.0, l2=1.0), 8)
    ls.add(dict(total=2.0, localization=0.5, confidence=1.0, l2=0.5), 2)
    means = ls.push(3)
    assert means == dict(total=3.6, localization=0.9, confidence=1.8, l2=0.9) and ls.loss_values['total'] == 0.0
    doubled = LossSummary(None, 'validation', 4)
    doubled.add(dict(total=1.0, localization=1.0, confidence=1.0, l2=1.0), 2)
    assert doubled.push(1, reduce=lambda v: [2 * x for x in v])['total'] == 1.0        # two ranks, 2 of 4 samples each
    ps = PrecisionSummary(w, 'training', ['cat', 'dog'])
    ps.push(3, 0.5, {'cat': 0.25, 'dog': 0.75})
    ps.push(4, 0.0, {})                                                                 # nothing computed: nothing written
    w.close()
    rows = [json.loads(l) for l in open(tmp_path / 'scalars.jsonl')]
    assert [r['tag'] for r in rows] == ['training_total_loss', 'training_localization_loss', 'training_confidence_loss', 'training_l2_loss',
                                        'training_mAP', 'training_AP_cat', 'training_AP_dog']
    assert rows[0] == {'tag': 'training_total_loss', 'value': 3.6, 'step': 3} and rows[-1]['value'] == 0.75


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
    flat_b = flat.clone()
    parallel.allreduce_flat(flat, world)                       # one buffer
    parallel.allreduce_flat(flat2, world, bucket_floats=5_000_000)   # bucketed, async
    assert torch.equal(flat, flat2)
    # ---- the bf16 MESSAGE option (parallel.Bf16Message): each rank's contribution rounded to bf16, summed in bf16, unpacked over
    # the fp32 arena.  Against the fp32 all-reduce of the same gradients: bounded error, identical replicas.
    msgs = parallel.Bf16Message()
    done = [msgs.all_reduce(flat_b, 0, 7_000_001), msgs.all_reduce(flat_b, 7_000_001, flat_b.numel() - 7_000_001)]      # two odd-sized ranges
    for fin in done:
        fin()
    bf16_err = float((flat_b - flat).norm() / flat.norm())
    bf16_max = float((flat_b - flat).abs().max() / flat.abs().max())
    sums = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(sums, flat_b.double().sum().reshape(1))
    mean_loss = parallel.mean_scalars([L['total'], L['confidence']], world)
    if rank == 0:
        out.put(dict(bf16_err=bf16_err, bf16_max=bf16_max, bf16_replicas=[float(v) for v in sums]))
        _, Lg, gg = m.grads(x, y)                              # the global batch in one process
        want = torch.cat([torch.from_numpy(gg[k]).reshape(-1) for k in names])
        got = flat / world
        err = float((got - want).norm() / want.norm())
        out.put(dict(err=err, loss=(mean_loss[0], Lg['total']), conf=(mean_loss[1], Lg['confidence']), idx=int(idx[0])))
    # ---- unequal shards: a global batch of 3 over 2 ranks (2 + 1).  Each rank normalises its data term by
    # global_count / world = 1.5 instead of its own shard size (ssd_set_loss_normalizer); the plain mean over ranks
    # is then the gradient of the 3-sample batch.  The oracle's gradient (mean over the shard + wd * w) is rescaled
    # the way the library's normaliser does.
    x3, y3 = np.concatenate([x, x[:1]]), np.concatenate([y, y[:1]])
    x3[2] = x3[2][::-1].copy()                                  # a third, different image
    shard, count = list(parallel.ShardSampler(3, 2, rank, world, seed=0).batches_with_count(0))[0]
    assert count == 3 and len(shard) == (2 if rank == 0 else 1)
    _, Ls, gs = m.grads(x3[shard], y3[shard])
    wd = m.weight_decay
    scaled = []
    for k in names:
        decay = wd * m.params[k].detach().numpy() if k.endswith('/filter') else 0.0
        scaled.append(torch.from_numpy(((gs[k] - decay) * (len(shard) / (count / world)) + decay).astype(np.float32)).reshape(-1))
    flat3 = torch.cat(scaled)
    parallel.allreduce_flat(flat3, world)
    if rank == 0:
        _, _, g3 = m.grads(x3, y3)
        want3 = torch.cat([torch.from_numpy(g3[k]).reshape(-1) for k in names])
        out.put(dict(err3=float((flat3 / world - want3).norm() / want3.norm())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_averaging_gloo():
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    resb = out.get(timeout=600)
    res = out.get(timeout=600)
    res3 = out.get(timeout=900)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    # bf16 message: 2^-9 relative rounding per contribution and per add -> measured 2.5e-3 of the gradient's norm at world size 2
    print('    bf16 message all-reduce vs fp32:', resb)
    assert resb['bf16_err'] < 1e-2 and resb['bf16_max'] < 1e-2, resb
    assert resb['bf16_replicas'][0] == resb['bf16_replicas'][1], 'every rank must unpack the same reduced message'
    assert res3['err3'] < 1e-5, res3
    assert res['err'] < 1e-5, res
    assert abs(res['loss'][0] - res['loss'][1]) < 1e-4 * abs(res['loss'][1])
    assert abs(res['conf'][0] - res['conf'][1]) < 1e-4 * abs(res['conf'][1])
