"""-m gpu: the data-parallel step on hardware as far as one GPU allows: a single-rank RCCL group (backend "nccl" on
ROCm) runs the staged backward with bucketed all-reduces enqueued behind the weight-gradient stream -- RCCL
initialisation, stream ordering and side-stream collectives execute for real -- and must reproduce the plain step bit
for bit.  The multi-rank arithmetic is covered on CPU (tests/test_parallel_cpu.py, gloo, world size 2)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from oracle import boxes as ob, ssdvgg_ref as ref
from ssd_tensorflow_amd import parallel
from ssd_tensorflow_amd.ssdvgg import SSDVGG, Session

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.fixture(scope='module')
def nccl_group():
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    yield
    dist.destroy_process_group()


def _net(b, w, dtype='f32'):
    sess = Session(0)
    net = SSDVGG(sess, 'vgg300')
    net.build_from_vgg(None, 20, max_batch=b, weights=w, dtype=dtype)
    net.build_optimizer(learning_rate=0.001, weight_decay=0.0005, momentum=0.9)
    net.set_stream(torch.cuda.current_stream().cuda_stream)
    return sess, net


@pytest.mark.parametrize('overlap', [True, False])
def test_bucketed_allreduce_over_rccl_equals_plain_step(nccl_group, overlap):
    b = 2
    preset = ob.get_preset('vgg300')
    w = ref.init_params(preset, 20, seed=3, alive=True)
    rng = np.random.default_rng(17)
    x, y, _ = ref.synth_batch(rng, b, preset)
    xt = torch.from_numpy(x).cuda(); yt = torch.from_numpy(y).cuda()
    sess1, plain = _net(b, w)
    sess2, dp = _net(b, w)
    if not overlap:      # weight gradients on the main stream: the collectives behind the side stream must still see them
        from ssd_tensorflow_amd._lib import lib, check
        check(lib.ssd_set_overlap(dp._h, 0))
    for step in range(3):
        plain.train_step_dev(xt, yt)
        parallel.train_step_dp(dp, xt, yt, 1, bucket_floats=4_000_000, force_collectives=True)
    torch.cuda.synchronize()
    assert torch.equal(plain.grads_flat, dp.grads_flat), 'all-reduce over one rank is the identity'
    assert torch.equal(plain.params_flat, dp.params_flat) and torch.equal(plain.momentum_flat, dp.momentum_flat)
    assert plain.get_losses() == dp.get_losses() and dp.global_step == 3
    # single all-reduce after backward, and the unequal-shard normaliser at its neutral value
    parallel.train_step_dp(dp, xt, yt, 1, bucket_floats=0, force_collectives=True, global_count=b)
    plain.train_step_dev(xt, yt)
    torch.cuda.synchronize()
    assert torch.equal(plain.params_flat, dp.params_flat)
    sess1.close(); sess2.close()


def test_loss_normalizer_and_null_gradients(nccl_group):
    """Unequal shards: a rank that normalises by global_count / world = 3/2 instead of its own b = 2 scales its data
    gradient by 2/1.5; a rank with an empty shard contributes weight_decay * filters only."""
    b = 2
    preset = ob.get_preset('vgg300')
    w = ref.init_params(preset, 20, seed=4, alive=True)
    rng = np.random.default_rng(18)
    x, y, _ = ref.synth_batch(rng, b, preset)
    xt = torch.from_numpy(x).cuda(); yt = torch.from_numpy(y).cuda()
    sess, net = _net(b, w)
    net.forward_backward_dev(xt, yt)
    g_own = net.grads_flat.clone(); L_own = net.get_losses()
    net.set_loss_normalizer(1.5)
    net.forward_backward_dev(xt, yt)
    g_n = net.grads_flat.clone(); L_n = net.get_losses()
    net.set_loss_normalizer(0.0)
    nf = net.filter_floats
    wd_term = 0.0005 * net.params_flat
    wd_term[nf:] = 0
    want = (g_own - wd_term) * (2.0 / 1.5) + wd_term
    err = float((g_n - want).norm() / want.norm())
    assert err < 1e-6, err
    assert abs(L_n['confidence'] - L_own['confidence'] * 2 / 1.5) < 1e-5 * L_n['confidence'] and L_n['l2'] == L_own['l2']
    net.null_gradients_dev()
    torch.cuda.synchronize()
    assert torch.equal(net.grads_flat, wd_term)
    # an empty shard issues the collectives of the ranks that run backward: the same ranges, in the same order
    staged = list(net.backward_ranges(4_000_000))
    net.forward_dev(xt, yt)
    assert staged == list(net.backward_staged(yt, b, 4_000_000)) and len(staged) >= 4
    assert staged[0][0] + staged[0][1] == nf and staged[-1][0] == 0 and all(a[0] == c[0] + c[1] for a, c in zip(staged, staged[1:]))
    for bucket in (0, 4_000_000):
        p0 = net.params_flat.clone(); m0 = net.momentum_flat.clone(); step0 = net.global_step
        parallel.train_step_dp(net, None, None, 1, bucket_floats=bucket, force_collectives=True, global_count=0)
        torch.cuda.synchronize()
        assert net.global_step == step0 + 1 and net.get_losses_step(0) == dict(total=0.0, localization=0.0, confidence=0.0, l2=0.0)
        wd = 0.0005 * p0; wd[nf:] = 0
        assert torch.allclose(net.momentum_flat, 0.9 * m0 + wd, rtol=1e-6, atol=1e-12)
    sess.close()


def test_bf16_message_allreduce_on_one_rank(nccl_group):
    """allreduce_dtype='bf16' (parallel.Bf16Message; HIP pack / unpack kernels around the collective): on a single rank the
    all-reduce is the identity, so after the step the FILTER gradients are exactly their bf16 roundings (round to nearest even, as
    torch rounds), the bias / scale tail is untouched fp32 -- with and without buckets, for odd range sizes too."""
    from ssd_tensorflow_amd._lib import lib, check
    g = torch.randn(1_000_003, device='cuda') * 3.0
    msg = torch.empty(g.numel(), dtype=torch.bfloat16, device='cuda')
    back = torch.empty_like(g)
    s = torch.cuda.current_stream().cuda_stream
    check(lib.ssd_grads_to_bf16(0, g.data_ptr(), msg.data_ptr(), g.numel(), s))
    check(lib.ssd_grads_from_bf16(0, msg.data_ptr(), back.data_ptr(), g.numel(), s))
    torch.cuda.synchronize()
    assert torch.equal(msg, g.bfloat16()) and torch.equal(back, g.bfloat16().float())
    b = 2
    preset = ob.get_preset('vgg300')
    w = ref.init_params(preset, 20, seed=5, alive=True)
    rng = np.random.default_rng(19)
    x, y, _ = ref.synth_batch(rng, b, preset)
    xt = torch.from_numpy(x).cuda(); yt = torch.from_numpy(y).cuda()
    sess1, plain = _net(b, w)
    plain.forward_backward_dev(xt, yt)
    torch.cuda.synchronize()
    want = plain.grads_flat.clone()
    nf = plain.filter_floats
    want[:nf] = want[:nf].bfloat16().float()
    for bucket in (0, 4_000_000):
        sess2, dp = _net(b, w)
        p0 = dp.params_flat.clone()
        parallel.train_step_dp(dp, xt, yt, 1, bucket_floats=bucket, force_collectives=True, allreduce_dtype='bf16')
        torch.cuda.synchronize()
        assert torch.equal(dp.grads_flat, want), bucket
        assert torch.equal(dp.momentum_flat, want)                                       # first step: momentum = gradient
        assert torch.allclose(dp.params_flat, p0 - 0.001 * want, rtol=1e-6, atol=1e-9)   # (the kernel's fused multiply-add rounds once)
        sess2.close()
    sess1.close()
