"""CPU: oracle/ssdvgg_ref.py self-consistency (its TF half is 'parity unpinned':
no reference test or golden vector exists).  Independent checks:
explicit numpy loss/gradient restatement vs autograd, TF padding arithmetic,
optimizer semantics."""
import numpy as np
import torch
import pytest
from oracle import boxes as ob
from oracle import ssdvgg_ref as ref


def test_same_padding_arithmetic():
    assert ref.same_pad(19, 3, 2) == (1, 1, 10)       # conv8_2 vgg300
    assert ref.same_pad(10, 3, 2) == (0, 1, 5)        # conv9_2: asymmetric
    assert ref.same_pad(75, 2, 2) == (0, 1, 38)       # pool3: ceil mode
    assert ref.same_pad(300, 2, 2) == (0, 0, 150)
    assert ref.same_pad(19, 3, 1, 6) == (6, 6, 19)    # mod_conv6 rate 6
    assert ref.same_pad(8, 3, 2) == (0, 1, 4)         # vgg512 conv10_2


def test_param_inventory():
    p3 = ref.param_shapes(ob.get_preset('vgg300')); p5 = ref.param_shapes(ob.get_preset('vgg512'))
    n3 = sum(int(np.prod(s)) for s in p3.values()); n5 = sum(int(np.prod(s)) for s in p5.values())
    assert n3 == 26285486 and n5 == 26959300
    assert 'classifiers/classifier1_5/filter' in p3 and p3['mod_conv6/filter'] == (3, 3, 512, 1024)


def test_piecewise_lr():
    v, b = [0.00075, 0.0001, 0.00001], [320000, 400000]
    assert ref.piecewise_lr(0, b, v) == 0.00075 and ref.piecewise_lr(320000, b, v) == 0.00075
    assert ref.piecewise_lr(320001, b, v) == 0.0001 and ref.piecewise_lr(400001, b, v) == 0.00001


def test_loss_numpy_vs_autograd():
    rng = np.random.default_rng(3)
    preset = ob.get_preset('vgg300')
    B, A = 3, 8732
    _, y, _ = ref.synth_batch(rng, B, preset)
    y[2] = 0; y[2, :, 20] = 1                            # a sample with no positives -> contributes 0
    out = rng.normal(0, 1.5, (B, A, 25)).astype(np.float32)
    t = torch.tensor(out, requires_grad=True)
    L = ref.losses(t, torch.tensor(y), {}, 20, 0.0)
    (L['confidence'] + L['localization']).backward()
    conf, loc, d_out, sel = ref.loss_numpy(out, y)
    assert abs(conf - float(L['confidence'])) < 1e-4 * abs(conf)
    assert abs(loc - float(L['localization'])) < 1e-4 * abs(loc)
    g = t.grad.numpy()
    assert np.abs(g - d_out).max() < 1e-5 * np.abs(d_out).max() + 1e-9
    assert np.all(g[2] == 0)
    # hard-negative count: k = min(neg_n, 3*pos_n)
    pos_n = (y[:, :, 20] == 0).sum(1)
    assert np.array_equal(sel.sum(1), np.minimum(A - pos_n, 3 * pos_n))


def test_forward_backward_vgg300_b1():
    torch.manual_seed(0)
    rng = np.random.default_rng(1234)
    preset = ob.get_preset('vgg300')
    m = ref.RefModel('vgg300', params=ref.init_params(preset, seed=42, bias_scale=0.01))
    x, y, _ = ref.synth_batch(rng, 1, preset)
    w0 = m.numpy_params()
    result, L, g = m.grads(x, y)
    assert result.shape == (1, 8732, 25)
    assert np.allclose(result[:, :, :21].sum(-1), 1, atol=1e-5)
    assert all(np.isfinite(v) for v in L.values())
    assert abs(L['total'] - (L['confidence'] + L['localization'] + L['l2'])) < 1e-5
    l2 = 0.0005 * sum(float((w.astype(np.float64) ** 2).sum() / 2) for k, w in w0.items() if k.endswith('/filter'))
    assert abs(L['l2'] - l2) < 1e-4 * l2
    assert all(np.isfinite(v).all() for v in g.values())
    assert np.abs(g['conv1_1/filter']).max() > 0 and np.abs(g['l2_norm_conv4_3/scale']).max() > 0
    m.set_optimizer([0.001], [], 0.9, 0.0005)
    m.train_step(x, y)
    w1 = m.numpy_params()
    k = 'conv4_2/filter'
    assert np.allclose(w1[k], w0[k] - 0.001 * g[k], rtol=0, atol=1e-7 + 1e-5 * np.abs(g[k]).max() * 0.001)
