"""-m gpu: round 6, the latency-bound tail of the graph as one launch per direction (csrc/tail_bf16.hip; bf16 configuration).

Reference layers: ssdvgg.py:300-332 (conv9_1 ... conv11_2, vgg512: ... conv12_2 incl. the bottom/right tf.pad) and the classifiers of
the maps they produce (ssdvgg.py:353-365).  Three things are checked:
  * every stage SHAPE of both presets, one stage at a time through the chain kernel's op-level entry points, against the oracle
    with the tolerances of tests/test_gpu_bf16.py (1e-3 for fp32-stored results, one bf16 rounding for bf16-stored ones);
  * the whole step with the chain (default) against the same handle configuration with SSD_TAIL_FUSE=0 (the per-layer launches):
    another summation order of the same bf16 products, so activations agree to a bf16 rounding and the fp32 results to 1e-3;
  * which launches the step makes (the profiler's labels): the chain's layers must not appear as launches of their own.
The layer-local oracle tests of the step (test_gpu_bf16.py, test_gpu_bench_config.py) run with the chain on: it is the default."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import boxes as ob
from oracle import ssdvgg_ref as ref
from gpu_util import lib, check, max_rel, rel_err
from test_gpu_bf16 import conv_case_check
from ssd_tensorflow_amd.ssdvgg import SSDVGG, Session

pytestmark = pytest.mark.gpu

# (name, b, hi, wi, ci, co, k, stride, dil, padding, relu, y_f32): every layer of the two presets' chains, plus ragged batches
TAIL_CASES = [
    ('conv9_1 1x1 512->128 on 10x10 (two 64-row tiles per image)', 3, 10, 10, 512, 128, 1, 1, 1, 'SAME', True, False),
    ('conv9_2 s2 10->5 asym', 2, 10, 10, 128, 256, 3, 2, 1, 'SAME', True, False),
    ('conv10_1 1x1 256->128 on 5x5', 4, 5, 5, 256, 128, 1, 1, 1, 'SAME', True, False),
    ('conv10_2 VALID 5->3', 2, 5, 5, 128, 256, 3, 1, 1, 'VALID', True, False),
    ('conv11_1 1x1 on 3x3', 5, 3, 3, 256, 128, 1, 1, 1, 'SAME', True, False),
    ('conv11_2 VALID 3->1', 3, 3, 3, 128, 256, 3, 1, 1, 'VALID', True, False),
    ('vgg512 conv10_2 s2 8->4', 2, 8, 8, 128, 256, 3, 2, 1, 'SAME', True, False),
    ('vgg512 conv11_2 VALID 4->2', 2, 4, 4, 128, 256, 3, 1, 1, 'VALID', True, False),
    ('vgg512 conv12_2 pad-BR 2->1', 2, 2, 2, 128, 256, 3, 1, 1, 'BR1', True, False),
    ('head 6 types N=152 on 5x5, f32 out', 2, 5, 5, 256, 152, 3, 1, 1, 'SAME', False, True),
    ('head 6 types N=152 on 8x8, f32 out', 1, 8, 8, 256, 152, 3, 1, 1, 'SAME', False, True),
    ('head 4 types N=104 on 3x3, f32 out', 3, 3, 3, 256, 104, 3, 1, 1, 'SAME', False, True),
    ('head 4 types N=104 on 1x1, f32 out', 7, 1, 1, 256, 104, 3, 1, 1, 'SAME', False, True),
    ('wide layer 320 outputs (two channel tiles), ragged', 1, 7, 6, 72, 320, 3, 1, 1, 'SAME', True, False),
]


@pytest.mark.parametrize('case', TAIL_CASES, ids=[c[0] for c in TAIL_CASES])
def test_tail_stage_shapes_against_the_oracle(case):
    conv_case_check(case, chain=True)


def _step(pname, b, fuse, monkeypatch, x, y, w):
    monkeypatch.setenv('SSD_TAIL_FUSE', fuse)
    sess = Session(0)
    net = SSDVGG(sess, pname)
    net.build_from_vgg(None, 20, max_batch=b, training=True, weights=w, dtype='bf16')
    net.build_optimizer(learning_rate=0.001, weight_decay=0.0005, momentum=0.9)
    out = {}
    xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    check(lib.ssd_profile_enable(net._h, 2))
    net.forward_backward_dev(xt, yt)
    torch.cuda.synchronize()
    buf = C.create_string_buffer(1 << 16)
    check(lib.ssd_profile_report(net._h, buf, len(buf)))
    check(lib.ssd_profile_enable(net._h, 0))
    out['kernels'] = [ln.split('\t')[0] for ln in buf.value.decode().strip().split('\n') if ln]
    out['losses'] = net.get_losses()
    out['result'] = net._dev_result(b, True)
    names = ['conv8_2', 'conv9_1', 'conv9_2', 'conv10_1', 'conv10_2', 'conv11_1', 'conv11_2'] + (['conv12_1', 'conv12_2'] if pname == 'vgg512' else [])
    for n in names:
        out[n] = net.activation(n, b)
        out['grad:' + n] = net.activation('grad:' + n, b)
    out['grads'] = net.save_gradients()
    sess.close()
    return out


@pytest.mark.parametrize('pname,b', [('vgg300', 3), ('vgg512', 2), ('vgg300', 32)])
def test_step_with_the_chain_agrees_with_the_per_layer_launches(pname, b, monkeypatch):
    preset = ob.get_preset(pname)
    w = ref.init_params(preset, 20, seed=42, alive=True)
    rng = np.random.default_rng(77)
    x, y, _ = ref.synth_batch(rng, b, preset)
    plain = _step(pname, b, '0', monkeypatch, x, y, w)
    fused = _step(pname, b, '3', monkeypatch, x, y, w)
    # which launches: the chain's layers (vgg300: conv9_1 on, vgg512: conv10_1 on) are launches of their own only without it
    first = 9 if pname == 'vgg300' else 10
    own = lambda ks: sorted({k for k in ks if k.split(':')[-1].startswith(tuple(f'conv{i}_' for i in range(first, 13)))})
    assert own(plain['kernels']) and 'tail_fwd_bf16:tail' not in plain['kernels']
    assert {'tail_fwd_bf16:tail', 'tail_dgrad_bf16:tail', 'conv_wgrad_group_bf16_64x128:tail'} <= set(fused['kernels'])
    left = own(fused['kernels'])
    assert all(k.split(':')[-1] == f'conv{first}_1' and 'fwd' not in k for k in left), left      # only the first layer's data and weight gradient are still launches
    assert np.array_equal(fused['conv8_2'], plain['conv8_2']), 'everything in front of the chain is the same launches'
    worst_a = worst_g = 0.0
    dead = []
    for k in plain:
        if k in ('losses', 'result', 'grads', 'kernels', 'conv8_2'):
            continue
        if np.count_nonzero(plain[k]) == 0:      # (a map none of whose anchors is a positive or a mined negative: nothing flows back)
            assert np.count_nonzero(fused[k]) == 0
            dead.append(k)
            continue
        e = max_rel(fused[k], plain[k])
        if k.startswith('grad:'):
            worst_g = max(worst_g, e)
        else:
            worst_a = max(worst_a, e)
    print(f'    chain vs per-layer launches: activations max-rel {worst_a:.2e}, data gradients {worst_g:.2e}; without gradient: {dead}')
    assert len(dead) <= 4 and 'grad:conv9_1' not in dead and 'grad:conv10_1' not in dead
    assert worst_a < 8e-3 and worst_g < 1.6e-2          # a bf16 rounding here and there that falls the other way (2^-8 of the value), carried through a layer or two
    assert max_rel(fused['result'], plain['result']) < 4e-3      # (softmax of head outputs computed from activations one bf16 rounding apart)
    for k in plain['losses']:
        assert abs(fused['losses'][k] - plain['losses'][k]) < 1e-3 * abs(plain['losses'][k]) + 1e-6
    tail_vars = [k for k in plain['grads'] if k.startswith(('conv9', 'conv10', 'conv11', 'conv12', 'classifiers/classifier3', 'classifiers/classifier4',
                                                            'classifiers/classifier5', 'classifiers/classifier6'))]
    assert tail_vars
    worst_w = max(rel_err(fused['grads'][k], plain['grads'][k]) for k in tail_vars)
    print(f'    worst weight-gradient rel-L2 of the chain\'s layers: {worst_w:.2e}')
    assert worst_w < 1e-2
