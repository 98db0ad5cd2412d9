"""-m gpu: the bf16 configuration (BASELINE.json configs[2], SSD_DTYPE_BF16) through the C ABI.

What bf16 changes is storage and the multiplier inputs; products accumulate in fp32 and the loss,
the gradient arena and the optimizer stay fp32.  The oracle therefore applies the SAME roundings at
the same places (inputs and filters rounded to bf16, exact fp32 math in between: torch-CPU on
bf16-representable values) and the kernels must agree with it
  * to 1e-3 (north_star's float tolerance) wherever the result is stored in fp32 (weight and bias
    gradients, head outputs, losses), and
  * to one bf16 rounding of the stored value (2^-9 relative, bound 4e-3 of the tensor's scale)
    where the result is stored in bf16 (activations, data gradients); 8e-3 where a gradient tensor
    is accumulated from two consumers (two roundings).
The distance between the bf16 and the fp32 configurations is reported, not asserted tightly."""
import zlib
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import boxes as ob
from oracle import ssdvgg_ref as ref
from gpu_util import lib, check, dev, ptr, host, rel_err, max_rel, conv_geom
from test_gpu_kernels import oracle_conv
from test_gpu_model import layer_local_backward_check, head_out_from_buffers, nchw, report, WD
from ssd_tensorflow_amd.ssdvgg import SSDVGG, Session

pytestmark = pytest.mark.gpu
TOL = 1e-3          # fp32-stored results
TOL_BF = 4e-3       # bf16-stored results: one rounding (2^-9) of values up to the tensor's scale
TOL_BF2 = 8e-3      # bf16 gradient tensors accumulated from two consumers


def q(a):
    """round to bf16, back to fp32 (numpy in, numpy out)"""
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).bfloat16().float().numpy()


def qt(t):
    return t.bfloat16().float()


def bdev(a):
    """numpy fp32 (bf16-representable) -> bf16 device tensor"""
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to('cuda').bfloat16().contiguous()


def bhost(t):
    torch.cuda.synchronize()
    return t.float().cpu().numpy()


# (name, b, hi, wi, ci, co, k, stride, dil, padding, relu, y_f32)
CONV_CASES = [
    ('conv1_2-like 64->64', 2, 40, 33, 64, 64, 3, 1, 1, 'SAME', True, False),
    ('conv2_1-like 64->128', 1, 30, 30, 64, 128, 3, 1, 1, 'SAME', True, False),
    ('conv3-like 128->256', 1, 23, 23, 128, 256, 3, 1, 1, 'SAME', True, False),
    ('conv4-like 256->512', 1, 19, 19, 256, 512, 3, 1, 1, 'SAME', True, False),
    ('mod_conv6 dil6', 2, 19, 19, 512, 1024, 3, 1, 6, 'SAME', True, False),
    ('mod_conv7 1x1', 2, 19, 19, 1024, 1024, 1, 1, 1, 'SAME', True, False),
    ('conv8_1 1x1 ->256', 2, 19, 19, 1024, 256, 1, 1, 1, 'SAME', True, False),
    ('conv8_2 s2 19->10', 2, 19, 19, 256, 512, 3, 2, 1, 'SAME', True, False),
    ('conv9_2 s2 10->5 asym', 2, 10, 10, 128, 256, 3, 2, 1, 'SAME', True, False),
    ('conv10_2 VALID 5->3', 2, 5, 5, 128, 256, 3, 1, 1, 'VALID', True, False),
    ('conv11_2 VALID 3->1', 3, 3, 3, 128, 256, 3, 1, 1, 'VALID', True, False),
    ('conv12_2 pad-BR 2->1', 2, 2, 2, 128, 256, 3, 1, 1, 'BR1', True, False),
    ('head 6 types N=152 f32 out', 2, 10, 10, 512, 152, 3, 1, 1, 'SAME', False, True),
    ('head 4 types N=104 f32 out', 2, 38, 38, 512, 104, 3, 1, 1, 'SAME', False, True),
    ('ragged M, 1 image', 1, 13, 7, 128, 128, 3, 1, 1, 'SAME', True, False),
    ('wide rows 70x70 (incremental pixel walk)', 1, 70, 70, 64, 128, 3, 1, 1, 'SAME', True, False),
    # the 8-wave kernel-row weight gradient: several ragged pixel splits; channel tiles that are not whole (192 = 128 + 64, 136 = 128 + 8)
    ('rows8 wgrad 5 ragged splits 128->128', 2, 40, 33, 128, 128, 3, 1, 1, 'SAME', True, False),
    ('rows8 wgrad partial tiles 192->136', 1, 9, 11, 192, 136, 3, 1, 1, 'SAME', True, False),
]


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_bf16_fwd_dgrad_wgrad(case):
    conv_case_check(case, chain=False)


def conv_case_check(case, chain):
    """One layer's three passes against the oracle.  chain: through the forms the step uses for the latency-bound tail (round 6:
    one workgroup per image, csrc/tail_bf16.hip; the weight gradient with a single pixel split and the direct epilogue)."""
    name, b, hi, wi, ci, co, k, stride, dil, padding, relu, y_f32 = case
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    ph, pw, ho, wo = conv_geom(hi, wi, k, stride, dil, padding)
    x = q(rng.normal(0, 1, (b, hi, wi, ci)))
    w = (rng.normal(0, 1, (k, k, ci, co)) / np.sqrt(k * k * ci)).astype(np.float32)     # the fp32 master
    bias = rng.normal(0, 0.1, (co,)).astype(np.float32)
    nreal = {152: 150, 104: 100}.get(co, co)      # fused-head padding columns are zero in the product
    w[..., nreal:] = 0; bias[nreal:] = 0
    wq_ = q(w)
    dy = q(rng.normal(0, 1, (b, ho, wo, co)))

    xt, wt, bt, pre, y_ref = oracle_conv(x, wq_, bias, stride, dil, padding, relu)
    g = torch.tensor(dy).permute(0, 3, 1, 2)
    # the product works on pre-activation gradients: dy masked by relu, itself a bf16 tensor
    gpre = qt(g * (pre > 0).float()) if relu else g
    pre.backward(gpre)
    dx_ref = xt.grad.permute(0, 2, 3, 1).numpy()
    dw_ref = wt.grad.numpy()
    db_ref = bt.grad.numpy()
    dy_pre = gpre.permute(0, 2, 3, 1).contiguous().numpy()

    x_, w_, b_ = bdev(x), dev(w), dev(bias)
    wio_ = torch.empty((k * k, ci, co), dtype=torch.bfloat16, device='cuda')
    woi_ = torch.empty((k * k, co, ci), dtype=torch.bfloat16, device='cuda')
    check(lib.ssd_op_cast_filter(ptr(w_), ptr(wio_), ptr(woi_), k * k, ci, co, None))
    assert np.array_equal(bhost(wio_).reshape(w.shape), wq_), 'filter mirror [tap][Ci][Co] is not round-to-nearest-even'
    assert np.array_equal(bhost(woi_), np.transpose(wq_.reshape(k * k, ci, co), (0, 2, 1))), 'filter mirror [tap][Co][Ci]'

    geom = (b, hi, wi, ci, ho, wo, co, k, k, stride, dil, ph, pw)
    y_ = torch.full((b, ho, wo, co), 9.0, dtype=torch.float32 if y_f32 else torch.bfloat16, device='cuda')
    fwd = lib.ssd_op_conv2d_fwd_bf16_chain if chain else lib.ssd_op_conv2d_fwd_bf16
    dgrad = lib.ssd_op_conv2d_dgrad_bf16_chain if chain else lib.ssd_op_conv2d_dgrad_bf16
    check(fwd(ptr(x_), ptr(woi_), ptr(b_), ptr(y_), int(y_f32), *geom, int(relu), None))
    yr = y_ref.detach().permute(0, 2, 3, 1).numpy()
    e = max_rel(bhost(y_), yr)
    assert e < (TOL if y_f32 else TOL_BF), f'{name}: forward max-rel {e:.3e}'
    if not y_f32:      # and elementwise: the stored value is the oracle's, rounded once (1 ulp slack for summation order)
        got = bhost(y_)
        assert np.all(np.abs(got - yr) <= np.abs(yr) * 2.0 ** -7 + 1e-3 * np.abs(yr).max())

    wd = 0.0005
    nws = lib.ssd_op_conv2d_wgrad_bf16_ws_floats(*geom)
    ws_ = torch.empty((nws,), dtype=torch.float32, device='cuda')
    gdy_ = bdev(dy_pre)
    gw_ = torch.full((k, k, ci, co), 7.0, dtype=torch.float32, device='cuda')
    gb_ = torch.full((co,), 7.0, dtype=torch.float32, device='cuda')
    if chain:
        check(lib.ssd_op_conv2d_wgrad_bf16_direct(ptr(x_), ptr(gdy_), ptr(gw_), ptr(gb_), ptr(w_), wd, *geom, None))
    else:
        check(lib.ssd_op_conv2d_wgrad_bf16(ptr(x_), ptr(gdy_), ptr(gw_), ptr(gb_), ptr(w_), wd, ptr(ws_), *geom, None))
    e = max_rel(host(gw_), dw_ref + wd * w)
    assert e < TOL, f'{name}: wgrad max-rel {e:.3e}'
    e = max_rel(host(gb_), db_ref)
    assert e < TOL, f'{name}: bias-grad max-rel {e:.3e}'

    gx_ = torch.full((b, hi, wi, ci), 3.0, dtype=torch.bfloat16, device='cuda')
    check(dgrad(ptr(gdy_), ptr(wio_), ptr(gx_), None, 0, *geom, None))
    e = max_rel(bhost(gx_), dx_ref)
    assert e < TOL_BF, f'{name}: dgrad max-rel {e:.3e}'
    prev = q(rng.normal(0, 1, x.shape))
    gx_ = bdev(prev)
    check(dgrad(ptr(gdy_), ptr(wio_), ptr(gx_), ptr(x_), 1, *geom, None))
    expect = (dx_ref + prev) * (x > 0)
    e = max_rel(bhost(gx_), expect)
    assert e < TOL_BF, f'{name}: dgrad accumulate+mask max-rel {e:.3e}'


def test_conv_bf16_full_size_layer():
    """conv1_2 at the full vgg300 size for one image (M = 90,000 pixels): tile tails, XCD remap, split-M slabs."""
    rng = np.random.default_rng(5)
    b, hi, wi, ci, co = 1, 300, 300, 64, 64
    x = q(rng.normal(0, 1, (b, hi, wi, ci)))
    w = (rng.normal(0, 1, (3, 3, ci, co)) / 24).astype(np.float32)
    bias = rng.normal(0, 0.1, (co,)).astype(np.float32)
    dy = q(rng.normal(0, 1, (b, hi, wi, co)))
    xt, wt, bt, pre, y_ref = oracle_conv(x, q(w), bias, 1, 1, 'SAME', False)
    pre.backward(torch.tensor(dy).permute(0, 3, 1, 2))
    x_, w_, b_, dy_ = bdev(x), dev(w), dev(bias), bdev(dy)
    wio_ = torch.empty((9, ci, co), dtype=torch.bfloat16, device='cuda'); woi_ = torch.empty((9, co, ci), dtype=torch.bfloat16, device='cuda')
    check(lib.ssd_op_cast_filter(ptr(w_), ptr(wio_), ptr(woi_), 9, ci, co, None))
    geom = (b, hi, wi, ci, hi, wi, co, 3, 3, 1, 1, 1, 1)
    y_ = torch.empty((b, hi, wi, co), dtype=torch.bfloat16, device='cuda')
    check(lib.ssd_op_conv2d_fwd_bf16(ptr(x_), ptr(woi_), ptr(b_), ptr(y_), 0, *geom, 0, None))
    assert max_rel(bhost(y_), y_ref.detach().permute(0, 2, 3, 1).numpy()) < TOL_BF
    ws_ = torch.empty((lib.ssd_op_conv2d_wgrad_bf16_ws_floats(*geom),), dtype=torch.float32, device='cuda')
    gw_ = torch.empty((3, 3, ci, co), dtype=torch.float32, device='cuda'); gb_ = torch.empty((co,), dtype=torch.float32, device='cuda')
    check(lib.ssd_op_conv2d_wgrad_bf16(ptr(x_), ptr(dy_), ptr(gw_), ptr(gb_), ptr(w_), 0.0, ptr(ws_), *geom, None))
    assert max_rel(host(gw_), wt.grad.numpy()) < TOL and max_rel(host(gb_), bt.grad.numpy()) < TOL
    gx_ = torch.empty((b, hi, wi, ci), dtype=torch.bfloat16, device='cuda')
    check(lib.ssd_op_conv2d_dgrad_bf16(ptr(dy_), ptr(wio_), ptr(gx_), None, 0, *geom, None))
    assert max_rel(bhost(gx_), xt.grad.permute(0, 2, 3, 1).numpy()) < TOL_BF



@pytest.mark.parametrize('forced', [dict(SSD_GATHER_ROWS256_BF16='2', SSD_C64_BF16='2'), dict(SSD_GATHER_ROWS_N64_BF16='2'),
                                    dict(SSD_GATHER_ROWS_N64_BF16='0', SSD_SMALL_TILE='0'), dict(SSD_SMALL_KSPLIT='0')],
                         ids=['rows256+c64', 'rows128x64', 'no-n64-no-smalltile', 'deep-ring-instead-of-ksplit'])
def test_conv_bf16_large_layer_kernels_forced(forced):
    """Two kernels are picked only at batch-32 sizes: the 256-row kernel-row gather (conv2_2 / conv3_x, where its tiles
    fill the chip twice) and the persistent 64 -> 64 kernel with the resident filter (conv1_2, >= 4 tiles per CU).
    SSD_GATHER_ROWS256_BF16=2 / SSD_C64_BF16=2 select them for every eligible layer.  The library reads the switches once
    per process: the conv cases (and the full-size conv1_2 image) run again in a child process with them set."""
    import os, subprocess, sys
    if all(os.environ.get(k) == v for k, v in forced.items()):
        pytest.skip('already the forced configuration')
    # (round 4: the 128 x 64 kernel-row tiles of the 19x19 maps -- SSD_GATHER_ROWS_N64_BF16=2 takes them everywhere, =0 together
    # with SSD_SMALL_TILE=0 restores round 3's choices, which stay reachable through the A/B switches)
    env = dict(os.environ, **forced)
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-x', '-q', '-k',
                        'test_conv_bf16_fwd_dgrad_wgrad or test_conv_bf16_full_size_layer', '-p', 'no:cacheprovider'], env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


FIRST_CASES = [('conv1_1 37x41 b2', 2, 37, 41), ('conv1_1 300x300 b1', 1, 300, 300), ('conv1_1 5x3 b3', 3, 5, 3),
               ('conv1_1 64x64 b2 (whole tiles)', 2, 64, 64)]


@pytest.mark.parametrize('case', FIRST_CASES, ids=[c[0] for c in FIRST_CASES])
def test_first_layer_bf16(case):
    """the dedicated conv1_1 kernels: fp32 image / master filter in, bf16 activations, fp32 gradients"""
    name, b, hi, wi = case
    ci, co, k = 3, 64, 3
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    ph, pw, ho, wo = conv_geom(hi, wi, k, 1, 1, 'SAME')
    x = rng.integers(0, 256, (b, hi, wi, ci)).astype(np.float32)            # what the data pipeline feeds (training_data.py:100)
    w = (rng.normal(0, 1, (k, k, ci, co)) / 100).astype(np.float32)
    bias = rng.normal(0, 0.5, (co,)).astype(np.float32)
    dy = q(rng.normal(0, 1, (b, ho, wo, co)))
    xt, wt, bt, pre, y_ref = oracle_conv(x, q(w), bias, 1, 1, 'SAME', True)
    gpre = qt(torch.tensor(dy).permute(0, 3, 1, 2) * (pre > 0).float())
    pre.backward(gpre)
    x_, w_, b_ = dev(x), dev(w), dev(bias)
    geom = (b, hi, wi, ci, ho, wo, co, k, k, 1, 1, ph, pw)
    y_ = torch.full((b, ho, wo, co), 9.0, dtype=torch.bfloat16, device='cuda')
    check(lib.ssd_op_conv2d_first_fwd_bf16(ptr(x_), ptr(w_), ptr(b_), ptr(y_), *geom, 1, None))
    e = max_rel(bhost(y_), y_ref.detach().permute(0, 2, 3, 1).numpy())
    assert e < TOL_BF, f'{name}: forward max-rel {e:.3e}'
    gdy_ = bdev(gpre.permute(0, 2, 3, 1).contiguous().numpy())
    ws_ = torch.empty((lib.ssd_op_conv2d_first_wgrad_bf16_ws_floats(*geom),), dtype=torch.float32, device='cuda')
    gw_ = torch.full((k, k, ci, co), 7.0, dtype=torch.float32, device='cuda'); gb_ = torch.full((co,), 7.0, dtype=torch.float32, device='cuda')
    check(lib.ssd_op_conv2d_first_wgrad_bf16(ptr(x_), ptr(gdy_), ptr(gw_), ptr(gb_), ptr(w_), 0.0005, ptr(ws_), *geom, None))
    e = max_rel(host(gw_), wt.grad.numpy() + 0.0005 * w)
    assert e < TOL, f'{name}: wgrad max-rel {e:.3e}'
    e = max_rel(host(gb_), bt.grad.numpy())
    assert e < TOL, f'{name}: bias-grad max-rel {e:.3e}'


def test_first_layer_bf16_raster_form():
    """SSD_FIRST_ROWS_BF16=0: the pixel-raster conv1_1 forward (bias added behind the sum) that the row-aligned kernel replaced in
    round 6 stays reachable and under test (the same cases in a process of its own: the switch is read once)."""
    import os, subprocess, sys
    if os.environ.get('SSD_FIRST_ROWS_BF16') == '0':
        pytest.skip('already the forced configuration')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-x', '-q', '-k', 'test_first_layer_bf16 and not raster',
                        '-p', 'no:cacheprovider'], env=dict(os.environ, SSD_FIRST_ROWS_BF16='0'),
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_first_layer_bf16_many_tiles_per_wave():
    """SSD_FIRST_GRID=16: the row-aligned conv1_1 forward with 64 waves for everything -- dozens of tiles per wave, so the
    one-tile-ahead ring, its null tiles past the end and the counted waits all run (the default grid gives the small cases one tile
    per wave)."""
    import os, subprocess, sys
    if os.environ.get('SSD_FIRST_GRID') == '16':
        pytest.skip('already the forced configuration')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-x', '-q', '-k', 'test_first_layer_bf16 and not raster and not many',
                        '-p', 'no:cacheprovider'], env=dict(os.environ, SSD_FIRST_GRID='16'),
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def layer_local_forward_check(net, m, preset, b, x, only=None):
    """Every op's forward recomputed by the oracle from the GPU's own (bf16) input activation and the
    bf16-rounded filter; head outputs are fp32.  only: optional list of op names ('conv4_2', 'pool3', 'heads/map0',
    'l2_norm_conv4_3') for the large-batch tests, which cannot afford every layer on the CPU."""
    act = {'image_input': x}

    def A(name):
        if name not in act:
            act[name] = net.activation(name, b)
        return act[name]

    worst = 0.0
    for op in ref.graph(preset):
        if only is not None and ('heads/map%d' % op[1] if op[0] == 'head' else 'l2_norm_conv4_3' if op[0] == 'l2norm' else op[1]) not in only:
            continue
        a = nchw(A(op[2]))
        if op[0] == 'conv':
            _, name, _, k, stride, padding, dil = op
            w = qt(m.params[name + '/filter'].detach())
            a = qt(a)                                      # conv1_1 rounds the fp32 image too (0..255 integers are exact)
            xin = F.pad(a, (0, 1, 0, 1)) if padding == 'BR1' else a
            want = F.relu(ref.conv2d_tf(xin, w, stride, 'SAME' if padding == 'SAME' else 'VALID', dil)
                          + m.params[name + '/biases'].detach().view(1, -1, 1, 1))
            got = A(name)
            assert np.count_nonzero(got) > 0.2 * got.size, f'{name} is (nearly) dead: the test would prove nothing'
            worst = max(worst, report('forward ' + name, max_rel(got, want.permute(0, 2, 3, 1).numpy())))
        elif op[0] == 'pool':
            _, name, _, k, s = op
            want = ref.maxpool_tf(a, k, s)
            assert np.array_equal(A(name), want.permute(0, 2, 3, 1).numpy()), f'{name}: max pooling of bf16 values is exact'
        elif op[0] == 'l2norm':
            want = ref.l2norm_tf(a, m.params['l2_norm_conv4_3/scale'].detach())
            worst = max(worst, report('forward norm_conv4_3', max_rel(A('norm_conv4_3'), want.permute(0, 2, 3, 1).numpy())))
        elif op[0] == 'head':
            i = op[1]
            buf = A(f'head{i}')
            for j in range(2 + len(preset['maps'][i][2])):
                n = f'classifiers/classifier{i}_{j}'
                want = ref.conv2d_tf(a, qt(m.params[n + '/filter'].detach())) + m.params[n + '/biases'].detach().view(1, -1, 1, 1)
                e = max_rel(buf[..., j * 25:(j + 1) * 25], want.permute(0, 2, 3, 1).numpy())
                assert e < TOL, f'head {i}.{j}: fp32 output max-rel {e:.3e}'
    return worst


@pytest.mark.usefixtures('unfused_pools')
@pytest.mark.parametrize('pname,b', [('vgg300', 2), ('vgg512', 1)])
def test_bf16_step_layer_local(pname, b):
    preset = ob.get_preset(pname)
    w = ref.init_params(preset, 20, seed=42, alive=True)
    m = ref.RefModel(pname, params=w)
    sess = Session(0)
    net = SSDVGG(sess, pname)
    net.build_from_vgg(None, 20, max_batch=b, training=True, weights=w, dtype='bf16')
    assert net.dtype == 'bf16'
    rng = np.random.default_rng(1234)
    x, y, _ = ref.synth_batch(rng, b, preset)
    m.set_optimizer([0.001], [], 0.9, WD)
    net.build_optimizer(learning_rate=0.001, weight_decay=WD, momentum=0.9)

    # ---- forward, layer-local; result and losses from the GPU's own fp32 head outputs ---------
    r, L = sess.run([net.result, net.losses], feed_dict={net.image_input: x, net.labels: y})
    worst = layer_local_forward_check(net, m, preset, b, x)
    assert worst < TOL_BF
    out_gpu = head_out_from_buffers(net, preset, b)
    conf, loc, _, _ = ref.loss_numpy(out_gpu, y)
    assert abs(L['confidence'] - conf) < TOL * abs(conf) and abs(L['localization'] - loc) < TOL * abs(loc)
    sm = torch.softmax(torch.from_numpy(out_gpu[..., :21]), -1).numpy()
    assert max_rel(r[..., :21], sm) < TOL and np.array_equal(r[..., 21:], out_gpu[..., 21:])
    # distance to the fp32 configuration (reported; bf16 inputs move every activation by ~2^-9 per layer)
    r_ref, L_ref = m.eval_step(x, y)
    print('    bf16 vs fp32 oracle: losses', {k: (round(L[k], 4), round(float(L_ref[k]), 4)) for k in L},
          ' result max-rel', max_rel(r, r_ref))
    assert abs(L['l2'] - L_ref['l2']) < TOL * L_ref['l2']            # fp32 masters
    assert abs(L['total'] - L_ref['total']) < 0.05 * abs(L_ref['total'])

    # ---- backward, layer-local ------------------------------------------------------------------
    xt = torch.from_numpy(x).cuda(); yt = torch.from_numpy(y).cuda()
    net.forward_backward_dev(xt, yt)
    torch.cuda.synchronize()
    worst_w, worst_x = layer_local_backward_check(net, m, preset, b, x, y, wq=qt, tol_dout=TOL_BF)
    print('    worst layer-local weight-gradient error', worst_w, ' data-gradient error', worst_x)
    assert worst_w < TOL and worst_x < TOL_BF2

    # ---- the update is the fp32 one ---------------------------------------------------------------
    g = net.save_gradients(); w0 = net.save_variables()
    net.apply_gradients_dev(1.0)
    w1 = net.save_variables()
    for k in ('conv4_2/filter', 'classifiers/classifier1_3/biases', 'l2_norm_conv4_3/scale', 'conv1_1/filter'):
        assert np.allclose(w1[k], w0[k] - np.float32(0.001) * g[k], rtol=1e-6, atol=1e-9)
    # and the next forward sees the updated masters through fresh mirrors
    r2, L2 = sess.run([net.result, net.losses], feed_dict={net.image_input: x, net.labels: y})
    assert L2['total'] != L['total']
    sess.close()


def test_bf16_inference_and_ragged_batch():
    """inference handle (no gradient storage), b < max_batch, detections agree with the fp32 handle's on clear boxes."""
    preset = ob.get_preset('vgg300')
    w = ref.init_params(preset, 20, seed=7, alive=True)
    sess = Session(0)
    nb = SSDVGG(sess, 'vgg300'); nb.build_from_vgg(None, 20, max_batch=4, training=False, weights=w, dtype='bf16')
    nf = SSDVGG(sess, 'vgg300'); nf.build_from_vgg(None, 20, max_batch=4, training=False, weights=w)
    rng = np.random.default_rng(3)
    x, _, _ = ref.synth_batch(rng, 3, preset)
    rb = sess.run(nb.result, feed_dict={nb.image_input: x, nb.keep_prob: 1})
    rf = sess.run(nf.result, feed_dict={nf.image_input: x, nf.keep_prob: 1})
    assert rb.shape == rf.shape == (3, 8732, 25)
    e = max_rel(rb, rf)
    print('    bf16 vs fp32 inference result max-rel', e)
    assert e < 0.05 and np.abs(rb[..., :21].sum(-1) - 1).max() < 1e-4
    sess.close()
