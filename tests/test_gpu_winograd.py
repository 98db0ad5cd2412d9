"""-m gpu: the Winograd F(4x4, 3x3) forms of the fp32 3x3 / stride 1 / SAME passes (csrc/winograd.hip) against the CPU oracle
(torch-CPU fp32 with TF semantics, oracle/ssdvgg_ref.py; tf.nn.conv2d at ssdvgg.py:195-207), through the C ABI.  Tolerance: 1e-3
relative (BASELINE.json north_star); the minimal-filtering transforms are exact identities, their fp32 rounding measures 6e-7..6e-6 on
the interpolation points 0, +-3/4, +-3/2, inf (Lavin's 0, +-1, +-2: up to 1.7e-5).
The fused-pool forms are checked bit for bit against the stand-alone pooling passes applied to the Winograd kernels' own output."""
import zlib
import numpy as np
import pytest
import torch

from gpu_util import lib, check, dev, ptr, host, max_rel
from test_gpu_kernels import oracle_conv

pytestmark = pytest.mark.gpu
TOL = 1e-3

# (name, b, h, w, ci, co)
CASES = [
    ('conv5-like 19x19 (5 tile rows, last one 3 deep) 256->512', 1, 19, 19, 256, 512),
    ('conv4_2-size 38x38 512->512 b2', 2, 38, 38, 512, 512),
    ('conv3_2-size 75x75 256->256 b3 (ragged tile count)', 3, 75, 75, 256, 256),
    ('ragged channel tiles 33x29 192->320 b2', 2, 33, 29, 192, 320),
    ('conv2_1-like 150x150 64->128', 1, 150, 150, 64, 128),
    ('tiny 3x2 32->32 b5 (one tile per image)', 5, 3, 2, 32, 32),
    ('exact tiles 8x12 128->96', 2, 8, 12, 128, 96),
    # dilated (a-trous, ssdvgg.py:260-262): every residue class (h mod 6, w mod 6) is an image of its own, one tile each at 19x19
    ('mod_conv6 dil 6 19x19 256->512', 2, 19, 19, 256, 512, 6),
    ('mod_conv6 of vgg512: dil 6 32x32 128->256 (two tiles per class)', 1, 32, 32, 128, 256, 6),
    ('dil 2 odd 13x9 64->64', 3, 13, 9, 64, 64, 2),
    # the multibox heads' channel counts (4 / 6 boxes x 25, padded to 104 / 152): the data gradient's k is padded to 128 / 160 with zeros
    ('head map0-like 38x38 256->104', 1, 38, 38, 256, 104),
    ('head map1-like 19x19 512->152 b2', 2, 19, 19, 512, 152),
    ('12 output channels 9x9 32->12', 2, 9, 9, 32, 12),
    ('128 -> 64 channels 21x10 (128 x 64 tile of the weight-gradient GEMM)', 2, 21, 10, 128, 64),
]


def raw(t):
    torch.cuda.synchronize()
    return t.cpu().numpy().view(np.uint8 if t.dtype != torch.float32 else np.uint32)


def _run(name, b, h, w, ci, co, dil=1, relu=True):
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    x = np.maximum(rng.normal(0, 1, (b, h, w, ci)), 0).astype(np.float32)
    wt = (rng.normal(0, 1, (3, 3, ci, co)) / np.sqrt(9 * ci)).astype(np.float32)
    bias = rng.normal(0, 0.1, (co,)).astype(np.float32)
    dy = rng.normal(0, 1, (b, h, w, co)).astype(np.float32)
    xt, wtt, bt, pre, y_ref = oracle_conv(x, wt, bias, 1, dil, 'SAME', relu)
    g = torch.tensor(dy).permute(0, 3, 1, 2)
    gpre = g * (pre > 0).float() if relu else g
    pre.backward(gpre)
    dx_ref = xt.grad.permute(0, 2, 3, 1).numpy()
    dw_ref, db_ref = wtt.grad.numpy(), bt.grad.numpy()
    dy_pre = gpre.permute(0, 2, 3, 1).contiguous().numpy()

    geom = (b, h, w, ci, h, w, co, 3, 3, 1, dil, dil, dil)
    nws = lib.ssd_op_conv2d_wino_ws_floats(*geom)
    assert nws > 0
    ws_ = torch.empty((nws,), dtype=torch.float32, device='cuda')
    x_, w_, b_ = dev(x), dev(wt), dev(bias)
    y_ = torch.full((b, h, w, co), 7.0, dtype=torch.float32, device='cuda')
    check(lib.ssd_op_conv2d_wino_fwd(ptr(x_), ptr(w_), ptr(b_), ptr(y_), None, None, None, ptr(ws_), 0, *geom, int(relu), None))
    e_f = max_rel(host(y_), y_ref.detach().permute(0, 2, 3, 1).numpy())
    assert e_f < TOL, f'{name}: forward max-rel {e_f:.3e}'

    wd = 0.0005
    gdy_ = dev(dy_pre)
    gw_ = torch.full((3, 3, ci, co), 7.0, dtype=torch.float32, device='cuda')
    gb_ = torch.full((co,), 7.0, dtype=torch.float32, device='cuda')
    # flags 3: filter transforms and the input's transform are the forward call's
    check(lib.ssd_op_conv2d_wino_wgrad(ptr(x_), ptr(gdy_), ptr(gw_), ptr(gb_), ptr(w_), wd, ptr(ws_), 3, *geom, None))
    e_w = max_rel(host(gw_), dw_ref + wd * wt)
    assert e_w < TOL, f'{name}: wgrad max-rel {e_w:.3e}'
    e_b = max_rel(host(gb_), db_ref)
    assert e_b < TOL, f'{name}: bias-grad max-rel {e_b:.3e}'
    # ... and from scratch (the call transforms x itself): the same bits
    gw2_ = torch.full((3, 3, ci, co), 8.0, dtype=torch.float32, device='cuda')
    gb2_ = torch.full((co,), 8.0, dtype=torch.float32, device='cuda')
    ws2_ = torch.zeros((nws,), dtype=torch.float32, device='cuda')
    check(lib.ssd_op_conv2d_wino_wgrad(ptr(x_), ptr(gdy_), ptr(gw2_), ptr(gb2_), ptr(w_), wd, ptr(ws2_), 0, *geom, None))
    assert np.array_equal(raw(gw_), raw(gw2_)) and np.array_equal(raw(gb_), raw(gb2_))

    gx_ = torch.full((b, h, w, ci), 3.0, dtype=torch.float32, device='cuda')
    check(lib.ssd_op_conv2d_wino_dgrad(ptr(gdy_), ptr(w_), ptr(gx_), None, None, 0, None, 0, 0, ptr(ws_), 0, *geom, None))
    e_d = max_rel(host(gx_), dx_ref)
    assert e_d < TOL, f'{name}: dgrad max-rel {e_d:.3e}'
    prev = rng.normal(0, 1, x.shape).astype(np.float32)
    gx_ = dev(prev)
    check(lib.ssd_op_conv2d_wino_dgrad(ptr(gdy_), ptr(w_), ptr(gx_), ptr(x_), None, 1, None, 0, 0, ptr(ws_), 1, *geom, None))
    e_m = max_rel(host(gx_), (dx_ref + prev) * (x > 0))
    assert e_m < TOL, f'{name}: dgrad accumulate+mask max-rel {e_m:.3e}'
    # the relu mask as the forward's bits (one 64-bit word per tile and 4 channels) instead of the fp32 tensor: the same bits out
    bits_ = torch.full((lib.ssd_op_conv2d_wino_bits_words(*geom),), -1, dtype=torch.int64, device='cuda')
    check(lib.ssd_op_conv2d_wino_fwd(ptr(x_), ptr(w_), ptr(b_), ptr(y_), None, None, ptr(bits_), ptr(ws_), 1, *geom, int(relu), None))
    gx2_ = dev(prev)
    check(lib.ssd_op_conv2d_wino_dgrad(ptr(gdy_), ptr(w_), ptr(gx2_), ptr(x_), ptr(bits_), 1, None, 0, 0, ptr(ws_), 1, *geom, None))
    assert np.array_equal(raw(gx2_), raw(gx_)), f'{name}: mask bits differ from the fp32 mask'
    print(f'{name}: fwd {e_f:.2e} wgrad {e_w:.2e} bias {e_b:.2e} dgrad {e_d:.2e}')
    return x_, w_, b_, y_, ws_, geom


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_winograd_fwd_dgrad_wgrad(case):
    _run(*case)


def test_winograd_no_relu():
    _run('no relu 10x10 64->64', 2, 10, 10, 64, 64, relu=False)


def test_winograd_fused_pool_refuses_dilation():
    geom = (1, 8, 8, 32, 8, 8, 32, 3, 3, 1, 2, 2, 2)
    ws_ = torch.empty((lib.ssd_op_conv2d_wino_ws_floats(*geom),), dtype=torch.float32, device='cuda')
    t = torch.zeros((1, 8, 8, 32), dtype=torch.float32, device='cuda')
    w_ = torch.zeros((3, 3, 32, 32), dtype=torch.float32, device='cuda')
    rc = lib.ssd_op_conv2d_wino_fwd(ptr(t), ptr(w_), None, None, ptr(t), None, None, ptr(ws_), 0, *geom, 1, None)
    assert rc != 0 and b'undilated' in lib.ssd_last_error()


POOL_CASES = [
    ('pool even 20x24 64->64', 2, 20, 24, 64, 64),
    ('pool3-like odd 75x75 (ceil) 32->128', 1, 75, 75, 32, 128),
    ('odd 7x5, ragged windows, 3 images', 3, 7, 5, 32, 32),
    ('1-pixel-wide image 9x1', 2, 9, 1, 32, 64),
]


@pytest.mark.parametrize('case', POOL_CASES, ids=[c[0] for c in POOL_CASES])
def test_winograd_fused_pool_bit_identical_to_the_pooling_passes(case):
    name, b, h, w, ci, co = case
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    x = np.maximum(rng.normal(0, 1, (b, h, w, ci)), 0).astype(np.float32)
    wt = (rng.normal(0, 1, (3, 3, ci, co)) / np.sqrt(9 * ci)).astype(np.float32)
    bias = rng.normal(0, 0.3, (co,)).astype(np.float32)
    ph, pw = (h + 1) // 2, (w + 1) // 2
    geom = (b, h, w, ci, h, w, co, 3, 3, 1, 1, 1, 1)
    x_, w_, b_ = dev(x), dev(wt), dev(bias)
    ws_ = torch.empty((lib.ssd_op_conv2d_wino_ws_floats(*geom),), dtype=torch.float32, device='cuda')
    y_ = torch.empty((b, h, w, co), dtype=torch.float32, device='cuda')
    check(lib.ssd_op_conv2d_wino_fwd(ptr(x_), ptr(w_), ptr(b_), ptr(y_), None, None, None, ptr(ws_), 0, *geom, 1, None))
    p_ref = torch.full((b, ph, pw, co), 7.0, dtype=torch.float32, device='cuda')
    r_ref = torch.full((b, ph, pw, co // 4), -1, dtype=torch.int16, device='cuda')
    check(lib.ssd_op_maxpool_rec_fwd(ptr(y_), ptr(p_ref), ptr(r_ref), 0, b, h, w, co, None))
    p_got = torch.full((b, ph, pw, co), 9.0, dtype=torch.float32, device='cuda')
    r_got = torch.full((b, ph, pw, co // 4), -2, dtype=torch.int16, device='cuda')
    check(lib.ssd_op_conv2d_wino_fwd(ptr(x_), ptr(w_), ptr(b_), None, ptr(p_got), ptr(r_got), None, ptr(ws_), 1, *geom, 1, None))
    assert np.array_equal(raw(p_got), raw(p_ref)), f'{name}: pooled tensor differs'
    assert np.array_equal(raw(r_got), raw(r_ref)), f'{name}: record differs'
    assert np.count_nonzero(host(p_ref)) > 0.3 * p_ref.numel()
    p2 = torch.full((b, ph, pw, co), 9.0, dtype=torch.float32, device='cuda')
    check(lib.ssd_op_conv2d_wino_fwd(ptr(x_), ptr(w_), ptr(b_), None, ptr(p2), None, None, ptr(ws_), 1, *geom, 1, None))
    assert np.array_equal(raw(p2), raw(p_ref))
    # ... and against the direct kernel's fused pool: same values to rounding
    p_dir = torch.empty_like(p_ref)
    check(lib.ssd_op_conv2d_fwd_pool(ptr(x_), ptr(w_), ptr(b_), ptr(p_dir), None, *geom, None))
    assert max_rel(host(p_got), host(p_dir)) < 1e-4

    # backward: a conv that READS the pooled tensor (co -> c2 channels); its data gradient through the record
    c2 = 64
    w2 = (rng.normal(0, 1, (3, 3, co, c2)) / np.sqrt(9 * co)).astype(np.float32)
    dy = rng.normal(0, 1, (b, ph, pw, c2)).astype(np.float32)
    w2_, dy_ = dev(w2), dev(dy)
    geom2 = (b, ph, pw, co, ph, pw, c2, 3, 3, 1, 1, 1, 1)
    ws2_ = torch.empty((lib.ssd_op_conv2d_wino_ws_floats(*geom2),), dtype=torch.float32, device='cuda')
    dxp = torch.full((b, ph, pw, co), 3.0, dtype=torch.float32, device='cuda')
    check(lib.ssd_op_conv2d_wino_dgrad(ptr(dy_), ptr(w2_), ptr(dxp), None, None, 0, None, 0, 0, ptr(ws2_), 0, *geom2, None))
    dx_ref = torch.full((b, h, w, co), 5.0, dtype=torch.float32, device='cuda')
    check(lib.ssd_op_maxpool_rec_bwd(ptr(r_ref), ptr(dxp), ptr(dx_ref), 1, 0, b, h, w, co, None))
    dx_got = torch.full((b, h, w, co), 6.0, dtype=torch.float32, device='cuda')
    check(lib.ssd_op_conv2d_wino_dgrad(ptr(dy_), ptr(w2_), ptr(dx_got), None, None, 0, ptr(r_ref), h, w, ptr(ws2_), 1, *geom2, None))
    assert np.array_equal(raw(dx_got), raw(dx_ref)), f'{name}: un-pooled data gradient differs'
    assert np.count_nonzero(host(dx_ref)) > 0.02 * dx_ref.numel()
    dx_dir = torch.empty_like(dx_ref)
    check(lib.ssd_op_conv2d_dgrad_unpool(ptr(dy_), ptr(w2_), ptr(dx_dir), ptr(r_ref), h, w, *geom2, None))
    assert max_rel(host(dx_got), host(dx_dir)) < 1e-4


def test_winograd_refuses_other_shapes():
    # stride 2, VALID padding, 1x1, input channels not in multiples of 32 / output channels not of 4: not this algorithm's
    assert lib.ssd_op_conv2d_wino_ws_floats(2, 19, 19, 256, 10, 10, 512, 3, 3, 2, 1, 0, 0) == 0
    assert lib.ssd_op_conv2d_wino_ws_floats(2, 5, 5, 128, 3, 3, 256, 3, 3, 1, 1, 0, 0) == 0
    assert lib.ssd_op_conv2d_wino_ws_floats(2, 19, 19, 512, 19, 19, 1024, 3, 3, 1, 6, 6, 6) > 0
    assert lib.ssd_op_conv2d_wino_ws_floats(2, 19, 19, 1024, 19, 19, 1024, 1, 1, 1, 1, 0, 0) == 0
    assert lib.ssd_op_conv2d_wino_ws_floats(2, 38, 38, 100, 38, 38, 512, 3, 3, 1, 1, 1, 1) == 0
    assert lib.ssd_op_conv2d_wino_ws_floats(2, 38, 38, 512, 38, 38, 102, 3, 3, 1, 1, 1, 1) == 0
    rc = lib.ssd_op_conv2d_wino_fwd(None, None, None, None, None, None, None, None, 0, 2, 38, 38, 100, 38, 38, 512, 3, 3, 1, 1, 1, 1, 1, None)
    assert rc != 0 and b'winograd' in lib.ssd_last_error()
