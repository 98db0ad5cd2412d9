"""-m gpu: every HIP kernel of the conv stack against the CPU oracle (torch-CPU fp32 ops with
TF semantics, oracle/ssdvgg_ref.py), through the C ABI.  Tolerance: 1e-3 relative
(BASELINE.json north_star), in practice ~1e-6 (fp32 MFMA is an exact fmaf chain)."""
import zlib
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ssdvgg_ref as ref
from gpu_util import lib, check, dev, ptr, host, rel_err, max_rel, conv_geom

pytestmark = pytest.mark.gpu
TOL = 1e-3

# (name, b, hi, wi, ci, co, k, stride, dil, padding, relu)
CONV_CASES = [
    ('conv1_1-like smallC', 2, 37, 41, 3, 64, 3, 1, 1, 'SAME', True),
    ('conv1_1-like smallC, 1-wide image rows, many pixels', 3, 130, 1, 3, 64, 3, 1, 1, 'SAME', True),
    ('conv1_2-like 64->64', 2, 40, 33, 64, 64, 3, 1, 1, 'SAME', True),
    ('conv2_1-like 64->128', 1, 30, 30, 64, 128, 3, 1, 1, 'SAME', True),
    ('conv4-like 256->512', 1, 19, 19, 256, 512, 3, 1, 1, 'SAME', True),
    ('mod_conv6 dil6', 2, 19, 19, 512, 1024, 3, 1, 6, 'SAME', True),
    ('mod_conv7 1x1', 2, 19, 19, 1024, 1024, 1, 1, 1, 'SAME', True),
    ('conv8_1 1x1 ->256', 2, 19, 19, 1024, 256, 1, 1, 1, 'SAME', True),
    ('conv8_2 s2 19->10', 2, 19, 19, 256, 512, 3, 2, 1, 'SAME', True),
    ('conv9_2 s2 10->5 asym', 2, 10, 10, 128, 256, 3, 2, 1, 'SAME', True),
    ('conv10_2 VALID 5->3', 2, 5, 5, 128, 256, 3, 1, 1, 'VALID', True),
    ('conv11_2 VALID 3->1', 3, 3, 3, 128, 256, 3, 1, 1, 'VALID', True),
    ('conv12_2 pad-BR 2->1', 2, 2, 2, 128, 256, 3, 1, 1, 'BR1', True),
    ('head 6 types N=152', 2, 10, 10, 512, 152, 3, 1, 1, 'SAME', False),
    ('head 4 types N=100', 2, 38, 38, 512, 100, 3, 1, 1, 'SAME', False),
    ('ragged M, 1 image', 1, 13, 7, 128, 128, 3, 1, 1, 'SAME', True),
]


# sizes at which the batch-32 step picks its big tiles (M = 90k..180k pixels) + a ragged one: run under forced variants
LARGE_CASES = [
    ('conv2_2-size 150x150 128->128 b4', 4, 150, 150, 128, 128, 3, 1, 1, 'SAME', True),
    ('conv3_2-size 75x75 256->256 b3 ragged M', 3, 75, 75, 256, 256, 3, 1, 1, 'SAME', True),
    ('conv4_2-size 38x38 512->512 b5', 5, 38, 38, 512, 512, 3, 1, 1, 'SAME', True),
    ('ragged channels 33x29 192->320 b2', 2, 33, 29, 192, 320, 3, 1, 1, 'SAME', True),
]


def oracle_conv(x, w, bias, stride, dil, padding, relu):
    xt = torch.tensor(x).permute(0, 3, 1, 2).requires_grad_(True)
    wt = torch.tensor(w).requires_grad_(True)
    bt = torch.tensor(bias).requires_grad_(True)
    xin = F.pad(xt, (0, 1, 0, 1)) if padding == 'BR1' else xt
    y = ref.conv2d_tf(xin, wt, stride, 'SAME' if padding == 'SAME' else 'VALID', dil) + bt.view(1, -1, 1, 1)
    pre = y
    if relu:
        y = F.relu(y)
    return xt, wt, bt, pre, y


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(case):
    name, b, hi, wi, ci, co, k, stride, dil, padding, relu = case
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    ph, pw, ho, wo = conv_geom(hi, wi, k, stride, dil, padding)
    x = rng.normal(0, 1, (b, hi, wi, ci)).astype(np.float32)
    w = (rng.normal(0, 1, (k, k, ci, co)) / np.sqrt(k * k * ci)).astype(np.float32)
    bias = rng.normal(0, 0.1, (co,)).astype(np.float32)
    if co == 152:          # fused-head padding columns are zero in the product
        w[..., 150:] = 0; bias[150:] = 0
    dy = rng.normal(0, 1, (b, ho, wo, co)).astype(np.float32)

    xt, wt, bt, pre, y_ref = oracle_conv(x, w, bias, stride, dil, padding, relu)
    assert tuple(y_ref.shape) == (b, co, ho, wo)
    # the product works on pre-activation gradients: dy masked by relu
    g = torch.tensor(dy).permute(0, 3, 1, 2)
    gpre = g * (pre > 0).float() if relu else g
    pre.backward(gpre)
    dx_ref = xt.grad.permute(0, 2, 3, 1).numpy()
    dw_ref = wt.grad.numpy()
    db_ref = bt.grad.numpy()
    dy_pre = gpre.permute(0, 2, 3, 1).contiguous().numpy()

    dx_, dw_, db_ = dev(x), dev(w), dev(bias)
    y_ = torch.empty((b, ho, wo, co), dtype=torch.float32, device='cuda')
    geom = (b, hi, wi, ci, ho, wo, co, k, k, stride, dil, ph, pw)
    check(lib.ssd_op_conv2d_fwd(ptr(dx_), ptr(dw_), ptr(db_), ptr(y_), *geom, int(relu), None))
    y = host(y_)
    e = max_rel(y, y_ref.detach().permute(0, 2, 3, 1).numpy())
    assert e < TOL, f'{name}: forward max-rel {e:.3e}'

    # weight gradient (+ bias gradient + weight decay)
    wd = 0.0005
    nws = lib.ssd_op_conv2d_wgrad_ws_floats(*geom)
    ws_ = torch.empty((nws,), dtype=torch.float32, device='cuda')
    gdy_ = dev(dy_pre)
    gw_ = torch.full((k, k, ci, co), 7.0, dtype=torch.float32, device='cuda')
    gb_ = torch.full((co,), 7.0, dtype=torch.float32, device='cuda')
    check(lib.ssd_op_conv2d_wgrad(ptr(dx_), ptr(gdy_), ptr(gw_), ptr(gb_), ptr(dw_), wd, ptr(ws_), *geom, None))
    e = max_rel(host(gw_), dw_ref + wd * w)
    assert e < TOL, f'{name}: wgrad max-rel {e:.3e}'
    e = max_rel(host(gb_), db_ref)
    assert e < TOL, f'{name}: bias-grad max-rel {e:.3e}'

    # data gradient: plain, then accumulate + relu mask of the producer
    if ci % 4 == 0:
        gx_ = torch.full((b, hi, wi, ci), 3.0, dtype=torch.float32, device='cuda')
        check(lib.ssd_op_conv2d_dgrad(ptr(gdy_), ptr(dw_), ptr(gx_), None, 0, *geom, None))
        e = max_rel(host(gx_), dx_ref)
        assert e < TOL, f'{name}: dgrad max-rel {e:.3e}'
        prev = rng.normal(0, 1, x.shape).astype(np.float32)
        gx_ = dev(prev)
        check(lib.ssd_op_conv2d_dgrad(ptr(gdy_), ptr(dw_), ptr(gx_), ptr(dx_), 1, *geom, None))
        expect = (dx_ref + prev) * (x > 0)
        e = max_rel(host(gx_), expect)
        assert e < TOL, f'{name}: dgrad accumulate+mask max-rel {e:.3e}'


@pytest.mark.parametrize('case', LARGE_CASES, ids=[c[0] for c in LARGE_CASES])
def test_conv_large_layers(case):
    test_conv_fwd_dgrad_wgrad(case)


# every fp32 tile / parity switch the step can take, forced for ALL layers of a child process (the library
# reads the switches once): the cost model only picks most of them at batch-32 sizes.  (Round 5 removed the register-staged
# gather kernels, the deep-ring / k-split small-layer twins and the split-count switch together with their variants here.)
FP32_VARIANTS = [
    dict(SSD_TILE='0', SSD_WGRAD_CFG='0'),                                            # 128x128 everywhere
    dict(SSD_TILE='1', SSD_WGRAD_CFG='1'),                                            # 128x64 / 64x64
    dict(SSD_TILE='2', SSD_WGRAD_CFG='2', SSD_DGRAD_PARITY='0'),                      # 64x128, all-taps strided data gradient
    dict(SSD_TILE='3', SSD_WGRAD_CFG='3', SSD_FIRST_F32='0', SSD_FIRST_WGRAD_F32='0'),  # 64x64 / 128x64, conv1_1 on the generic small-C kernels
]


@pytest.mark.parametrize('variant', FP32_VARIANTS, ids=[' '.join(f'{k[4:]}={v}' for k, v in d.items()) for d in FP32_VARIANTS])
def test_conv_fp32_forced_variants(variant):
    import os, subprocess, sys
    if os.environ.get('SSD_VARIANT_CHILD') == '1':
        pytest.skip('already a forced configuration')
    env = dict(os.environ, SSD_VARIANT_CHILD='1', **variant)
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-x', '-q', '-k',
                        'test_conv_fwd_dgrad_wgrad or test_conv_large_layers or test_conv_full_size_layer', '-p', 'no:cacheprovider'],
                       env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_conv_full_size_layer():
    """conv1_2 at the full vgg300 size for one image (M = 90,000 pixels): tile tails, XCD remap."""
    rng = np.random.default_rng(5)
    b, hi, wi, ci, co = 1, 300, 300, 64, 64
    x = rng.normal(0, 1, (b, hi, wi, ci)).astype(np.float32)
    w = (rng.normal(0, 1, (3, 3, ci, co)) / 24).astype(np.float32)
    bias = rng.normal(0, 0.1, (co,)).astype(np.float32)
    _, _, _, _, y_ref = oracle_conv(x, w, bias, 1, 1, 'SAME', True)
    x_, w_, b_ = dev(x), dev(w), dev(bias)
    y_ = torch.empty((b, hi, wi, co), dtype=torch.float32, device='cuda')
    check(lib.ssd_op_conv2d_fwd(ptr(x_), ptr(w_), ptr(b_), ptr(y_), b, hi, wi, ci, hi, wi, co, 3, 3, 1, 1, 1, 1, 1, None))
    assert max_rel(host(y_), y_ref.detach().permute(0, 2, 3, 1).numpy()) < TOL


POOL_CASES = [('pool 2x2 s2 even', 2, 20, 20, 64, 2, 2), ('pool3 75->38 ceil', 2, 75, 75, 64, 2, 2),
              ('mod_pool5 3x3 s1', 2, 19, 19, 128, 3, 1), ('pool odd 7x5', 1, 7, 5, 8, 2, 2)]


@pytest.mark.parametrize('case', POOL_CASES, ids=[c[0] for c in POOL_CASES])
def test_maxpool(case):
    name, b, hi, wi, c, k, s = case
    rng = np.random.default_rng(11)
    # relu-like input: many exact zeros (ties), distinct positives
    x = np.maximum(rng.normal(0, 1, (b, hi, wi, c)), 0).astype(np.float32)
    ph, pw, ho, wo = conv_geom(hi, wi, k, s, 1, 'SAME')
    xt = torch.tensor(x).permute(0, 3, 1, 2).requires_grad_(True)
    y_ref = ref.maxpool_tf(xt, k, s)
    assert tuple(y_ref.shape) == (b, c, ho, wo)
    dy = rng.normal(0, 1, (b, ho, wo, c)).astype(np.float32)
    y_ref.backward(torch.tensor(dy).permute(0, 3, 1, 2))
    dx_ref = xt.grad.permute(0, 2, 3, 1).numpy()

    x_, dy_ = dev(x), dev(dy)
    y_ = torch.empty((b, ho, wo, c), dtype=torch.float32, device='cuda')
    geom = (b, hi, wi, c, ho, wo, k, s, ph, pw)
    check(lib.ssd_op_maxpool_fwd(ptr(x_), ptr(y_), *geom, None))
    assert np.array_equal(host(y_), y_ref.detach().permute(0, 2, 3, 1).numpy())
    # with the relu mask of the producer (zeros carry no gradient, so tie order among zeros is moot)
    dx_ = torch.full(x.shape, 5.0, dtype=torch.float32, device='cuda')
    check(lib.ssd_op_maxpool_bwd(ptr(x_), ptr(dy_), ptr(dx_), 0, 1, *geom, None))
    expect = dx_ref * (x > 0)
    assert np.abs(host(dx_) - expect).max() < 1e-6
    # accumulate
    prev = rng.normal(0, 1, x.shape).astype(np.float32)
    dx_ = dev(prev)
    check(lib.ssd_op_maxpool_bwd(ptr(x_), ptr(dy_), ptr(dx_), 1, 1, *geom, None))
    assert np.abs(host(dx_) - (dx_ref + prev) * (x > 0)).max() < 1e-5
    # no mask, strictly positive input (no ties): must equal autograd exactly
    xp = (np.abs(rng.normal(0, 1, x.shape)) + 0.01).astype(np.float32)
    xt = torch.tensor(xp).permute(0, 3, 1, 2).requires_grad_(True)
    ref.maxpool_tf(xt, k, s).backward(torch.tensor(dy).permute(0, 3, 1, 2))
    x_ = dev(xp)
    check(lib.ssd_op_maxpool_bwd(ptr(x_), ptr(dy_), ptr(dx_), 0, 0, *geom, None))
    assert np.abs(host(dx_) - xt.grad.permute(0, 2, 3, 1).numpy()).max() < 1e-6


@pytest.mark.parametrize('npix,c', [(2 * 38 * 38, 512), (77, 512), (5, 256), (9, 260), (7, 1000), (130, 1024), (6, 64)])
def test_l2norm(npix, c):
    rng = np.random.default_rng(2)
    x = np.maximum(rng.normal(0, 1, (npix, c)), 0).astype(np.float32)
    x[3] = 0                                   # an all-zero pixel: eps branch
    scale = (20 + rng.normal(0, 1, (c,))).astype(np.float32)
    dy = rng.normal(0, 1, (npix, c)).astype(np.float32)
    xt = torch.tensor(x).t().reshape(1, c, npix, 1).contiguous().requires_grad_(True)   # NCHW, W=1
    st = torch.tensor(scale).requires_grad_(True)
    y_ref = ref.l2norm_tf(xt, st)
    y_ref.backward(torch.tensor(dy).t().reshape(1, c, npix, 1))
    y_ref = y_ref.detach().reshape(c, npix).t().numpy()
    dx_ref = xt.grad.reshape(c, npix).t().numpy()
    ds_ref = st.grad.numpy()

    x_, s_, dy_ = dev(x), dev(scale), dev(dy)
    y_ = torch.empty_like(x_)
    check(lib.ssd_op_l2norm_fwd(ptr(x_), ptr(s_), ptr(y_), npix, c, None))
    assert max_rel(host(y_), y_ref) < 1e-5
    dx_ = torch.empty_like(x_)
    ds_ = torch.empty_like(s_)
    ws_ = torch.empty((lib.ssd_op_l2norm_bwd_ws_floats(npix, c),), dtype=torch.float32, device='cuda')
    check(lib.ssd_op_l2norm_bwd(ptr(x_), ptr(s_), ptr(dy_), ptr(dx_), ptr(ds_), ptr(ws_), npix, c, None))
    got = host(dx_)
    mask = np.ones(npix, bool); mask[3] = False      # d/dx at x == 0 under max(., eps): see below
    assert max_rel(got[mask], dx_ref[mask]) < 1e-4
    # at the all-zero pixel the norm is the constant sqrt(eps): dx = scale*dy/sqrt(eps)
    assert max_rel(got[3], scale * dy[3] * 1e6) < 1e-4
    assert max_rel(host(ds_), ds_ref) < 1e-4


def test_errors_are_reported():
    rc = lib.ssd_op_conv2d_fwd(None, None, None, None, 1, 8, 8, 6, 8, 8, 8, 3, 3, 1, 1, 1, 1, 1, None)   # Ci=6
    assert rc != 0 and b'multiple of 4' in lib.ssd_last_error()
