"""Micro-benchmark of the conv kernels on representative vgg300 layers at batch 32 (GPU).
  python tools/bench_conv.py [layer,layer,...] [f32|bf16]
Env: SSD_TILE / SSD_WGRAD_CFG (fp32) and SSD_TILE_BF16 / SSD_WGRAD_CFG_BF16 (bf16) force a tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from gpu_util import lib, check, ptr, conv_geom

LAYERS = [('conv1_2', 300, 64, 64, 3, 1, 1), ('conv2_1', 150, 64, 128, 3, 1, 1), ('conv2_2', 150, 128, 128, 3, 1, 1), ('conv3_1', 75, 128, 256, 3, 1, 1),
          ('conv3_2', 75, 256, 256, 3, 1, 1), ('conv4_1', 38, 256, 512, 3, 1, 1),
          ('conv4_2', 38, 512, 512, 3, 1, 1), ('conv5_2', 19, 512, 512, 3, 1, 1), ('mod_conv6', 19, 512, 1024, 3, 1, 6),
          ('mod_conv7', 19, 1024, 1024, 1, 1, 1), ('head1', 19, 1024, 152, 3, 1, 1), ('head0', 38, 512, 100, 3, 1, 1),
          # the small tail of the network (latency-bound: a handful of workgroups each)
          ('conv8_1', 19, 1024, 256, 1, 1, 1), ('conv8_2', 19, 256, 512, 3, 2, 1), ('conv9_1', 10, 512, 128, 1, 1, 1),
          ('conv9_2', 10, 128, 256, 3, 2, 1), ('conv10_1', 5, 256, 128, 1, 1, 1), ('conv10_2', 5, 128, 256, 3, 1, 1),
          ('head2', 10, 512, 152, 3, 1, 1), ('head3', 5, 256, 152, 3, 1, 1)]
only = sys.argv[1].split(',') if len(sys.argv) > 1 and sys.argv[1] != 'all' else None
BF16 = len(sys.argv) > 2 and sys.argv[2] == 'bf16'
B = int(os.environ.get("SSD_BENCH_B", "32"))      # batch (SSD_BENCH_B: quantisation / tail experiments)


def bench_bf16(name, hw, ci, co, k, s, d):
    co = (co + 7) // 8 * 8
    ph, pw, ho, wo = conv_geom(hw, hw, k, s, d, 'SAME')
    bf = torch.bfloat16
    ZERO = os.environ.get('SSD_BENCH_ZERO') == '1'      # DVFS probe: all-zero operands draw less power
    x = torch.randn((B, hw, hw, ci), device='cuda').to(bf); w = torch.randn((k, k, ci, co), device='cuda') * 0.05
    if ZERO:
        x.zero_(); w.zero_()
    RELU = os.environ.get('SSD_BENCH_RELU') == '1'      # operands like the real step's: post-relu activations, masked gradients
    if RELU:
        x = torch.relu(x)
    bias = torch.zeros(co, device='cuda'); y = torch.empty((B, ho, wo, co), device='cuda', dtype=bf)
    dy = torch.randn((B, ho, wo, co), device='cuda').to(bf)
    if ZERO:
        dy.zero_()
    if RELU:
        dy = (dy.float() * (torch.rand_like(dy.float()) > 0.5)).to(bf)
    dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty_like(bias)
    wio = torch.empty((k * k, ci, co), device='cuda', dtype=bf); woi = torch.empty((k * k, co, ci), device='cuda', dtype=bf)
    check(lib.ssd_op_cast_filter(ptr(w), ptr(wio), ptr(woi), k * k, ci, co, None))
    geom = (B, hw, hw, ci, ho, wo, co, k, k, s, d, ph, pw)
    ws = torch.empty((lib.ssd_op_conv2d_wgrad_bf16_ws_floats(*geom),), device='cuda')
    fl = 2.0 * B * ho * wo * co * ci * k * k
    return fl, dict(
        fwd=lambda: check(lib.ssd_op_conv2d_fwd_bf16(ptr(x), ptr(woi), ptr(bias), ptr(y), 0, *geom, 1, None)),
        dgrad=lambda: check(lib.ssd_op_conv2d_dgrad_bf16(ptr(dy), ptr(wio), ptr(dx), ptr(x), 0, *geom, None)),
        wgrad=lambda: check(lib.ssd_op_conv2d_wgrad_bf16(ptr(x), ptr(dy), ptr(dw), ptr(db), ptr(w), 0.0005, ptr(ws), *geom, None)))


def timeit(fns, fl, name):
    out = []
    for tag, fn in fns.items():
        for _ in range(3):
            fn()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        out.append(f'{tag} {ms:7.3f} ms {fl / ms / 1e9:6.1f} TF')
    print(f'{name:10s} ' + ' | '.join(out), flush=True)


for name, hw, ci, co, k, s, d in LAYERS:
    if only and name not in only:
        continue
    if BF16:
        fl, fns = bench_bf16(name, hw, ci, co, k, s, d)
        timeit(fns, fl, name)
        continue
    ph, pw, ho, wo = conv_geom(hw, hw, k, s, d, 'SAME')
    x = torch.randn((B, hw, hw, ci), device='cuda'); w = torch.randn((k, k, ci, co), device='cuda') * 0.05
    bias = torch.zeros(co, device='cuda'); y = torch.empty((B, ho, wo, co), device='cuda'); dy = torch.randn_like(y)
    if os.environ.get('SSD_BENCH_ZERO') == '1':
        x.zero_(); w.zero_(); dy.zero_()
    dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty_like(bias)
    geom = (B, hw, hw, ci, ho, wo, co, k, k, s, d, ph, pw)
    ws = torch.empty((lib.ssd_op_conv2d_wgrad_ws_floats(*geom),), device='cuda')
    fl = 2.0 * B * ho * wo * co * ci * k * k
    fns = dict(fwd=lambda: check(lib.ssd_op_conv2d_fwd(ptr(x), ptr(w), ptr(bias), ptr(y), *geom, 1, None)),
               dgrad=lambda: check(lib.ssd_op_conv2d_dgrad(ptr(dy), ptr(w), ptr(dx), ptr(x), 0, *geom, None)),
               wgrad=lambda: check(lib.ssd_op_conv2d_wgrad(ptr(x), ptr(dy), ptr(dw), ptr(db), ptr(w), 0.0005, ptr(ws), *geom, None)))
    nws = lib.ssd_op_conv2d_wino_ws_floats(*geom)
    if nws and os.environ.get('SSD_BENCH_WINO', '1') != '0':      # the Winograd forms, steady state (filter transforms / V current)
        wws = torch.empty((nws,), device='cuda')
        check(lib.ssd_op_conv2d_wino_fwd(ptr(x), ptr(w), ptr(bias), ptr(y), None, None, None, ptr(wws), 0, *geom, 1, None))
        fns.update(wino_fwd=lambda: check(lib.ssd_op_conv2d_wino_fwd(ptr(x), ptr(w), ptr(bias), ptr(y), None, None, None, ptr(wws), 1, *geom, 1, None)),
                   wino_dgrad=lambda: check(lib.ssd_op_conv2d_wino_dgrad(ptr(dy), ptr(w), ptr(dx), ptr(x), None, 0, None, 0, 0, ptr(wws), 1, *geom, None)),
                   wino_wgrad=lambda: check(lib.ssd_op_conv2d_wino_wgrad(ptr(x), ptr(dy), ptr(dw), ptr(db), ptr(w), 0.0005, ptr(wws), 3, *geom, None)))
    timeit(fns, fl, name)
