#!/usr/bin/env python3
"""Times ONE stage of the tail chain kernel (csrc/tail_bf16.hip, one workgroup per image) against the per-layer kernel on the same
shape, over the number of images: is a workgroup's filter stream slowed by the other workgroups reading the same lines?
    tools/probes/tail_stage_time.py [shape]      shape: conv9_2 (default), head3, conv10_1"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from gpu_util import lib, check, ptr, conv_geom

SHAPES = {'conv9_2': (10, 10, 128, 256, 3, 2, 'SAME', 0), 'head3': (5, 5, 256, 152, 3, 1, 'SAME', 1), 'conv10_1': (5, 5, 256, 128, 1, 1, 'SAME', 0),
          'conv10_2': (5, 5, 128, 256, 3, 1, 'VALID', 0)}
name = sys.argv[1] if len(sys.argv) > 1 else 'conv9_2'
hi, wi, ci, co, k, stride, padding, f32 = SHAPES[name]
ph, pw, ho, wo = conv_geom(hi, wi, k, stride, 1, padding)
for b in (1, 2, 4, 8, 16, 32, 64, 128):
    x = torch.randn((b, hi, wi, ci), device='cuda').bfloat16()
    w = torch.randn((k * k, co, ci), device='cuda').bfloat16()
    bias = torch.zeros((co,), device='cuda')
    y = torch.empty((b, ho, wo, co), device='cuda', dtype=torch.float32 if f32 else torch.bfloat16)
    geom = (b, hi, wi, ci, ho, wo, co, k, k, stride, 1, ph, pw)
    out = []
    for fn in (lib.ssd_op_conv2d_fwd_bf16_chain, lib.ssd_op_conv2d_fwd_bf16):
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(5):
            check(fn(ptr(x), ptr(w), ptr(bias), ptr(y), f32, *geom, 1, s))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50):
            check(fn(ptr(x), ptr(w), ptr(bias), ptr(y), f32, *geom, 1, s))
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 50 * 1e3)
    print(f'{name} b={b:4d}: chain {out[0]:7.1f} us   per-layer kernel {out[1]:7.1f} us', flush=True)
