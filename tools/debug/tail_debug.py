import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from gpu_util import lib, check, ptr, conv_geom
def run(name, b, hi, wi, ci, co, k, stride, padding, f32, relu=1):
    ph, pw, ho, wo = conv_geom(hi, wi, k, stride, 1, padding)
    torch.manual_seed(0)
    x = torch.randn((b, hi, wi, ci), device='cuda').bfloat16()
    w = (torch.randn((k * k, co, ci), device='cuda') / (k * k * ci) ** 0.5).bfloat16()
    bias = torch.randn((co,), device='cuda') * 0.1
    geom = (b, hi, wi, ci, ho, wo, co, k, k, stride, 1, ph, pw)
    ys = []
    for fn in (lib.ssd_op_conv2d_fwd_bf16_chain, lib.ssd_op_conv2d_fwd_bf16):
        y = torch.full((b, ho, wo, co), 9.0, device='cuda', dtype=torch.float32 if f32 else torch.bfloat16)
        check(fn(ptr(x), ptr(w), ptr(bias), ptr(y), f32, *geom, relu, None))
        torch.cuda.synchronize()
        ys.append(y.float().cpu().numpy())
    d = np.abs(ys[0] - ys[1])
    print(name, 'max diff', d.max(), 'ref max', np.abs(ys[1]).max(), 'frac bad', (d > 1e-2 * np.abs(ys[1]).max()).mean())
    bad = np.argwhere(d > 1e-2 * np.abs(ys[1]).max())
    if len(bad):
        print('  first bad', bad[:6].tolist(), ' chain', ys[0][tuple(bad[0])], 'ref', ys[1][tuple(bad[0])])
        print('  bad pixels (h,w) set', sorted({(int(r[1]), int(r[2])) for r in bad})[:30])
        print('  bad channels', sorted({int(r[3]) for r in bad})[:40])
run('1x1 256->128 5x5', 1, 5, 5, 256, 128, 1, 1, 'SAME', 0)
run('1x1 32->16 1x1', 1, 1, 1, 32, 16, 1, 1, 'SAME', 0)
run('1x1 64->256 4x4', 1, 4, 4, 64, 256, 1, 1, 'SAME', 0)
run('3x3 128->256 5x5 valid', 2, 5, 5, 128, 256, 3, 1, 'VALID', 0)
run('head', 1, 5, 5, 256, 152, 3, 1, 'SAME', 1, 0)
