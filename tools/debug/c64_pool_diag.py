"""diagnostic: where does the fused 64->64 pool forward differ from conv + pool?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
from gpu_util import lib, check, dev, ptr
from test_gpu_pool_fusion import bdev, _inputs

for case in [('c64 even 40x36 64->64', 2, 40, 36, 64, 64), ('c64 tiny', 1, 4, 6, 64, 64), ('c64 8x62', 1, 8, 62, 64, 64)]:
    name, b, h, w, ci, co = case
    rng, x, wt, bias = _inputs(*case)
    ph, pw = (h + 1) // 2, (w + 1) // 2
    geom = (b, h, w, ci, h, w, co, 3, 3, 1, 1, 1, 1)
    x_, w_, b_ = bdev(x), dev(wt), dev(bias)
    wio = torch.empty((3, 3, ci, co), dtype=torch.bfloat16, device='cuda'); woi = torch.empty((3, 3, co, ci), dtype=torch.bfloat16, device='cuda')
    check(lib.ssd_op_cast_filter(ptr(w_), ptr(wio), ptr(woi), 9, ci, co, None))
    y_ = torch.empty((b, h, w, co), dtype=torch.bfloat16, device='cuda')
    check(lib.ssd_op_conv2d_fwd_bf16(ptr(x_), ptr(woi), ptr(b_), ptr(y_), 0, *geom, 1, None))
    p_ref = torch.zeros((b, ph, pw, co), dtype=torch.bfloat16, device='cuda'); r_ref = torch.zeros((b, ph, pw, co // 4), dtype=torch.int16, device='cuda')
    check(lib.ssd_op_maxpool_rec_fwd(ptr(y_), ptr(p_ref), ptr(r_ref), 1, b, h, w, co, None))
    p_got = torch.full((b, ph, pw, co), 9.0, dtype=torch.bfloat16, device='cuda'); r_got = torch.full((b, ph, pw, co // 4), -2, dtype=torch.int16, device='cuda')
    check(lib.ssd_op_conv2d_fwd_pool_bf16(ptr(x_), ptr(woi), ptr(b_), ptr(p_got), ptr(r_got), *geom, None))
    torch.cuda.synchronize()
    a, r = p_got.float().cpu().numpy(), p_ref.float().cpu().numpy()
    bad = np.argwhere(a != r)
    print(name, 'mismatches', len(bad), 'of', a.size, ' rec mismatches', int((r_got != r_ref).sum().item()))
    if len(bad):
        for ax, nm in enumerate(('b', 'ph', 'pw', 'c')):
            vals, cnt = np.unique(bad[:, ax], return_counts=True)
            print('   by', nm, dict(zip(vals.tolist()[:40], cnt.tolist()[:40])))
        for i in bad[:8]:
            print('   at', i.tolist(), 'got', a[tuple(i)], 'want', r[tuple(i)], ' unpooled window', y_.float().cpu().numpy()[i[0], 2 * i[1]:2 * i[1] + 2, 2 * i[2]:2 * i[2] + 2, i[3]].tolist())
