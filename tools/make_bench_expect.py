#!/usr/bin/env python3
"""Step-0 losses of bench.py's own configurations, computed by the CPU oracle (oracle/ssdvgg_ref.py, fp32 torch-CPU)
on the very inputs bench.py generates on rank 0: images and boxes from default_rng(1234) in bench.py's draw order,
labels from the oracle's label encoder, weights = the library's own initialisation (seed 42) restated by
oracle.ssdvgg_ref.init_params_lib.  Written to tests/golden/bench_expect.json; bench.py refuses to report a training
number whose step-0 losses differ from these by more than 1e-3 (fp32).

    python tools/make_bench_expect.py [vgg300:32 vgg512:16 ...]        (minutes of CPU per configuration)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                     # noqa: E402  (synth_gt: the benchmark's own box generator)
from oracle import boxes as ob, ssdvgg_ref as ref      # noqa: E402


def expect(pname, b):
    preset = ob.get_preset(pname)
    rng = np.random.default_rng(1234)
    H, W = preset['image_size'][1], preset['image_size'][0]
    x = rng.integers(0, 256, (b, H, W, 3)).astype(np.float32)
    gt, cls, offs = bench.synth_gt(rng, b)
    anch = ob.anchors(preset); anch_abs = ob.anchors_abs(anch)
    y = np.stack([ob.encode_labels(gt[offs[i]:offs[i + 1]], cls[offs[i]:offs[i + 1]], preset, 20, anch, anch_abs) for i in range(b)])
    m = ref.RefModel(pname, params=ref.init_params_lib(preset, 20, seed=42))
    m.set_optimizer([0.00075], [], 0.9, 0.0005)
    t0 = time.time()
    L = None
    for i0 in range(0, b, 4):          # the loss is a mean of per-sample terms: accumulate over chunks of 4 images
        _, Lc = m.eval_step(x[i0:i0 + 4], y[i0:i0 + 4])
        n = min(4, b - i0)
        if L is None:
            L = {k: 0.0 for k in Lc}
        for k in ('localization', 'confidence'):
            L[k] += Lc[k] * n / b
        L['l2'] = Lc['l2']
    L['total'] = L['localization'] + L['confidence'] + L['l2']
    print(pname, b, L, f'{time.time() - t0:.0f}s', flush=True)
    return L


def main():
    todo = sys.argv[1:] or ['vgg300:32', 'vgg512:16']
    path = os.path.join(ROOT, 'tests', 'golden', 'bench_expect.json')
    table = json.load(open(path)) if os.path.exists(path) else {}
    for t in todo:
        pname, b = t.split(':')
        table[f'{pname}_b{int(b)}'] = expect(pname, int(b))
        json.dump(table, open(path, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
