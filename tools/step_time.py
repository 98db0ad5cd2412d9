#!/usr/bin/env python3
"""Times the bare training step (no checks, random inputs) -- the A/B and ablation driver.
    tools/step_time.py [--dtype bf16|f32] [--preset vgg300] [--batch 32] [--steps 40] [--warmup 10] [--reps 2]
prints one line per repetition: ms per step.  Environment switches (SSD_*) select the variant under test."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import boxes as ob, ssdvgg_ref as ref          # synthetic batch only (test infrastructure)
from ssd_tensorflow_amd.ssdvgg import SSDVGG, Session
from ssd_tensorflow_amd._lib import lib, check


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--preset', default='vgg300')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--reps', type=int, default=2)
    ap.add_argument('--tag', default='')
    a = ap.parse_args()
    preset = ob.get_preset(a.preset)
    rng = np.random.default_rng(0)
    x, y, _ = ref.synth_batch(rng, a.batch, preset)
    with Session(0) as sess:
        net = SSDVGG(sess, a.preset)
        net.build_from_vgg(None, 20, max_batch=a.batch, dtype=a.dtype)
        net.build_optimizer(learning_rate=1e-4, weight_decay=0.0005, momentum=0.9)
        net.set_stream(torch.cuda.current_stream().cuda_stream)
        xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
        ablate = os.environ.pop('SSD_ABLATE', None)      # (this TOOL's switch; the library is told through ssd_debug_set_ablate)
        # warm up un-ablated: every buffer holds what a real step leaves in it
        for _ in range(a.warmup):
            net.train_step_dev(xd, yd)
        torch.cuda.synchronize()
        if ablate:
            check(lib.ssd_debug_set_ablate(ablate.encode()))
            for _ in range(3):
                net.train_step_dev(xd, yd)
        out = []
        for r in range(a.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                net.train_step_dev(xd, yd)
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / a.steps * 1e3)
        print(a.tag or ablate or 'base', a.dtype, ' '.join('%.3f' % v for v in out), 'ms/step', flush=True)


if __name__ == '__main__':
    main()
