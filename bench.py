#!/usr/bin/env python3
"""Headline benchmark of the SSD-VGG hot path on MI355X (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one forward + multibox loss + backward + momentum update of `--preset` at
`--batch` images per GPU on synthetic data resident in HBM: the workload of
BASELINE.json configs[1] (vgg300, batch 32, fp32, one MI355X).  N > 1: one process per GPU,
the batch is sharded (weak scaling, 32 images per GPU), gradients are all-reduced over
RCCL/xGMI between backward and the update.

Rank 0 prints ONE JSON line: images/s over the whole job, the roofline block of the
dominant kernel (per-launch HIP events, on the launching stream, over the timed region) and
the CPU baseline (the fp32 torch-CPU oracle timed on this host; rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic conv FLOPs per image (SURVEY.md 8d / BASELINE.md 3)
FLOPS_FWD_BWD = {'vgg300': 187.93e9, 'vgg512': 540.34e9}
FLOPS_FWD = {'vgg300': 62.75e9, 'vgg512': 180.42e9}
PEAK_FP32_MFMA = 157.3      # TFLOP/s, MI355X_MICROARCH.md
PEAK_BF16_MFMA = 2500.0     # TFLOP/s dense bf16 (2.5 PFLOP/s), same guide
PEAK_HBM = 8000.0           # GB/s spec


# kernel label (event profiler) -> PREFIX of the demangled kernel symbol in rocprofv3 output.  Only the leading
# template arguments that select the tile are spelled out, so a kernel that gains a trailing template parameter
# still matches (r01's table carried the full argument list and went stale).
def _gather(mode, wm, wn, tm, tn):
    return ['conv_gather_dma_kernel<%d, %d, %d, %d, %d,' % (mode, wm, wn, tm, tn)]      # (LDS-DMA staging)


def _wgrad(wm, wn, tm, tn):
    t = '%d, %d, %d, %d' % (wm, wn, tm, tn)
    return ['conv_wgrad_kernel<' + t + ',', 'conv_wgrad_kernel<' + t + '>', 'conv_wgrad_dma_kernel<' + t + '>', 'conv_wgrad_dma_kernel<' + t + ',']


def _gbf(mode, wm, wn, tm, tn):
    return ['conv_gather_bf16_kernel<%d, %d, %d, %d, %d,' % (mode, wm, wn, tm, tn)]


def _wbf(wm, wn, tm, tn):
    return ['conv_wgrad_bf16_kernel<%d, %d, %d, %d,' % (wm, wn, tm, tn), 'conv_wgrad_bf16_kernel<%d, %d, %d, %d>' % (wm, wn, tm, tn)]


KERNEL_SYMBOLS = {
    'conv_fwd_128x128': _gather(0, 2, 2, 2, 2), 'conv_fwd_128x64': _gather(0, 4, 1, 1, 2),
    'conv_fwd_64x128': _gather(0, 2, 2, 1, 2), 'conv_fwd_64x64': _gather(0, 2, 2, 1, 1),
    'conv_dgrad_128x128': _gather(1, 2, 2, 2, 2), 'conv_dgrad_128x64': _gather(1, 4, 1, 1, 2),
    'conv_dgrad_64x128': _gather(1, 2, 2, 1, 2), 'conv_dgrad_64x64': _gather(1, 2, 2, 1, 1),
    'conv_wgrad_128x128': _wgrad(2, 2, 2, 2), 'conv_wgrad_64x64': _wgrad(2, 2, 1, 1),
    'conv_wgrad_64x128': _wgrad(2, 2, 1, 2), 'conv_wgrad_128x64': _wgrad(2, 2, 2, 1),
    # round 5: the same kernels with the 2x2 pool fused (forward: a POOL instantiation; data gradient: the same instantiation, the
    # un-pool is a run-time branch of its epilogue)
    'conv_fwd_pool_128x128': _gather(0, 2, 2, 2, 2), 'conv_fwd_pool_128x64': _gather(0, 4, 1, 1, 2),
    'conv_fwd_pool_64x128': _gather(0, 2, 2, 1, 2), 'conv_fwd_pool_64x64': _gather(0, 2, 2, 1, 1),
    'conv_dgrad_unpool_128x128': _gather(1, 2, 2, 2, 2), 'conv_dgrad_unpool_128x64': _gather(1, 4, 1, 1, 2),
    'conv_dgrad_unpool_64x128': _gather(1, 2, 2, 1, 2), 'conv_dgrad_unpool_64x64': _gather(1, 2, 2, 1, 1),
    'conv_fwd_pool_bf16_c64': ['conv_fwd_pool_bf16_c64_kernel'], 'conv_fwd_pool_bf16_rows_256x128': ['conv_fwd_pool_bf16_rows_kernel'],
    'conv_dgrad_unpool_bf16_rows_256x128': ['conv_gather_bf16_rows_kernel<1, 4'], 'conv_dgrad_unpool_bf16_rows_128x128': ['conv_gather_bf16_rows_kernel<1, 2, 2'],
    'conv_dgrad_unpool_bf16_256x64_8w': _gbf(1, 8, 1, 1, 2), 'conv_dgrad_unpool_bf16_128x64': _gbf(1, 4, 1, 1, 2),
    # round 6: the Winograd F(4x4, 3x3) kernels of the fp32 trunk (csrc/winograd.hip)
    'wino_gemm_128x128': ['wino_gemm_nn_kernel<2, 2, 2, 2>'], 'wino_gemm_128x64': ['wino_gemm_nn_kernel<4, 1, 1, 2>'],
    'wino_gemm_64x128': ['wino_gemm_nn_kernel<2, 2, 1, 2>'], 'wino_gemm_64x64': ['wino_gemm_nn_kernel<2, 2, 1, 1>'],
    'wino_gemm_tn_128x128': ['wino_gemm_tn_kernel<2, 2, 2, 2>'], 'wino_gemm_tn_64x128': ['wino_gemm_tn_kernel<2, 2, 1, 2>'],
    'wino_gemm_tn_128x64': ['wino_gemm_tn_kernel<4, 1, 1, 2>'], 'wino_gemm_tn_64x64': ['wino_gemm_tn_kernel<2, 2, 1, 1>'],
    'wino_in': ['wino_in_kernel<true, false>'], 'wino_in_wgrad': ['wino_in_kernel<false, true>'],
    'wino_out': ['wino_out_kernel<0>'], 'wino_out_dgrad': ['wino_out_kernel<1>'], 'wino_out_pool': ['wino_out_kernel<2>'], 'wino_out_unpool': ['wino_out_kernel<3>'],
    'wino_wgrad_reduce': ['wino_wgrad_reduce_kernel'], 'wino_filter': ['wino_filter_kernel<false>'], 'wino_filter_flip': ['wino_filter_kernel<true>'],
    'detect_scan': ['detect_scan_kernel'], 'detect_image': ['detect_image_kernel'],
    'multibox_loss': ['heads_kernel<true>'], 'multibox_loss_grad': ['loss_grad_kernel<'], 'heads_result': ['heads_kernel<false>'],
    'conv_fwd_bf16_128x128': _gbf(0, 2, 2, 2, 2), 'conv_fwd_bf16_128x64': _gbf(0, 4, 1, 1, 2),
    'conv_fwd_bf16_64x128': _gbf(0, 2, 2, 1, 2), 'conv_fwd_bf16_256x64_8w': _gbf(0, 8, 1, 1, 2),
    'conv_dgrad_bf16_128x128': _gbf(1, 2, 2, 2, 2), 'conv_dgrad_bf16_128x64': _gbf(1, 4, 1, 1, 2),
    'conv_dgrad_bf16_64x128': _gbf(1, 2, 2, 1, 2), 'conv_dgrad_bf16_256x64_8w': _gbf(1, 8, 1, 1, 2),
    'conv_wgrad_bf16_128x128': _wbf(2, 2, 2, 2), 'conv_wgrad_bf16_64x64': _wbf(2, 2, 1, 1),
    'conv_wgrad_bf16_64x128': _wbf(2, 2, 1, 2), 'conv_wgrad_bf16_128x64': _wbf(2, 2, 2, 1),
    'conv_wgrad_bf16_rows_64x64': ['conv_wgrad_bf16_rows_kernel<1'], 'conv_wgrad_bf16_rows_64x128': ['conv_wgrad_bf16_rows_kernel<2'],
    'conv_wgrad_bf16_rows8_128x128': ['conv_wgrad_bf16_rows8_kernel<'],
    'conv_fwd_bf16_rows_128x128': ['conv_gather_bf16_rows_kernel<0, 2, 2'], 'conv_dgrad_bf16_rows_128x128': ['conv_gather_bf16_rows_kernel<1, 2, 2'],
    'conv_fwd_bf16_rows_128x64': ['conv_gather_bf16_rows_kernel<0, 2, 1'], 'conv_dgrad_bf16_rows_128x64': ['conv_gather_bf16_rows_kernel<1, 2, 1'],
    'conv_fwd_bf16_64x64x6': _gbf(0, 2, 2, 1, 1), 'conv_dgrad_bf16_64x64x6': _gbf(1, 2, 2, 1, 1),
    'conv_fwd_bf16_c64': ['conv_gather_bf16_c64_kernel<0'], 'conv_dgrad_bf16_c64': ['conv_gather_bf16_c64_kernel<1'],
    'conv_fwd_bf16_rows_256x128': ['conv_gather_bf16_rows_kernel<0, 4'], 'conv_dgrad_bf16_rows_256x128': ['conv_gather_bf16_rows_kernel<1, 4'],
}


def pmc_traffic(label, dtype='f32', mode='train'):
    """HBM bytes per launch of `label`'s kernel from the newest committed rocprofv3 --pmc passes of that
    configuration (profiles/*_pmc_FETCH_SIZE.txt / *_pmc_WRITE_SIZE.txt; separate passes, KB units, FETCH_SIZE
    doubled on gfx950 as MI355X_MICROARCH.md prescribes).  None when no pass is committed."""
    import glob
    syms = KERNEL_SYMBOLS.get(label)

    def pick(n):
        n = os.path.basename(n)
        return ('bf16' in n) == (dtype == 'bf16') and ('decode' in n) == (mode == 'decode')
    f = sorted(n for n in glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_FETCH_SIZE.txt')) if pick(n))
    w = sorted(n for n in glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_WRITE_SIZE.txt')) if pick(n))
    if not syms or not f or not w:
        return None, None

    def avg(path):
        tot = cnt = 0.0
        for line in open(path):
            if any(sym in line for sym in syms):        # a label may cover several instantiations (e.g. a trailing flag)
                n = float(line.split('launches=')[1].split()[0])
                tot += float(line.split('avg=')[1].split()[0]) * n; cnt += n
        return tot / cnt if cnt else None
    fa, wa = avg(f[-1]), avg(w[-1])
    if fa is None or wa is None:
        return None, None
    return (2.0 * fa + wa) * 1024.0, os.path.basename(f[-1]) + ' + ' + os.path.basename(w[-1])


def synth_gt(rng, b):
    """n ~ U{1..5} boxes per image, w,h ~ U(0.1,0.6), inside the image."""
    boxes, cls, offs = [], [], [0]
    for _ in range(b):
        n = int(rng.integers(1, 6))
        w = rng.uniform(0.1, 0.6, n); h = rng.uniform(0.1, 0.6, n)
        boxes.append(np.stack([rng.uniform(w / 2, 1 - w / 2), rng.uniform(h / 2, 1 - h / 2), w, h], 1))
        cls.append(rng.integers(0, 20, n))
        offs.append(offs[-1] + n)
    return np.concatenate(boxes), np.concatenate(cls).astype(np.int32), np.array(offs, np.int32)


def cpu_baseline(preset, seconds_hint=15):
    """The fp32 CPU restatement (oracle/ssdvgg_ref.py: torch-CPU ops, every host core) on a bounded sample of the same workload: full
    steps (fwd + loss + bwd + update) at batch 2 -- `value` -- and, beside it, at batch 1 (BASELINE.json configs[0] names a single
    image; the larger of the two is what the host can do, so it is the reported baseline and the single-image figure is stated)."""
    import torch
    from oracle import boxes as ob, ssdvgg_ref as ref
    p = ob.get_preset(preset)
    rng = np.random.default_rng(1234)
    rates, notes = {}, {}
    for b in (2, 1):
        m = ref.RefModel(preset, params=ref.init_params(p, 20, seed=42))
        x, y, _ = ref.synth_batch(rng, b, p)
        m.train_step(x, y)                     # warm
        times = []
        t_all = time.perf_counter()
        budget = seconds_hint if b == 2 else seconds_hint / 2
        while len(times) < (5 if b == 2 else 3) or (time.perf_counter() - t_all < budget and len(times) < 9):
            t0 = time.perf_counter()
            m.train_step(x, y)
            times.append(time.perf_counter() - t0)
        med = float(np.median(times))
        rates[b] = round(b / med, 3)
        notes[b] = f'median of {len(times)} warm full training steps at batch {b}: {min(times):.2f}..{max(times):.2f} s per step'
    return dict(value=rates[2], unit='images/s', cores=torch.get_num_threads(), kind='port', value_batch1=rates[1],
                sample=f'{preset}, fp32 torch-CPU restatement oracle/ssdvgg_ref.py; value = {notes[2]}; value_batch1 (the single image of '
                       f'BASELINE configs[0]) = {notes[1]}')


def bench_augment(args, rank, world, local):
    """the GPU half of the train augmentation recipe for one batch: the two launches of
    ssd_augment_batch_dev on plans and source images already resident.  The host half (the transforms' decisions: a few
    milliseconds per image in Python, dominated by the sample-picker's trials) is NOT in this number: it is timed where it
    matters, in the end-to-end block (run_train_e2e), where worker processes run it beside the step.  CPU figure beside
    it: the numpy restatement (oracle/augment.py), one core."""
    import random
    import ctypes as C2
    import torch
    from ssd_tensorflow_amd._lib import lib, check
    from ssd_tensorflow_amd import transforms as T
    from ssd_tensorflow_amd.ssdutils import get_preset_by_name
    from ssd_tensorflow_amd.utils import Sample, Box, Point, Size
    preset = get_preset_by_name(args.preset)
    W, H = preset.image_size.w, preset.image_size.h
    nrng = np.random.default_rng(1234 + rank)
    random.seed(1234 + rank)
    b = args.batch
    plans, raw = [], []
    for i in range(b):
        w0, h0 = int(nrng.integers(300, 640)), int(nrng.integers(300, 640))       # VOC-like image sizes
        img = nrng.integers(0, 256, (h0, w0, 3)).astype(np.uint8)
        n = int(nrng.integers(1, 6))
        bw = nrng.uniform(0.1, 0.6, n); bh = nrng.uniform(0.1, 0.6, n)
        boxes = [Box('c', int(c), Point(float(x), float(y)), Size(float(ww), float(hh)))
                 for x, y, ww, hh, c in zip(nrng.uniform(bw / 2, 1 - bw / 2), nrng.uniform(bh / 2, 1 - bh / 2), bw, bh, nrng.integers(0, 20, n))]
        tfs = [t for t in T.build_train_transforms(preset, 20, 50, 0.5, images={'im': img}) if not isinstance(t, T.LabelCreatorTransform)]
        a = (None, None, Sample('im', boxes, Size(w0, h0)))
        for t in tfs:
            a = t(*a)
        plans.append(a[0]); raw.append((img, boxes, (w0, h0)))
    arr, packed = T.plan_params(plans, W, H)
    dev = torch.device('cuda', local)
    images = torch.from_numpy(packed).to(dev)
    out = torch.empty((b, H, W, 3), dtype=torch.float32, device=dev)
    ws = torch.empty((lib.ssd_augment_ws_bytes(b, W, H),), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        check(lib.ssd_augment_batch_dev(images.data_ptr(), C2.cast(arr, C2.c_void_p), b, W, H, out.data_ptr(), ws.data_ptr(), stream))
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gpu_ms = e0.elapsed_time(e1) / args.steps
    bytes_launch = out.numel() * 4 + packed.size
    ach = bytes_launch / (gpu_ms * 1e-3) / 1e9
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import augment as oa
        ncpu = min(b, 8)
        random.seed(1234 + rank)
        t1 = time.perf_counter()
        for i in range(ncpu):
            img, boxes, size = raw[i]
            p = oa.plan(oa.new_rng(99 + i), size, [(bb.center.x, bb.center.y, bb.size.w, bb.size.h) for bb in boxes], [bb.labelid for bb in boxes])
            oa.apply(p, img, (W, H))
        cpu = dict(value=round(ncpu / (time.perf_counter() - t1), 2), unit='images/s', cores=1, kind='port',
                   sample=f'{ncpu} images through oracle/augment.py (numpy restatement of the recipe; OpenCV is not installed)')
    if rank == 0:
        _OUT.emit(json.dumps({
            'metric': 'images/sec (train augmentation recipe -> %dx%d float32 batch) %s batch%d' % (W, H, args.preset, b),
            'value': round(b * args.steps / dt, 1), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'u8->f32', 'data': 'synthetic',
            'config': {'workload': 'process_dataset.py train recipe on %d synthetic uint8 images (300..640 px), plans and sources resident in HBM' % b},
            'roofline': {'bound': 'hbm', 'kernel': 'augment_gather', 'achieved': round(ach, 1), 'peak': PEAK_HBM, 'unit': 'GB/s',
                         'frac': round(ach / PEAK_HBM, 4), 'traffic': None, 'avg_launch_us': round(gpu_ms * 1e3, 2), 'bytes_per_launch': bytes_launch,
                         'measured': 'HIP events around %d back-to-back batches (taps + gather launches)' % args.steps},
            'cpu_baseline': cpu}))


EXPECT_PATH = os.path.join(ROOT, 'tests', 'golden', 'bench_expect.json')


def expected_losses(preset, batch, dtype):
    """Step-0 losses of the benchmark's own inputs (rank 0: images / boxes from default_rng(1234), library weights
    from seed 42) computed by the CPU oracle in the build container (tools/make_bench_expect.py); None if that
    configuration was not generated.  bf16 runs are checked against the fp32 values at 2e-3 (measured distance ~1e-4; the
    bf16 kernels themselves are checked layer by layer at these sizes by tests/test_gpu_bench_config.py)."""
    try:
        table = json.load(open(EXPECT_PATH))
    except OSError:
        return None, None
    e = table.get(f'{preset}_b{batch}')
    return (e, 1e-3 if dtype == 'f32' else 2e-3) if e else (None, None)


def self_check(out, serialized):
    """Cross-checks a block must pass before it is printed (a block that fails is replaced by its error, the headline
    aborts): the dominant kernel cannot take longer per step than the step; kernels measured one at a time (serialized
    pass) cannot sum to much more than the overlapped step that ran them side by side -- 1.3x: the overlap's measured gain
    is 2-17 % (bf16 training 1.16-1.17), while per-launch events taken INSIDE an overlapped region sum to ~2x (round 2's
    inference block) -- nor, measured inside a region that does not overlap (decode), to more than the step; no fraction
    above 1."""
    problems = []
    ms = out['ms_per_step']
    kms = out.get('kernel_ms_per_step') or {}
    ksum = out.get('kernel_ms_sum_per_step')
    r = out.get('roofline')
    if kms:
        dom, dom_ms = max(kms.items(), key=lambda kv: kv[1])
        if dom_ms > ms:
            problems.append(f'dominant kernel {dom} {dom_ms:.3f} ms/step > ms_per_step {ms:.3f}')
    if ksum is not None and ksum > (1.3 if serialized else 1.0) * ms:
        problems.append(f'kernel sum {ksum:.3f} ms/step > {"1.3 x " if serialized else ""}ms_per_step {ms:.3f}')
    if r is not None and not (0.0 < r['frac'] <= 1.0):
        problems.append(f'roofline.frac {r["frac"]} outside (0, 1]')
    # model_mfma_frac prices the step by the ALGORITHMIC (direct-convolution) FLOPs of SURVEY.md 8d; the fp32 step runs its 3x3 trunk
    # layers in the Winograd F(4x4, 3x3) form (csrc/winograd.hip: 4x fewer multiplies), so that figure may exceed 1 there and only
    # there -- what cannot exceed 1 is the fraction of the peak the matrix pipe was actually asked for (executed_mfma_frac)
    wino = bool(out.get('algorithm'))
    if out.get('model_mfma_frac') is not None and not (0.0 < out['model_mfma_frac'] <= (4.0 if wino else 1.0)):
        problems.append(f'model_mfma_frac {out["model_mfma_frac"]} outside (0, {4 if wino else 1}]')
    if out.get('executed_mfma_frac') is not None and not (0.0 < out['executed_mfma_frac'] <= 1.0):
        problems.append(f'executed_mfma_frac {out["executed_mfma_frac"]} outside (0, 1]')
    if problems:
        raise SystemExit('[bench] self-check failed: ' + '; '.join(problems))
    return 'dominant kernel <= step, kernel sum <= %sstep, fractions in (0, 1]' % ('1.3 x ' if serialized else '')


def run_config(a, rank, world, local):
    """One benchmark configuration -> the JSON-able result dict (rank 0; None elsewhere).
    a: namespace with mode, preset, batch, dtype, steps, warmup, bucket_mb, no_overlap, per_layer, no_kernel_events,
    no_cpu_baseline, zero_input, allow_fallback."""
    import torch
    import torch.distributed as dist
    from ssd_tensorflow_amd._lib import lib, check
    from ssd_tensorflow_amd.ssdvgg import SSDVGG, Session
    from ssd_tensorflow_amd import parallel

    b = a.batch
    bucket = int(a.bucket_mb * 1e6 / 4)
    state = {'bucket': bucket}
    sess = Session(local)
    net = SSDVGG(sess, a.preset)
    training = a.mode == 'train'
    net.build_from_vgg(None, 20, max_batch=b, training=training, seed=42, dtype=a.dtype)
    peak_mfma = PEAK_BF16_MFMA if a.dtype == 'bf16' else PEAK_FP32_MFMA
    if world > 1:      # identical replicas
        dist.broadcast(net.params_flat, 0)
    if training:
        net.build_optimizer(learning_rate=0.00075, weight_decay=0.0005, momentum=0.9)
    net.set_stream(torch.cuda.current_stream().cuda_stream)

    # synthetic shard, generated on this rank, resident in HBM before the timed region
    rng = np.random.default_rng(1234 + rank)
    H, W = net.preset.image_size.h, net.preset.image_size.w
    x = torch.from_numpy(rng.integers(0, 256, (b, H, W, 3)).astype(np.float32)).cuda()
    if a.zero_input:
        x.zero_()
    A, nv = net.preset.num_anchors, 25
    y = torch.empty((b, A, nv), dtype=torch.float32, device='cuda')
    gt, cls, offs = synth_gt(rng, b)
    check(lib.ssd_encode_labels_dev(a.preset.encode(), 20, local, gt.ctypes.data, cls.ctypes.data, offs.ctypes.data, b,
                                    y.data_ptr(), None))
    torch.cuda.synchronize()

    # parity guard of the benchmarked configuration itself: the step-0 losses of these very inputs must be the CPU
    # oracle's (stored by tools/make_bench_expect.py); a mismatch makes the run invalid, so it is fatal
    losses_check = None
    if training and rank == 0 and not a.zero_input:
        want, tol = expected_losses(a.preset, b, a.dtype)
        if want is not None:
            net.eval_step_dev(x, y)
            got = net.get_losses()
            worst = max(abs(got[k] - want[k]) / abs(want[k]) for k in want)
            losses_check = dict(step0=got, oracle=want, max_rel_err=worst, tol=tol, ok=bool(worst < tol))
            if not losses_check['ok']:
                raise SystemExit(f'[bench] step-0 losses differ from the oracle: {got} vs {want} (rel {worst:.3e} >= {tol})')

    if a.mode == 'decode':
        # BASELINE config 5 / SURVEY 8d: softmax of N(0,1) logits, +4 on background, +8 on 300 random
        # (anchor, class) pairs per image; loc ~ N(0, 0.5): ~300 detections per image at thr 0.5
        g = torch.Generator(device='cuda'); g.manual_seed(1234 + rank)
        logits = torch.randn((b, A, 21), generator=g, device='cuda')
        logits[:, :, 20] += 4
        hot_a = torch.randint(0, A, (b, 300), generator=g, device='cuda'); hot_c = torch.randint(0, 20, (b, 300), generator=g, device='cuda')
        logits[torch.arange(b, device='cuda')[:, None], hot_a, hot_c] += 8
        pred = torch.cat([torch.softmax(logits, -1), torch.randn((b, A, 4), generator=g, device='cuda') * 0.5], -1).contiguous()
        check(lib.ssd_set_result_dev(net._h, pred.data_ptr(), b))
        torch.cuda.synchronize()

    pend = {'t': None}

    def step():
        if a.mode == 'decode':
            # the inference loop's pattern (infer.py): launch batch k, collect batch k-1 (its small output has
            # already been copied to pinned host memory behind the kernels)
            t = net.detect_last_launch(b, 0.5, None, 200)
            if pend['t'] is not None:
                pend['t'].get()
            pend['t'] = t
            return
        if a.mode == 'train':
            # N > 1: bucketed all-reduce (sum over ranks, RCCL over xGMI) overlapped with backward
            parallel.train_step_dp(net, x, y, world, state['bucket'], force_collectives=getattr(a, 'force_collectives', False),
                                   allreduce_dtype=getattr(a, 'allreduce_dtype', 'f32'))
        elif a.mode == 'infer':
            net.infer_dev(x)
        else:
            net.infer_dev(x)
            net.detect_last(b, 0.5, None, 200)

    def drain():
        if pend['t'] is not None:
            pend['t'].get(); pend['t'] = None

    if a.no_overlap and a.mode in ('train', 'infer', 'detect'):
        check(lib.ssd_set_overlap(net._h, 0))
    allreduce_mode = 'none' if world == 1 else ('bucketed, overlapped with backward' if bucket > 0 else 'single, after backward')
    if world == 1 and getattr(a, 'force_collectives', False):
        allreduce_mode = ('bucketed, overlapped with backward' if bucket > 0 else 'single, after backward') + ' (single-rank group: the data-parallel plumbing alone)'
    if world > 1 and a.mode == 'train' and bucket > 0:
        # the overlapped path is exercised once up front.  A failure is fatal: a scaling line produced by a
        # silently downgraded collective would read like a measurement of the shipped path.  --allow-fallback
        # downgrades to ONE all-reduce after backward instead (and says so in config.allreduce).
        try:
            step()
            torch.cuda.synchronize()
        except Exception as e:      # noqa: BLE001
            if not a.allow_fallback:
                raise
            if rank == 0:
                print(f'[bench] bucketed all-reduce failed ({type(e).__name__}: {e}); falling back to a single all-reduce', file=sys.stderr)
            state['bucket'] = 0
            allreduce_mode = 'single, after backward (FALLBACK: the bucketed path failed)'
    use_events = not a.no_kernel_events
    # training AND inference run kernels side by side (weight-gradient stream, two forward lanes, heads on a side
    # stream): per-launch events inside such a region do not measure one kernel
    overlapped = a.mode in ('train', 'infer', 'detect') and not a.no_overlap
    events_in_timed_region = use_events and not overlapped
    if events_in_timed_region:
        # the per-launch events are created on first use (a fresh process pays ~0.5 ms for each): the warmup steps run
        # with the profiler on so that the timed region re-uses its pool
        check(lib.ssd_profile_enable(net._h, 1))
    t_w = time.perf_counter()
    warm_run = 0
    for _ in range(a.warmup):
        step(); warm_run += 1
    # A pass of a few tens of microseconds per step does not wake the GPU's clocks: a fresh process measured a fixed
    # ~40 ms of low-clock execution inside the timed region of the decode configuration (2.1 ms/step over 20 steps,
    # 0.36 over 200).  The sub-millisecond modes therefore keep warming up until 0.3 s of wall time have passed; the
    # number of warmup steps actually run is reported.  (Training and inference steps are untouched: exactly W.)
    if a.mode in ('decode', 'detect'):
        while time.perf_counter() - t_w < 0.3:
            step(); warm_run += 1
    drain()
    if events_in_timed_region:
        torch.cuda.synchronize()
        check(lib.ssd_profile_report(net._h, C.create_string_buffer(1 << 16), 1 << 16))      # discard, keep the pool
    # In training the weight gradients run on a side stream next to the data gradients, so kernels of
    # the timed region overlap and a per-launch event interval is not one kernel's own duration.  The
    # timed region therefore runs WITHOUT per-launch events; the roofline block comes from an equal
    # number of serialized steps (one kernel at a time, events on the launching stream) right after it.
    serialize_for_events = use_events and overlapped
    if use_events and not serialize_for_events:
        check(lib.ssd_profile_enable(net._h, 2 if a.per_layer else 1))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t_issued = time.perf_counter() - t0      # host time to ISSUE the steps (no sync inside): well below dt = the GPU is the bound
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    replicas_agree = None      # N > 1: do the replicas still hold identical parameters after the timed steps?
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        chk = net.params_flat[::4099].double().sum().reshape(1)
        lo = chk.clone(); hi = chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_agree = float(hi - lo) == 0.0
        if not replicas_agree and a.mode == 'train':
            msg = f'[bench] replicas diverged after {a.steps} data-parallel steps (checksum spread {float(hi - lo):.3e})'
            if not a.allow_fallback:
                raise SystemExit(msg)
            if rank == 0:
                print(msg + ' -- reported, not fatal (--allow-fallback)', file=sys.stderr)
        dt = float(t.item())

    roofline = None
    kernels = {}
    if serialize_for_events:
        check(lib.ssd_set_overlap(net._h, 0))
        check(lib.ssd_profile_enable(net._h, 2 if a.per_layer else 1))
        torch.cuda.synchronize()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
    if use_events:
        buf = C.create_string_buffer(1 << 16)
        check(lib.ssd_profile_report(net._h, buf, len(buf)))
        for line in buf.value.decode().strip().split('\n'):
            if not line:
                continue
            k, cnt, ms, fl, by = line.split('\t')
            if a.per_layer:
                if rank == 0:
                    tf = float(fl) / (float(ms) * 1e-3) / 1e12 if float(ms) > 0 else 0
                    print(f'{k:<48s} n={int(cnt) // a.steps:3d}  {float(ms) / a.steps:8.3f} ms/step  {tf:7.1f} TF/s  '
                          f'{float(by) / (float(ms) * 1e-3) / 1e9 if float(ms) > 0 else 0:8.1f} GB/s', file=sys.stderr)
                k = k.split(':')[0]
            d0 = kernels.setdefault(k, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d0['launches'] += int(cnt); d0['ms'] += float(ms); d0['flops'] += float(fl); d0['bytes'] += float(by)
        check(lib.ssd_profile_enable(net._h, 0))
        if kernels:
            dom = max(kernels, key=lambda k: kernels[k]['ms'])      # dominant = most time, whatever it is
            d = kernels[dom]
            if d['flops'] > 0:
                ach = d['flops'] / (d['ms'] * 1e-3) / 1e12
                roofline = dict(bound='mfma', kernel=dom, achieved=round(ach, 2), peak=peak_mfma, unit='TFLOP/s',
                                frac=round(ach / peak_mfma, 4), traffic=None,
                                launches_per_step=d['launches'] // a.steps,
                                avg_launch_us=round(d['ms'] * 1e3 / d['launches'], 2),
                                flops_per_launch=d['flops'] / d['launches'],
                                measured='HIP events per launch, %d serialized steps after the timed region' % a.steps
                                if serialize_for_events else 'HIP events per launch over the timed region')
            else:
                # an HBM-bound pass: algorithmic bytes of the WHOLE pass (every [A, C+5] row once for
                # decode + NMS) over the summed duration of all its kernels -- the candidate kernels move few bytes
                # but take time, so pricing one hand-picked kernel would flatter the pass
                if a.mode == 'decode':
                    by = sum(v['bytes'] for v in kernels.values()); ms = sum(v['ms'] for v in kernels.values())
                    name = ' + '.join(sorted(kernels, key=lambda k: -kernels[k]['ms']))
                    launches = kernels[dom]['launches']
                else:
                    by, ms, name, launches = d['bytes'], d['ms'], dom, d['launches']
                ach = by / (ms * 1e-3) / 1e9
                roofline = dict(bound='hbm', kernel=name, achieved=round(ach, 1), peak=PEAK_HBM, unit='GB/s',
                                frac=round(ach / PEAK_HBM, 4), traffic=None,
                                launches_per_step=launches // a.steps,
                                avg_launch_us=round(ms * 1e3 / launches, 2),
                                bytes_per_launch=by / launches, dominant_by_time=dom)

    out = None
    if rank == 0:
        imgs = b * world * a.steps
        value = imgs / dt
        flops_img = FLOPS_FWD_BWD[a.preset] if a.mode == 'train' else FLOPS_FWD[a.preset]
        cfg_no = {('train', 'vgg300', 'f32'): 1, ('train', 'vgg300', 'bf16'): 2, ('train', 'vgg512', 'f32'): 3,
                  ('train', 'vgg512', 'bf16'): 3, ('decode', 'vgg300', 'f32'): 4, ('detect', 'vgg300', 'f32'): 4}.get((a.mode, a.preset, a.dtype))
        what = {'train': 'training step (forward + multibox loss + backward + momentum update)', 'infer': 'inference forward',
                'detect': 'inference forward + decode + per-class NMS', 'decode': 'decode + per-class NMS of resident predictions'}[a.mode]
        out = {
            'metric': 'images/sec (fwd+bwd) %s batch%d' % (a.preset, b) if a.mode == 'train'
                      else 'images/sec (decode + per-class NMS of [b,A,25] predictions) %s batch%d' % (a.preset, b) if a.mode == 'decode'
                      else 'images/sec (%s) %s batch%d' % (a.mode, a.preset, b),
            'value': round(value, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'warmup_steps_run': warm_run,
            'ms_per_step': round(dt / a.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'host_issue_ms_per_step': round(t_issued / a.steps * 1e3, 4),
            'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
            'config': {'workload': f'{a.preset} {what}, {b} images/GPU x {world} GPU, synthetic {H}x{W} BGR 0..255, '
                                   + ('GPU-encoded labels, Xavier-init weights' if a.mode != 'decode' else 'synthetic predictions (~300 detections/image at 0.5), outputs collected on the host')
                                   + (f' (BASELINE.json configs[{cfg_no}]' + (', per-GPU share' if cfg_no in (2, 3) and world == 1 else '') + ')' if cfg_no is not None else ' (not a BASELINE.json config)'),
                       'global_batch': b * world, 'parallelism': f'dp{world}',
                       'allreduce': allreduce_mode + (', bf16 messages' if allreduce_mode != 'none' and getattr(a, 'allreduce_dtype', 'f32') == 'bf16' else ''),
                       'replicas_agree': replicas_agree},
            'ms_per_image': round(dt / a.steps / (b * world) * 1e3, 5),
            'model_tflops': round(value * flops_img / 1e12, 2) if a.mode != 'decode' else None,
            'model_mfma_frac': round(value * flops_img / 1e12 / (peak_mfma * world), 4) if a.mode != 'decode' else None,
            'roofline': roofline,
        }
        if roofline is not None and world == 1:
            if a.mode == 'decode':      # the whole pass is priced: traffic of all its kernels
                parts = [pmc_traffic(k, a.dtype, a.mode) for k in kernels]
                tr, src = (sum(p[0] for p in parts), parts[0][1]) if parts and all(p[0] is not None for p in parts) else (None, None)
            else:
                tr, src = pmc_traffic(roofline['kernel'], a.dtype, a.mode)
            if tr is not None:
                roofline['traffic'] = tr
                roofline['traffic_source'] = src
                # NOT counted in this run: rocprofv3 --pmc cannot wrap a run from inside it; the counters come from the
                # newest committed passes over the same command (tools/profile_round.sh), whose kernels are the same binary
                roofline['traffic_measured_in'] = 'builder profile (committed rocprofv3 --pmc passes under profiles/), not this run'
        if use_events:
            tot = sum(k['ms'] for k in kernels.values())
            out['kernel_ms_per_step'] = {k: round(v['ms'] / a.steps, 4) for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]['ms'])}
            out['kernel_ms_sum_per_step'] = round(tot / a.steps, 4)
            if a.mode != 'decode':
                # the FLOPs the launches actually executed (each launcher's own count: the Winograd GEMMs count their 36 small
                # products, not the direct convolution they replace) over the timed step
                ex = sum(k['flops'] for k in kernels.values()) / a.steps
                wl = sorted(k for k in kernels if k.startswith('wino_gemm'))
                out['executed_tflops'] = round(ex / (dt / a.steps) / 1e12 / world, 2)
                out['executed_mfma_frac'] = round(ex / (dt / a.steps) / 1e12 / peak_mfma, 4)
                if wl:
                    nl = sum(kernels[k]['launches'] for k in wl) // a.steps
                    out['algorithm'] = ('3x3 / stride 1 trunk layers in the Winograd F(4x4, 3x3) form (%d GEMM launches per step, csrc/winograd.hip); '
                                        'model_tflops / model_mfma_frac price the step by the direct-convolution FLOPs of SURVEY.md 8d '
                                        '(%.2f GFLOP per image), executed_* by the FLOPs the kernels ran' % (nl, flops_img / 1e9))
        if world == 1 and not a.no_cpu_baseline and a.mode == 'train':
            out['cpu_baseline'] = cpu_baseline(a.preset)
        else:
            out['cpu_baseline'] = None
        if a.mode == 'train':
            out['losses_last_step'] = net.get_losses()
            out['losses_check'] = losses_check
        out['self_check'] = self_check(out, serialize_for_events)
    sess.close()
    del net, x, y
    torch.cuda.empty_cache()
    return out


def run_train_e2e(a, rank, world, local):
    """The training driver's own loop (ssd_tensorflow_amd/train.py StepLoop = the reference's train.py:254-281) timed end
    to end on one GPU: the feeder (synthetic uint8 "files" -> the reference's augmentation recipe decided by `workers`
    forked processes -> upload -> augmentation + label kernels into the device slot ring), the training step, the loss
    fetch one step late, decode + NMS of every batch and the collection of its detections for the AP bookkeeping.
    Nothing is resident before the timed region except the dataset's source bytes in host RAM (a real source reads
    files).  One untimed epoch first (workers fork, anchors / caches warm), then `epochs` timed ones."""
    import torch
    from ssd_tensorflow_amd.ssdvgg import SSDVGG, Session
    from ssd_tensorflow_amd.training_data import TrainingData
    from ssd_tensorflow_amd.train import StepLoop
    from ssd_tensorflow_amd.average_precision import APCalculator
    from ssd_tensorflow_amd.summaries import LossSummary
    b = a.batch
    n_samples = b * a.e2e_steps
    t0 = time.perf_counter()
    # With --e2e-checkpoint the loop runs on the learnable 'shapes' set from weights trained on it (bench.py --mode pretrain): the
    # decode + NMS pass of every batch then HAS detections to emit and the host their boxes to collect, as the reference's loop
    # does from its second epoch on (train.py:273-281).  From Xavier weights no class ever passes 0.5 and the collection is free.
    ckpt = getattr(a, 'e2e_checkpoint', None) or None
    td = TrainingData('shapes' if ckpt else None, a.preset, num_train=n_samples, num_valid=b, augment=True, device=local, seed=1234 + rank)
    t_data = time.perf_counter() - t0
    sess = Session(local)
    net = SSDVGG(sess, a.preset)
    if ckpt:
        net.build_from_metagraph(None, ckpt, max_batch=b, training=True, dtype=a.dtype)
        net.build_optimizer_from_metagraph()
    else:
        net.build_from_vgg(None, 20, max_batch=b, training=True, seed=42, dtype=a.dtype)
        net.build_optimizer(learning_rate=0.00075, weight_decay=0.0005, momentum=0.9)
    net.set_stream(torch.cuda.current_stream().cuda_stream)
    res = {}
    try:
        for label, workers, epochs in (('prefetched', a.e2e_workers, a.e2e_epochs), ('serial', 0, 1)):
            if label == 'serial' and a.e2e_serial_steps <= 0:
                continue
            loop = StepLoop(net, sess, td, b, workers)
            calc = APCalculator(); summ = LossSummary(None, 'training', n_samples)
            td.epoch = 0
            if label == 'prefetched':
                loop.run_epoch(td.train_generator, True, summ, calc, True)          # untimed: forks the workers, warms everything
            else:
                td.num_train = b * a.e2e_serial_steps                                 # the serial feeder is slow: fewer steps
                td._recipes['train'].total = td.num_train
            calc.clear()
            summ = LossSummary(None, 'training', td.num_train * epochs)
            torch.cuda.synchronize()
            steps0 = loop.steps
            t0 = time.perf_counter()
            for e in range(epochs):
                td.epoch = 1 + e
                loop.run_epoch(td.train_generator, True, summ, calc, True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            steps = loop.steps - steps0
            res[label] = dict(value=round(steps * b / dt, 2), ms_per_step=round(dt / steps * 1e3, 4), steps=steps, epochs=epochs, workers=workers,
                              mean_losses=summ.push(0), detections_collected=len(calc.det_confidence))
            if workers:      # last epoch's account of the feeder: who waited for whom (ms per step)
                st = td.feeder_stats
                res[label]['feeder_ms_per_step'] = {k: round(v / max(st['batches'], 1) * 1e3, 3) for k, v in st.items() if k != 'batches'}
    finally:
        td.close(); sess.close()
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    r = res['prefetched']
    return {'metric': 'images/sec end to end (feeder + augmentation + fwd+bwd + loss fetch + decode/NMS collection) %s batch%d' % (a.preset, b),
            'value': r['value'], 'unit': 'images/s', 'ms_per_step': r['ms_per_step'], 'steps': r['steps'], 'epochs': r['epochs'], 'dtype': a.dtype,
            'feeder_workers': r['workers'], 'host_cores': os.cpu_count(), 'dataset_build_s': round(t_data, 2),
            'serial_feeder': res.get('serial'), 'mean_losses': r['mean_losses'], 'detections_collected': r['detections_collected'],
            'feeder_ms_per_step': r.get('feeder_ms_per_step'),
            'weights': ('trained %s (bench.py --mode pretrain on the shapes set)' % os.path.basename(ckpt)) if ckpt else 'Xavier (no detections above 0.5)',
            'config': {'workload': f"{a.preset} train.py StepLoop, {b} images/step, {n_samples} synthetic uint8 images of 200..640 px "
                                   f"({'textured rectangles' if ckpt else 'uniform noise'}) through the reference's "
                                   'train recipe (process_dataset.py:66-140), --num-workers %d' % r['workers']}}


def ordered_for_tail(out):
    """The line is long (a dozen blocks) and a reader that keeps only its tail should still see what matters: the per-kernel
    table goes first, the secondary blocks next, the contract's own keys after them and a compact `summary` of every block
    last.  (Key order means nothing to a JSON parser.)"""
    bulky = ('kernel_ms_per_step',)
    blocks = [k for k, v in out.items() if isinstance(v, dict) and k not in ('config', 'roofline', 'cpu_baseline', 'losses_check', 'losses_last_step') + bulky]
    o = {k: out[k] for k in bulky if k in out}
    o.update({k: out[k] for k in blocks if k != 'bf16'})
    if 'bf16' in out:
        o['bf16'] = out['bf16']
    o.update({k: v for k, v in out.items() if k not in o})
    summ = {}
    for k in blocks:
        v = out[k]
        if 'error' in v:
            summ[k] = 'error'
            continue
        e = {}
        for f in ('value', 'unit', 'ms_per_step', 'ms_per_image', 'model_mfma_frac', 'vs_resident_input', 'detections_collected', 'final_map_training'):
            if v.get(f) is not None:
                e[f] = v[f]
        if isinstance(v.get('roofline'), dict):
            e['roofline_frac'] = v['roofline'].get('frac')
        if isinstance(v.get('losses_check'), dict):
            e['losses_ok'] = v['losses_check'].get('ok')
        summ[k] = e
    if 'e2e_workload' in out:
        summ['e2e_workload'] = out['e2e_workload']
    if summ:
        o['summary'] = summ
    return o


# configurations BASELINE.json names beside the headline; timed by the default invocation with a few steps each so
# that they are driver-run numbers (the headline `value` stays configs[1])
SECONDARY = [
    # (bf16 blocks: 20 steps -- five 8-ms steps right after a context switch of configurations measured 3-5 % low)
    ('bf16', dict(mode='train', preset='vgg300', batch=32, dtype='bf16', steps=20, warmup=5)),
    ('vgg512_b16', dict(mode='train', preset='vgg512', batch=16, dtype='f32', steps=5, warmup=2)),
    ('vgg512_b16_bf16', dict(mode='train', preset='vgg512', batch=16, dtype='bf16', steps=20, warmup=5)),
    ('infer_b128', dict(mode='infer', preset='vgg300', batch=128, dtype='f32', steps=5, warmup=2)),
    ('infer_b128_bf16', dict(mode='infer', preset='vgg300', batch=128, dtype='bf16', steps=20, warmup=5)),
    ('decode_b128', dict(mode='decode', preset='vgg300', batch=128, dtype='f32', steps=20, warmup=3)),
]


_OUT = None


class _QuietStdout:
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner when a
    communicator is created): while the benchmark runs, file descriptor 1 points at stderr; `emit` restores it for the line."""

    def __init__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def emit(self, text):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        print(text, flush=True)
        os.dup2(2, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--preset', default='vgg300')
    ap.add_argument('--batch', type=int, default=32, help='images per GPU')
    ap.add_argument('--mode', default='train', choices=['train', 'infer', 'detect', 'decode', 'augment', 'train_e2e', 'pretrain'])
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help="f32 = BASELINE.json configs[1] (the headline); bf16 = configs[2]'s per-GPU step (bf16 MFMA, fp32 masters)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary configurations (bf16, vgg512, infer, decode blocks)')
    ap.add_argument('--zero-input', action='store_true', help='DVFS probe: all-zero images (NOT a valid benchmark number)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend for N > 1 (nccl = RCCL); gloo only for plumbing tests')
    ap.add_argument('--same-device', action='store_true', help='plumbing test: every rank uses GPU 0 (gloo only)')
    ap.add_argument('--null-collective', action='store_true', help='with --force-collectives on one GPU: the all-reduce is an event hand-off only (prices the plumbing without the single-rank RCCL kernels)')
    ap.add_argument('--force-collectives', action='store_true', default=os.environ.get('SSD_BENCH_FORCE_COLLECTIVES', '0') == '1',
                    help='one GPU: run the data-parallel step (staged backward + bucketed all-reduce on a single-rank RCCL group) to price its plumbing')
    ap.add_argument('--bucket-mb', type=float, default=float(os.environ.get('SSD_BENCH_BUCKET_MB', 44)), help='all-reduce finished gradient ranges of >= this many MB while backward still runs (0 = one all-reduce after backward)')
    ap.add_argument('--allreduce-dtype', default=os.environ.get('SSD_BENCH_ALLREDUCE_DTYPE', 'f32'), choices=['f32', 'bf16'],
                    help='N > 1 / --force-collectives: bf16 = filter gradients all-reduced as bf16 messages (half the bytes over xGMI)')
    ap.add_argument('--allow-fallback', action='store_true', help='N > 1: downgrade a failing bucketed all-reduce to a single one / report diverged replicas instead of aborting')
    ap.add_argument('--no-overlap', action='store_true', help='training: weight gradients on the main stream (one kernel at a time)')
    ap.add_argument('--per-layer', action='store_true', help='per-layer kernel table on stderr (events labelled kernel:layer)')
    ap.add_argument('--no-kernel-events', action='store_true', help='skip per-launch HIP events (roofline block = null)')
    ap.add_argument('--e2e-workers', type=int, default=0, help='train_e2e: feeder worker processes (0 = 8)')
    ap.add_argument('--e2e-steps', type=int, default=0, help='train_e2e: steps per epoch (0 = 20 fp32 / 60 bf16)')
    ap.add_argument('--e2e-epochs', type=int, default=1, help='train_e2e: timed epochs')
    ap.add_argument('--e2e-serial-steps', type=int, default=3, help='train_e2e: steps of the serial-feeder comparison (0 = skip)')
    ap.add_argument('--e2e-both', action='store_true', help='train_e2e: run both dtypes in this process')
    ap.add_argument('--no-e2e', action='store_true', help='skip the end-to-end blocks of the default invocation')
    ap.add_argument('--e2e-checkpoint', default='', help='train_e2e: start from this checkpoint and feed the learnable shapes set (detections to collect)')
    ap.add_argument('--pretrain-dir', default='', help='pretrain: directory that receives final.npz')
    ap.add_argument('--pretrain-epochs', type=int, default=40, help='pretrain: epochs of 1024 shapes images at batch 32 (bf16)')
    ap.add_argument('--no-pretrain', action='store_true', help='default invocation: end-to-end blocks from Xavier weights (no detections)')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ssd_tensorflow_amd import _lib

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    global _OUT
    _OUT = _QuietStdout()
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit(f'--gpus {args.gpus} needs torch.distributed.run --nproc-per-node {args.gpus}')
    if args.same_device:
        local = 0
    if world > 1 or args.force_collectives:
        from ssd_tensorflow_amd import parallel
        parallel.reserve_hw_queues()      # (before the first HIP call: an RCCL communicator's streams need queues of their own)
    torch.cuda.set_device(local)
    _lib.set_device(local)
    if world > 1 or args.force_collectives:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world == 1:
            os.environ.setdefault('MASTER_PORT', '29533'); os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(args.backend)

    if args.null_collective:
        # Measurement aid for --force-collectives on ONE GPU: RCCL's single-rank all-reduce is not free -- rocprofv3 shows 20
        # `__amd_rocclr_copyBuffer` launches per call (profiles/r05_e_dp_single_*_kernel_stats.csv) that a real ring never runs.
        # With this flag the collective is an event hand-off on a side stream and nothing else, so what remains is THIS
        # package's plumbing: staged backward, the cross-stream events, the update after the last bucket.
        assert world == 1, '--null-collective is a one-GPU measurement aid'
        _cs = torch.cuda.Stream()

        class _Work:
            def __init__(self, ev):
                self.ev = ev

            def wait(self):
                torch.cuda.current_stream().wait_event(self.ev)

        def _null_all_reduce(t, op=None, async_op=False):
            _cs.wait_stream(torch.cuda.current_stream())
            ev = torch.cuda.Event()
            ev.record(_cs)
            w = _Work(ev)
            if async_op:
                return w
            w.wait()
            return None
        dist.all_reduce = _null_all_reduce

    if args.mode == 'augment':
        return bench_augment(args, rank, world, local)
    if args.mode == 'pretrain':
        # trains a detector on the learnable shapes set with the product's own driver (the schedule of tests/test_gpu_learning.py:
        # bf16, 1280 steps) so that the end-to-end blocks have detections to decode and collect; prints {checkpoint, final mAP}
        import contextlib, io, re
        from ssd_tensorflow_amd import train
        buf = io.StringIO()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(buf):
            rc = train.main(['--name', args.pretrain_dir, '--tensorboard-dir', os.path.join(args.pretrain_dir, 'tb'), '--data-dir', 'shapes',
                             '--synthetic-train', '1024', '--synthetic-valid', '128', '--num-workers', '8', '--batch-size', '32',
                             '--checkpoint-interval', '1000', '--lr-values', '0.0003;0.00075;0.0001', '--lr-boundaries', '96;768',
                             '--epochs', str(args.pretrain_epochs), '--dtype', 'bf16', '--augment', 'false'])
        maps = re.findall(r'mAP +\d+/\d+ +training ([\d.]+) +validation ([\d.]+)', buf.getvalue())
        _OUT.emit(json.dumps({'rc': rc, 'checkpoint': os.path.join(args.pretrain_dir, 'final.npz'), 'seconds': round(time.perf_counter() - t0, 1),
                              'steps': args.pretrain_epochs * 32, 'final_map_training': float(maps[-1][0]) if maps else None,
                              'final_map_validation': float(maps[-1][1]) if maps else None}))
        return

    def e2e(dtype):
        sub = argparse.Namespace(**vars(args))
        sub.dtype = dtype
        sub.e2e_workers = args.e2e_workers or 8
        sub.e2e_steps = args.e2e_steps or (60 if dtype == 'bf16' else 20)
        return run_train_e2e(sub, rank, world, local)

    if args.mode == 'train_e2e':
        r = e2e(args.dtype)
        if args.e2e_both:      # both dtypes in one process, as the default invocation runs them
            r = {'f32' if args.dtype == 'f32' else 'bf16': r}
            other = 'bf16' if args.dtype == 'f32' else 'f32'
            r[other] = e2e(other)
        if rank == 0:
            _OUT.emit(json.dumps(r))
        return
    out = run_config(args, rank, world, local)
    headline = (args.mode, args.preset, args.batch, args.dtype) == ('train', 'vgg300', 32, 'f32')
    if headline and world == 1 and not args.no_secondary and not args.zero_input and not args.per_layer:
        for name, cfg in SECONDARY:
            sub = argparse.Namespace(**{**vars(args), **cfg, 'no_cpu_baseline': True, 'per_layer': False})
            try:
                r = run_config(sub, rank, world, local)
                out[name] = {k: r[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'host_issue_ms_per_step', 'steps', 'warmup', 'dtype', 'config',
                                               'model_tflops', 'model_mfma_frac', 'executed_tflops', 'executed_mfma_frac', 'algorithm', 'roofline', 'kernel_ms_sum_per_step', 'losses_check', 'self_check', 'ms_per_image')
                             if k in r}
            except (Exception, SystemExit) as e:      # noqa: BLE001 -- a secondary block never costs the headline line
                out[name] = {'error': f'{type(e).__name__}: {e}'}
        if not args.no_e2e:
            # The driver loop end to end, beside the resident-input numbers above (same box).  Each block runs in a FRESH
            # process, as train.py would: which streams end up sharing a hardware queue depends on how many streams the
            # process has created and destroyed before, and after eight configurations in this process the
            # feeder's stream measured 0.87 of the resident-input bf16 step instead of the 0.94-0.95 of a fresh process.
            import subprocess
            import tempfile

            def child(cmd, what):
                r = subprocess.run([sys.executable, os.path.abspath(__file__)] + cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
                lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
                if r.returncode != 0 or not lines:
                    raise RuntimeError('%s child failed (rc %d): %s' % (what, r.returncode, r.stderr[-400:]))
                return json.loads(lines[-1])
            tmpdir = tempfile.mkdtemp(prefix='ssd_bench_')
            ckpt = ''
            if not args.no_pretrain:
                try:
                    out['e2e_pretrain'] = child(['--mode', 'pretrain', '--pretrain-dir', tmpdir, '--pretrain-epochs', str(args.pretrain_epochs)], 'pretrain')
                    if out['e2e_pretrain'].get('rc') == 0 and os.path.exists(out['e2e_pretrain']['checkpoint']):
                        ckpt = out['e2e_pretrain']['checkpoint']
                except (Exception, SystemExit) as e:      # noqa: BLE001 -- the end-to-end blocks then run from Xavier weights
                    out['e2e_pretrain'] = {'error': f'{type(e).__name__}: {e}'}

            def e2e_child(dtype):
                return child(['--mode', 'train_e2e', '--dtype', dtype, '--preset', args.preset,
                              '--batch', str(args.batch), '--e2e-workers', str(args.e2e_workers), '--e2e-steps', str(args.e2e_steps),
                              '--e2e-epochs', str(args.e2e_epochs), '--e2e-serial-steps', str(args.e2e_serial_steps)]
                             + (['--e2e-checkpoint', ckpt] if ckpt and args.preset == 'vgg300' else []), 'train_e2e')
            for name, dtype, resident in (('train_e2e', 'f32', out), ('train_e2e_bf16', 'bf16', out.get('bf16'))):
                try:
                    r = e2e_child(dtype)
                    if resident and 'value' in resident:
                        r['resident_input_value'] = resident['value']
                        r['vs_resident_input'] = round(r['value'] / resident['value'], 4)
                    out[name] = r
                except (Exception, SystemExit) as e:      # noqa: BLE001
                    out[name] = {'error': f'{type(e).__name__}: {e}'}
            import shutil
            shutil.rmtree(tmpdir, ignore_errors=True)
            # which workload the end-to-end blocks ran on: never compare their numbers across workloads (round 4 advice)
            out['e2e_workload'] = ('shapes set, weights pretrained in this run (%s steps): detections decoded and collected' % out['e2e_pretrain'].get('steps')) \
                if ckpt else 'noise set, Xavier weights: no class passes 0.5, nothing to collect (--no-pretrain, or the pretrain child failed)'
    if rank == 0:
        _OUT.emit(json.dumps(ordered_for_tail(out)))
    if world > 1 or args.force_collectives:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
