#!/usr/bin/env python3
"""Inference driver: the counterpart of the reference's infer.py loop (infer.py:213-280) over the
HIP library: restore a checkpoint, batch images, run the net, decode + NMS, keep [:200], AP statistics,
VOC summary files.  Same flags (infer.py:62-89) plus --synthetic / --preset / --dtype.

The loop is pipelined: images are resized on the GPU (the augmentation kernel's cv2.INTER_LINEAR path),
the net and decode + NMS run on the result where it lies, and the (small) detections of batch k are
collected after batch k+1 has been launched -- the [b, A, C+5] predictions only come to the host for
--dump-predictions.  Files: anything Pillow decodes, or .npy arrays (uint8 / float32 BGR); cv2 drawing
(--annotate) is out of scope.
"""
import argparse
import os
import sys

import numpy as np

from .average_precision import APCalculator, APs2mAP
from .ssdvgg import SSDVGG, Session
from .ssdutils import get_preset_by_name, boxes_from_detection
from .training_data import VOC_NAMES
from .pascal_summary import PascalSummary
from .utils import Size, str2bool, load_data_source


def sample_generator(samples, image_size, batch_size, device=0):
    """infer.py:44-54: cv2.resize(cv2.imread(file), image_size).astype(float32) per batch -- here a batch of load +
    INTER_LINEAR resize plans executed by the augmentation kernel; yields (CUDA tensor [b,H,W,3], indices, sizes)."""
    from . import transforms as T
    for offset in range(0, len(samples), batch_size):
        files = samples[offset:offset + batch_size]
        plans, idxs, sizes, ready = [], [], [], []
        for i, f in enumerate(files):
            img = f if isinstance(f, np.ndarray) else T.load_image_bgr(f)
            idxs.append(offset + i)
            sizes.append(Size(img.shape[1], img.shape[0]))
            if img.dtype == np.uint8:
                plan = T.ImagePlan(img)
                plan.resize = (image_size.w, image_size.h, T.INTER_LINEAR)
                plans.append(plan); ready.append(None)
            else:       # a float image is fed as it is (the reference's resize would run on uint8 pixels)
                if img.shape[:2] != (image_size.h, image_size.w):
                    raise ValueError(f'{f}: float images must already be {image_size.h}x{image_size.w}, got {img.shape[:2]}')
                ready.append(np.ascontiguousarray(img, np.float32))
        import torch
        dev = torch.device('cuda', device)
        x = torch.empty((len(files), image_size.h, image_size.w, 3), dtype=torch.float32, device=dev)
        if plans:
            res = T.augment_batch(plans, image_size.w, image_size.h, device=device)
            k = 0
            for i, r in enumerate(ready):
                if r is None:
                    x[i] = res[k]; k += 1
        for i, r in enumerate(ready):
            if r is not None:
                x[i] = torch.from_numpy(r).to(dev)
        yield x, idxs, sizes


def main(argv=None):
    parser = argparse.ArgumentParser(description='SSD inference')
    parser.add_argument('files', type=str, nargs='*', help='image files (anything Pillow decodes, or .npy arrays)')
    parser.add_argument('--name', default='test', help='project name')
    parser.add_argument('--checkpoint', type=int, default=-1, help='checkpoint to restore; -1 is the most recent')
    parser.add_argument('--training-data', default='', help='unused: class names come from the data source (VOC defaults)')
    parser.add_argument('--output-dir', default='test-output', help='directory for the resulting predictions')
    parser.add_argument('--annotate', type=str2bool, default='False', help='out of scope (cv2 drawing)')
    parser.add_argument('--dump-predictions', type=str2bool, default='False', help='Dump raw predictions')
    parser.add_argument('--compute-stats', type=str2bool, default='True', help='Compute the mAP stats')
    parser.add_argument('--data-source', default=None, help='Use test files from the data source')
    parser.add_argument('--data-dir', default='pascal-voc', help='Use test files from the data source')
    parser.add_argument('--batch-size', type=int, default=32, help='batch size')
    parser.add_argument('--sample', default='test', choices=['test', 'trainval'], help='sample to run on')
    parser.add_argument('--threshold', type=float, default=0.5, help='confidence threshold')
    parser.add_argument('--pascal-summary', type=str2bool, default='False', help='dump the detections in Pascal VOC format')
    parser.add_argument('--synthetic', type=int, default=0, help='run on N synthetic images instead of files')
    parser.add_argument('--preset', default=None, help='preset when no checkpoint is given (random weights)')
    parser.add_argument('--dtype', default='f32', choices=['f32', 'bf16'], help='f32, or bf16 activations on the bf16 matrix cores')
    args = parser.parse_args(argv)

    print('[i] Project name:      ', args.name)
    print('[i] Batch size:        ', args.batch_size)
    print('[i] Data source:       ', args.data_source)
    print('[i] Data directory:    ', args.data_dir)
    print('[i] Output directory:  ', args.output_dir)
    print('[i] Dump predictions:  ', args.dump_predictions)
    print('[i] Sample:            ', args.sample)
    print('[i] Threshold:         ', args.threshold)
    print('[i] Pascal summary:    ', args.pascal_summary)
    if args.annotate:
        print('[!] --annotate needs OpenCV drawing, which this build does not have'); return 1

    # ---- checkpoint lookup (infer.py:111-126) --------------------------------------------------
    ckpt = None
    if os.path.isdir(args.name):
        if args.checkpoint == -1:
            cands = [f for f in os.listdir(args.name) if f.endswith('.npz')]
            epochs = sorted((f for f in cands if f[0] == 'e' and f[1:-4].isdigit()), key=lambda f: int(f[1:-4]))
            ckpt = os.path.join(args.name, 'final.npz') if 'final.npz' in cands else (os.path.join(args.name, epochs[-1]) if epochs else None)
        else:
            ckpt = '{}/e{}.npz'.format(args.name, args.checkpoint)
    if ckpt is None or not os.path.exists(ckpt):
        if args.preset is None:
            if ckpt is None:
                print('[!] No network state found in ' + args.name)
            else:
                print('[!] Cannot find checkpoint ' + ckpt)
            return 1                                                                        # infer.py:111-126
        ckpt = None

    # ---- data source (infer.py:147-171) ----------------------------------------------------------
    compute_stats = False
    source, samples = None, None
    lid2name = dict(enumerate(VOC_NAMES))
    if args.data_source:
        print('[i] Configuring the data source...')
        try:
            source = load_data_source(args.data_source)
            if args.sample == 'test':
                source.load_test_data(args.data_dir)
                num_samples, samples = source.num_test, source.test_samples
            else:
                source.load_trainval_data(args.data_dir, 0)
                num_samples, samples = source.num_train, source.train_samples
            print('[i] # samples:         ', num_samples)
            print('[i] # classes:         ', source.num_classes)
        except (ImportError, AttributeError, RuntimeError, OSError) as e:
            print('[!] Unable to load data source:', str(e)); return 1
        lid2name = source.lid2name
        compute_stats = bool(args.compute_stats)

    with Session(0) as sess:
        print('[i] Creating the model...')
        if ckpt:
            pname = str(np.load(ckpt)['__preset__'])
            net = SSDVGG(sess, get_preset_by_name(pname))
            net.build_from_metagraph(None, ckpt, max_batch=args.batch_size, dtype=args.dtype)
        else:
            net = SSDVGG(sess, get_preset_by_name(args.preset))
            net.build_from_vgg(None, 20, max_batch=args.batch_size, training=False, dtype=args.dtype)
        size = net.preset.image_size
        # ---- files to analyse (infer.py:177-193) ----------------------------------------------------
        if source:
            files = [s.filename for s in samples]
        elif args.synthetic:
            rng = np.random.default_rng(1)
            files = [rng.integers(0, 256, (size.h, size.w, 3)).astype(np.float32) for _ in range(args.synthetic)]
        else:
            files = list(args.files)
            if not files:
                print('[!] No files specified'); return 1
        keep = [i for i, f in enumerate(files) if isinstance(f, np.ndarray) or os.path.exists(f) or os.path.exists(f + '.npy')]
        files = [files[i] for i in keep]
        if samples is not None:
            samples = [samples[i] for i in keep]
        if files:
            os.makedirs(args.output_dir, exist_ok=True)
        print('[i] Compute stats:     ', compute_stats)
        print('[i] Network checkpoint:', ckpt or '(random weights, --preset ' + str(args.preset) + ')')
        print('[i] Image size:        ', size)
        print('[i] Number of files:   ', len(files))
        ap_calc = APCalculator() if compute_stats else None
        pascal_summary = PascalSummary() if args.pascal_summary else None                    # infer.py:208-209

        def name_of(i):
            return files[i] if isinstance(files[i], str) else f'{i:06d}.npy'

        total = 0

        def collect(pending):
            nonlocal total
            ticket, idxs, sizes = pending
            for i, det in enumerate(ticket.get()):
                # decode_boxes(enc, anchors, threshold, lid2name, None); suppress_overlaps(boxes)[:200]  (infer.py:233-235)
                boxes = boxes_from_detection(det, lid2name)
                total += len(boxes)
                if compute_stats:                                                            # infer.py:259-260
                    ap_calc.add_detections(samples[idxs[i]].boxes, boxes)
                if pascal_summary is not None:                                               # infer.py:263-264
                    pascal_summary.add_detections(name_of(idxs[i]), boxes, img_size=sizes[i])

        pending = None
        for x, idxs, sizes in sample_generator(files, size, args.batch_size):
            net.infer_dev(x)                                                                 # infer.py:225-227
            ticket = net.detect_last_launch(x.shape[0], args.threshold, None, 200)
            if args.dump_predictions:                                                        # infer.py:251-254
                enc_boxes = net._dev_result(x.shape[0], True)
                for i in range(x.shape[0]):
                    np.save(os.path.join(args.output_dir, os.path.basename(name_of(idxs[i])) + '.npy'), enc_boxes[i])
            if pending:
                collect(pending)
            pending = (ticket, idxs, sizes)
        if pending:
            collect(pending)

        if compute_stats:                                                                    # infer.py:269-273
            aps = ap_calc.compute_aps()
            for k, v in aps.items():
                print('[i] AP [{0}]: {1:.3f}'.format(k, v))
            print('[i] mAP: {0:.3f}'.format(APs2mAP(aps)))
        if pascal_summary is not None:                                                       # infer.py:278-279
            pascal_summary.write_summary(args.output_dir)
        print('[i] Processed {} images, {} detections'.format(len(files), total))
    print('[i] All done.')
    return 0


if __name__ == '__main__':
    sys.exit(main())
