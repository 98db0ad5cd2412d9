#!/usr/bin/env python3
"""Inference driver: the counterpart of the reference's infer.py loop (infer.py:213-264) over the
HIP library: restore a checkpoint, batch images, run the net, decode + NMS, keep [:200].
File I/O with cv2 (imread/resize/annotate), AP statistics and the VOC summary are out of scope;
--synthetic N feeds N random images, .npy files (already HxWx3 float32/uint8 BGR) are accepted as `files`.
"""
import argparse
import math
import os
import sys

import numpy as np

from .ssdvgg import SSDVGG, Session
from .ssdutils import get_preset_by_name, boxes_from_detection
from .training_data import VOC_NAMES
from .pascal_summary import PascalSummary
from .utils import Size


def sample_generator(samples, image_size, batch_size):
    """infer.py:44-54 with .npy arrays instead of cv2.imread + resize."""
    for offset in range(0, len(samples), batch_size):
        files = samples[offset:offset + batch_size]
        images, idxs = [], []
        for i, f in enumerate(files):
            img = f if isinstance(f, np.ndarray) else np.load(f)
            if img.shape[:2] != (image_size.h, image_size.w):
                raise ValueError(f'{f}: expected {image_size.h}x{image_size.w} pixels, got {img.shape[:2]} (no cv2 resize here)')
            images.append(img.astype(np.float32)); idxs.append(offset + i)
        yield np.array(images), idxs


def main(argv=None):
    parser = argparse.ArgumentParser(description='SSD inference')
    parser.add_argument('files', type=str, nargs='*', help='.npy image files')
    parser.add_argument('--name', default='test', help='project name')
    parser.add_argument('--checkpoint', type=int, default=-1, help='checkpoint to restore; -1 is the most recent')
    parser.add_argument('--output-dir', default='test-output', help='directory for the resulting predictions')
    parser.add_argument('--dump-predictions', type=lambda v: v.lower() in ('1', 'true', 'yes', 'y', 't'), default=False)
    parser.add_argument('--batch-size', type=int, default=32, help='batch size')
    parser.add_argument('--threshold', type=float, default=0.5, help='confidence threshold')
    # accepted for command-line compatibility with the reference (infer.py:62-89); the features behind
    # them (cv2 annotation, AP statistics, VOC dataset / summary) are out of scope
    parser.add_argument('--training-data', default='', help='unused: class names are the VOC defaults')
    parser.add_argument('--annotate', default='False', help='out of scope (cv2 drawing)')
    parser.add_argument('--compute-stats', default='True', help='out of scope (AP statistics)')
    parser.add_argument('--data-source', default=None, help='out of scope (dataset readers)')
    parser.add_argument('--data-dir', default='pascal-voc', help='out of scope (dataset readers)')
    parser.add_argument('--sample', default='test', choices=['test', 'trainval'], help='out of scope')
    parser.add_argument('--pascal-summary', type=lambda v: v.lower() in ('1', 'true', 'yes', 'y', 't'), default=False,
                        help='write VOC comp4 submission files (pascal_summary.py) to --output-dir')
    parser.add_argument('--synthetic', type=int, default=0, help='run on N synthetic images instead of files')
    parser.add_argument('--preset', default=None, help='preset when no checkpoint is given')
    parser.add_argument('--dtype', default='f32', choices=['f32', 'bf16'], help='f32, or bf16 activations on the bf16 matrix cores (fp32 master weights, loss and optimizer)')
    args = parser.parse_args(argv)

    # ---- checkpoint lookup (infer.py:111-126) --------------------------------------------------
    ckpt = None
    if os.path.isdir(args.name):
        if args.checkpoint == -1:
            cands = [f for f in os.listdir(args.name) if f.endswith('.npz')]
            ckpt = os.path.join(args.name, 'final.npz') if 'final.npz' in cands else (
                os.path.join(args.name, sorted(cands, key=lambda f: int(f[1:-4]))[-1]) if cands else None)
        else:
            ckpt = '{}/e{}.npz'.format(args.name, args.checkpoint)
    if ckpt is None or not os.path.exists(ckpt):
        if args.preset is None:
            print('[!] Cannot find checkpoint in ' + args.name); return 1                    # infer.py:113-126
        ckpt = None
    if args.data_source:
        print('[!] --data-source is not built (dataset readers are out of scope); use files or --synthetic'); return 1
    print('[i] Project name:      ', args.name)
    print('[i] Checkpoint:        ', ckpt or '(random weights, --preset ' + str(args.preset) + ')')
    print('[i] Batch size:        ', args.batch_size)
    print('[i] Threshold:         ', args.threshold)
    lid2name = dict(enumerate(VOC_NAMES))

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
        files = list(args.files)
        if args.synthetic:
            rng = np.random.default_rng(1)
            files = [rng.integers(0, 256, (size.h, size.w, 3)).astype(np.float32) for _ in range(args.synthetic)]
        if not files:
            print('[!] No files specified'); return 1
        sizes = [Size(size.w, size.h)] * len(files)       # the images fed ARE the files here (no decode / resize step)                                        # infer.py:147-149
        if args.dump_predictions or args.pascal_summary:
            os.makedirs(args.output_dir, exist_ok=True)
        pascal_summary = PascalSummary() if args.pascal_summary else None                    # infer.py:208-209
        total = 0
        for x, idxs in sample_generator(files, size, args.batch_size):
            enc_boxes = sess.run(net.result, feed_dict={net.image_input: x, net.keep_prob: 1})
            # decode_boxes(enc, anchors, threshold, lid2name, None); suppress_overlaps(boxes)[:200]  (infer.py:233-235)
            dets = net.detect_last(x.shape[0], args.threshold, None, 200)
            for i, det in enumerate(dets):
                boxes = boxes_from_detection(det, lid2name)
                total += len(boxes)
                if pascal_summary is not None:                                               # infer.py:263-264
                    name = files[idxs[i]] if isinstance(files[idxs[i]], str) else f'{idxs[i]:06d}.npy'
                    pascal_summary.add_detections(name, boxes, img_size=sizes[idxs[i]])
                if args.dump_predictions:
                    with open(os.path.join(args.output_dir, f'{idxs[i]:06d}.txt'), 'w') as f:
                        for conf, b in boxes:                                                # infer.py:251-258
                            f.write('{} {} {} {} {} {}\n'.format(b.label, b.center.x, b.center.y, b.size.w, b.size.h, conf))
        if pascal_summary is not None:                                                       # infer.py:278-279
            pascal_summary.write_summary(args.output_dir)
        print('[i] Processed {} images, {} detections'.format(len(files), total))
    return 0


if __name__ == '__main__':
    sys.exit(main())
