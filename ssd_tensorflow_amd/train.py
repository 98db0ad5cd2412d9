#!/usr/bin/env python3
"""Training driver: the counterpart of the reference's train.py loop (train.py:247-343) over
the HIP library.  Same flags (train.py:55-80) plus --preset / --synthetic-train / --synthetic-valid.
TensorBoard summaries, AP bookkeeping and image dumps are out of scope.

    python -m ssd_tensorflow_amd.train --name run1 --epochs 2 --batch-size 8
    python -m torch.distributed.run --nproc-per-node 8 -m ssd_tensorflow_amd.train ...   # data parallel
"""
import argparse
import math
import os
import sys

import numpy as np

from . import parallel
from .average_precision import APCalculator, APs2mAP
from .ssdutils import boxes_from_detection
from .ssdvgg import SSDVGG, Session, LearningRate
from .training_data import TrainingData
from .utils import str2bool


def main(argv=None):
    parser = argparse.ArgumentParser(description='Train the SSD')
    parser.add_argument('--name', default='test', help='project name')
    parser.add_argument('--data-dir', default='synthetic', help='data directory')
    parser.add_argument('--vgg-dir', default='vgg_graph', help='directory for the VGG-16 model')
    parser.add_argument('--epochs', type=int, default=200, help='number of training epochs')
    parser.add_argument('--batch-size', type=int, default=8, help='batch size (per GPU)')
    parser.add_argument('--tensorboard-dir', default='tb', help='name of the tensorboard data directory')
    parser.add_argument('--checkpoint-interval', type=int, default=5, help='checkpoint interval')
    parser.add_argument('--lr-values', type=str, default='0.00075;0.0001;0.00001', help='learning rate values')
    parser.add_argument('--lr-boundaries', type=str, default='320000;400000', help='learning rate chage boundaries (in batches)')
    parser.add_argument('--momentum', type=float, default=0.9, help='momentum for the optimizer')
    parser.add_argument('--weight-decay', type=float, default=0.0005, help='L2 normalization factor')
    parser.add_argument('--continue-training', type=str2bool, default='False', help='continue training from the latest checkpoint')
    parser.add_argument('--num-workers', type=int, default=0, help='number of parallel generators')
    parser.add_argument('--preset', default='vgg300')
    parser.add_argument('--dtype', default='f32', choices=['f32', 'bf16'], help='f32, or bf16 activations on the bf16 matrix cores (fp32 master weights, loss and optimizer)')
    parser.add_argument('--synthetic-train', type=int, default=64, help='synthetic training samples per epoch')
    parser.add_argument('--synthetic-valid', type=int, default=16)
    parser.add_argument('--augment', type=str2bool, default='False', help="run the reference's train augmentation recipe (process_dataset.py) on the GPU over a uint8 synthetic dataset")
    args = parser.parse_args(argv)

    rank, local, world = parallel.init()
    say = print if rank == 0 else (lambda *a, **k: None)
    say('[i] Project name:         ', args.name)
    say('[i] Data directory:       ', args.data_dir)
    say('[i] # epochs:             ', args.epochs)
    say('[i] Batch size:           ', args.batch_size, 'x', world, 'GPU(s)')
    say('[i] Learning rate values: ', args.lr_values)
    say('[i] Learning rate boundaries: ', args.lr_boundaries)
    say('[i] Momentum:             ', args.momentum)
    say('[i] Weight decay:         ', args.weight_decay)
    say('[i] Continue:             ', args.continue_training)

    try:
        lr_values = [float(v) for v in args.lr_values.split(';')]
        lr_boundaries = [int(v) for v in args.lr_boundaries.split(';')] if args.lr_boundaries else []
    except ValueError:
        print('[!] Learning rate values or boundaries are invalid'); return 1            # train.py:174-185
    if len(lr_values) != len(lr_boundaries) + 1:
        print('[!] Learning rate values must be one more than boundaries'); return 1

    # ---- find an existing checkpoint (train.py:99-134) ---------------------------------------
    start_epoch = 0
    ckpt = None
    if args.continue_training:
        cands = [f for f in (os.listdir(args.name) if os.path.isdir(args.name) else []) if f.startswith('e') and f.endswith('.npz')]
        if not cands:
            print('[!] No network state found in ' + args.name); return 1
        start_epoch = max(int(f[1:-4]) for f in cands)
        ckpt = os.path.join(args.name, f'e{start_epoch}.npz')
        say('[i] Last checkpoint:      ', ckpt)
    elif rank == 0:
        os.makedirs(args.name, exist_ok=True)

    try:
        td = TrainingData(args.data_dir, args.preset, args.synthetic_train, args.synthetic_valid, rank=rank, world=world,
                          augment=args.augment, device=local)
    except RuntimeError as e:
        print('[!] Unable to load training data:', str(e)); return 1                       # train.py:155-161
    say('[i] # training samples:   ', td.num_train)
    say('[i] # validation samples: ', td.num_valid)
    say('[i] # classes:            ', td.num_classes)
    say('[i] Image size:           ', td.preset.image_size)

    import torch
    lr = LearningRate(lr_values, lr_boundaries)
    with Session(local) as sess:
        say('[i] Creating the model...')
        net = SSDVGG(sess, td.preset)
        if ckpt:
            net.build_from_metagraph(None, ckpt, max_batch=args.batch_size, training=True, dtype=args.dtype)
            net.build_optimizer_from_metagraph()
        else:
            net.build_from_vgg(args.vgg_dir, td.num_classes, max_batch=args.batch_size, dtype=args.dtype)
            net.build_optimizer(learning_rate=lr, weight_decay=args.weight_decay, momentum=args.momentum)
        if world > 1:
            torch.distributed.broadcast(net.params_flat, 0)
        net.set_stream(torch.cuda.current_stream().cuda_stream)

        training_ap_calc = APCalculator()
        validation_ap_calc = APCalculator()
        say('[i] Training...')
        for e in range(start_epoch, args.epochs):
            td.epoch = e
            # ---- train (train.py:254-281) --------------------------------------------------------
            tot = np.zeros(4); seen = 0
            for x, y, gt_boxes in td.train_generator(args.batch_size, args.num_workers):
                if world > 1:
                    xt = x if torch.is_tensor(x) else torch.from_numpy(x).cuda(non_blocking=True)
                    yt = torch.from_numpy(y).cuda(non_blocking=True)
                    parallel.train_step_dp(net, xt, yt, world)
                    loss_batch = net.get_losses()
                else:
                    result, loss_batch, _ = sess.run([net.result, net.losses, net.optimizer],
                                                     feed_dict={net.image_input: x, net.labels: y})
                if math.isnan(loss_batch['confidence']):
                    print('[!] Confidence loss is NaN.')
                tot += np.array([loss_batch[k] for k in ('total', 'localization', 'confidence', 'l2')]) * x.shape[0]
                seen += x.shape[0]
                if e == 0:
                    continue
                # decode + NMS of the batch just computed, on the GPU (train.py:275-277), then AP bookkeeping
                dets = net.detect_last(x.shape[0], 0.5, 200, None)
                for i in range(x.shape[0]):
                    training_ap_calc.add_detections(gt_boxes[i], boxes_from_detection(dets[i], td.lid2name))
            tot = np.array(parallel.mean_scalars(tot / max(seen, 1), world, 'cuda' if world > 1 else None))
            say('[i] Train {:>2}/{}  total {:.4f}  localization {:.4f}  confidence {:.4f}  l2 {:.4f}'.format(e + 1, args.epochs, *tot))
            # ---- validate (train.py:286-306) ---------------------------------------------------
            vt = np.zeros(4); vs = 0
            for x, y, gt_boxes in td.valid_generator(args.batch_size, args.num_workers):
                result, loss_batch = sess.run([net.result, net.losses], feed_dict={net.image_input: x, net.labels: y})
                vt += np.array([loss_batch[k] for k in ('total', 'localization', 'confidence', 'l2')]) * x.shape[0]
                vs += x.shape[0]
                if e == 0:
                    continue
                dets = net.detect_last(x.shape[0], 0.5, 200, None)
                for i in range(x.shape[0]):
                    validation_ap_calc.add_detections(gt_boxes[i], boxes_from_detection(dets[i], td.lid2name))
            vt = np.array(parallel.mean_scalars(vt / max(vs, 1), world, 'cuda' if world > 1 else None))
            say('[i] Valid {:>2}/{}  total {:.4f}  localization {:.4f}  confidence {:.4f}  l2 {:.4f}'.format(e + 1, args.epochs, *vt))
            # ---- mAP of this rank's shard (train.py:317-323, VOC07 11-point, on the GPU) ---------------------
            if e > 0:
                say('[i] mAP  {:>2}/{}  training {:.4f}  validation {:.4f}'.format(
                    e + 1, args.epochs, APs2mAP(training_ap_calc.compute_aps()), APs2mAP(validation_ap_calc.compute_aps())))
            training_ap_calc.clear(); validation_ap_calc.clear()
            # ---- checkpoint (train.py:336-343) -------------------------------------------------
            if (e + 1) % args.checkpoint_interval == 0 and rank == 0:
                path = '{}/e{}.npz'.format(args.name, e + 1)
                net.save_checkpoint(path, lr, args.momentum, args.weight_decay)
                print('[i] Checkpoint saved:', path)
        if rank == 0:
            path = '{}/final.npz'.format(args.name)
            net.save_checkpoint(path, lr, args.momentum, args.weight_decay)
            print('[i] Checkpoint saved:', path)
    return 0


if __name__ == '__main__':
    sys.exit(main())
