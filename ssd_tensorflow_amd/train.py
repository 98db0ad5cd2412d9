#!/usr/bin/env python3
"""Training driver: the counterpart of the reference's train.py loop (train.py:247-343) over
the HIP library.  Same flags (train.py:55-80) plus --preset / --synthetic-train / --synthetic-valid /
--augment / --dtype / --allreduce-bucket-mb / --allreduce-dtype.  Scalar summaries go to <tensorboard-dir>/<name>/scalars.jsonl
(summaries.py); image summaries and TensorBoard event files are out of scope.

    python -m ssd_tensorflow_amd.train --name run1 --epochs 2 --batch-size 8
    python -m torch.distributed.run --nproc-per-node 8 -m ssd_tensorflow_amd.train ...   # data parallel

A batch never leaves the GPU: the feeder hands out device tensors, the step runs on them, decode + NMS
for the AP bookkeeping runs on the result where it lies (the reference fetches `result` to the host for
decode_boxes, train.py:262-277).  --num-workers N means what the reference's means: N forked processes
prepare the next batches beside the step (training_data.py).  The loop (StepLoop) never waits for the step
it has just launched: losses are read one step late, detections one pass late.  Data parallel: one process
per GPU, rank-sharded batches, bucketed all-reduce of the gradient arena overlapped with backward
(parallel.train_step_dp); losses are summed and detections gathered over ranks for the epoch summaries."""
import argparse
import math
import os
import sys

import numpy as np

from . import _lib, parallel
from .average_precision import APCalculator, APs2mAP
from .ssdutils import boxes_from_detection
from .ssdvgg import SSDVGG, Session, LearningRate
from .summaries import SummaryWriter, LossSummary, PrecisionSummary
from .training_data import TrainingData
from .utils import str2bool


def compute_lr(lr_values, lr_boundaries):
    """train.py:43-47"""
    return LearningRate(lr_values, lr_boundaries)


class StepLoop:
    """The body of the reference's epoch loops (train.py:254-281 training, 286-306 validation) over the HIP library,
    shared by main() below and by bench.py's end-to-end block.  The host never waits for the step it has just launched:
    the losses of step k are read after step k + 1 is in flight (ssd_get_losses_step), the detections of step k are
    collected after step k + 1's decode has been launched, and with num_workers > 0 batch k + 1 is already in HBM
    (training_data.py)."""

    def __init__(self, net, sess, td, batch_size, num_workers=0, world=1, rank=0, bucket=0, allreduce_dtype='f32'):
        self.net, self.sess, self.td = net, sess, td
        self.batch_size, self.num_workers = batch_size, num_workers
        self.world, self.rank, self.bucket = world, rank, bucket
        self.allreduce_dtype = allreduce_dtype
        self.steps = 0

    def booked(self, loss_batch, count):
        """(values, weight) for LossSummary.add.  One GPU: the batch's losses, weighted by its samples
        (train.py:271).  Data parallel: a rank's data terms are normalised by count / world, so summed over ranks
        with weight count / world they are count x the global-batch value; the l2 term is the same number on every
        rank that ran a step and absent on a rank whose shard was empty, so rank 0 (never empty) books it, once."""
        world = self.world
        if world <= 1:
            return loss_batch, count
        data = loss_batch['localization'] + loss_batch['confidence']
        l2 = loss_batch['l2'] * world if self.rank == 0 else 0.0
        return dict(total=data + l2, localization=loss_batch['localization'], confidence=loss_batch['confidence'], l2=l2), count / world

    def book(self, loss_batch, count, loss_summary):
        if math.isnan(loss_batch['confidence']):
            print('[!] Confidence loss is NaN.')                                     # train.py:268-269
        if loss_summary is not None:
            loss_summary.add(*self.booked(loss_batch, count))

    def collect(self, dets, gt_boxes, calc):
        if dets is None:
            return
        for gt, det in zip(gt_boxes, dets.get()):
            calc.add_detections(gt, boxes_from_detection(det, self.td.lid2name))

    def run_epoch(self, generator, train, loss_summary, ap_calc, with_ap):
        net, sess, td, world = self.net, self.sess, self.td, self.world
        pending_det = None
        pending_loss = None                     # sample count of the launched step whose losses are not booked yet
        for x, y, gt_boxes in generator(self.batch_size, self.num_workers):
            n = len(gt_boxes)
            count = td.global_count if world > 1 else n
            ran = True
            if train and world > 1:
                parallel.train_step_dp(net, x, y, world, self.bucket, td.global_count, allreduce_dtype=self.allreduce_dtype)       # an empty shard still steps
            elif n == 0:
                ran = False
            elif train:
                sess.run(net.optimizer, feed_dict={net.image_input: x, net.labels: y})
            else:
                if world > 1:
                    net.set_loss_normalizer(td.global_count / world)
                sess.run(net.eval_op, feed_dict={net.image_input: x, net.labels: y})
                if world > 1:
                    net.set_loss_normalizer(0.0)
            self.steps += int(ran)
            if pending_loss is not None:
                self.book(net.get_losses_step(1 if ran else 0), pending_loss, loss_summary)
            pending_loss = count if ran else None
            if not with_ap or n == 0:
                continue
            # decode + NMS of the batch just computed, on the GPU (train.py:275-277)
            launched = net.detect_last_launch(n, 0.5, 200, None)
            if pending_det:
                self.collect(pending_det[0], pending_det[1], ap_calc)
            pending_det = (launched, gt_boxes)
        if pending_loss is not None:
            self.book(net.get_losses_step(0), pending_loss, loss_summary)
        if pending_det:
            self.collect(pending_det[0], pending_det[1], ap_calc)


def main(argv=None):
    parser = argparse.ArgumentParser(description='Train the SSD')
    parser.add_argument('--name', default='test', help='project name')
    parser.add_argument('--data-dir', default='synthetic', help='data directory')
    parser.add_argument('--vgg-dir', default='vgg_graph', help='directory for the VGG-16 model')
    parser.add_argument('--epochs', type=int, default=200, help='number of training epochs')
    parser.add_argument('--batch-size', type=int, default=8, help='batch size (per GPU)')
    parser.add_argument('--tensorboard-dir', default='tb', help='name of the summary data directory')
    parser.add_argument('--checkpoint-interval', type=int, default=5, help='checkpoint interval')
    parser.add_argument('--lr-values', type=str, default='0.00075;0.0001;0.00001', help='learning rate values')
    parser.add_argument('--lr-boundaries', type=str, default='320000;400000', help='learning rate chage boundaries (in batches)')
    parser.add_argument('--momentum', type=float, default=0.9, help='momentum for the optimizer')
    parser.add_argument('--weight-decay', type=float, default=0.0005, help='L2 normalization factor')
    parser.add_argument('--continue-training', type=str2bool, default='False', help='continue training from the latest checkpoint')
    parser.add_argument('--num-workers', type=int, default=0, help='number of parallel generators')
    parser.add_argument('--preset', default='vgg300')
    parser.add_argument('--data-source', default='pascal_voc', help='data source module for a real --data-dir')
    parser.add_argument('--dtype', default='f32', choices=['f32', 'bf16'], help='f32, or bf16 activations on the bf16 matrix cores (fp32 master weights, loss and optimizer)')
    parser.add_argument('--synthetic-train', type=int, default=64, help='synthetic training samples per epoch')
    parser.add_argument('--synthetic-valid', type=int, default=16)
    parser.add_argument('--augment', type=str2bool, default='False', help="run the reference's train augmentation recipe (process_dataset.py) on the GPU over a uint8 synthetic dataset")
    parser.add_argument('--allreduce-dtype', default='f32', choices=['f32', 'bf16'], help='data parallel: bf16 = the filter gradients cross the links as bf16 messages of half the bytes (fp32 masters, momentum and arenas untouched)')
    parser.add_argument('--allreduce-bucket-mb', type=float, default=44, help='data parallel: all-reduce finished gradient ranges of >= this many MB while backward still runs (0 = one all-reduce after backward); 44 = three buckets: heads ... mod_conv6 | conv5_x + conv4_3/4_2 | the rest')
    args = parser.parse_args(argv)

    rank, local, world = parallel.init()
    import torch
    torch.cuda.set_device(local)           # every backend: kernels, .cuda() tensors and collectives of this rank on ITS GPU
    _lib.set_device(local)                 # ... and the package's free functions (label encoder, AP)
    say = print if rank == 0 else (lambda *a, **k: None)
    say('[i] Project name:         ', args.name)
    say('[i] Data directory:       ', args.data_dir)
    say('[i] # epochs:             ', args.epochs)
    say('[i] Batch size:           ', args.batch_size, 'x', world, 'GPU(s)')
    say('[i] Tensorboard directory:', args.tensorboard_dir)
    say('[i] Checkpoint interval:  ', args.checkpoint_interval)
    say('[i] Learning rate values: ', args.lr_values)
    say('[i] Learning rate boundaries: ', args.lr_boundaries)
    say('[i] Momentum:             ', args.momentum)
    say('[i] Weight decay:         ', args.weight_decay)
    say('[i] Continue:             ', args.continue_training)
    say('[i] Number of workers:    ', args.num_workers)

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
        cands = [f for f in (os.listdir(args.name) if os.path.isdir(args.name) else [])
                 if f.startswith('e') and f.endswith('.npz') and f[1:-4].isdigit()]
        if not cands:
            print('[!] No network state found in ' + args.name); return 1
        start_epoch = max(int(f[1:-4]) for f in cands)
        ckpt = os.path.join(args.name, f'e{start_epoch}.npz')
        say('[i] Last checkpoint:      ', ckpt)
    elif rank == 0:
        try:
            os.makedirs(args.name, exist_ok=True)
        except OSError as e:
            print('[!] Cannot create directory {}: {}'.format(args.name, e)); return 1      # train.py:143-145

    try:
        td = TrainingData(args.data_dir, args.preset, args.synthetic_train, args.synthetic_valid, rank=rank, world=world,
                          augment=args.augment, device=local, data_source=args.data_source)
    except RuntimeError as e:
        print('[!] Unable to load training data:', str(e)); return 1                       # train.py:155-161
    say('[i] # training samples:   ', td.num_train)
    say('[i] # validation samples: ', td.num_valid)
    say('[i] # classes:            ', td.num_classes)
    say('[i] Image size:           ', td.preset.image_size)

    lr = compute_lr(lr_values, lr_boundaries)
    bucket = int(args.allreduce_bucket_mb * 1e6 / 4)
    dev = torch.device('cuda', local)

    def rank_sum(values):
        if world <= 1:
            return list(values)
        t = torch.tensor(list(values), dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t)
        return t.tolist()

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

        writer = SummaryWriter(os.path.join(args.tensorboard_dir, os.path.basename(os.path.normpath(args.name)))) if rank == 0 else None
        training_ap_calc = APCalculator()
        validation_ap_calc = APCalculator()
        labels = list(td.lname2id.keys())
        training_ap = PrecisionSummary(writer, 'training', labels)
        validation_ap = PrecisionSummary(writer, 'validation', labels)
        training_loss = LossSummary(writer, 'training', td.num_train)
        validation_loss = LossSummary(writer, 'validation', td.num_valid)

        loop = StepLoop(net, sess, td, args.batch_size, args.num_workers, world, rank, bucket, args.allreduce_dtype)

        def gathered_aps(calc):
            """{label: AP} over the WHOLE sample (train.py:317-323): data parallel, every rank's detections and ground
            truth are gathered to rank 0, which computes; the other ranks get {}."""
            if world <= 1:
                return calc.compute_aps()
            states = [None] * world if rank == 0 else None
            torch.distributed.gather_object(calc.state(), states, dst=0)
            if rank != 0:
                return {}
            merged = APCalculator(calc.minoverlap)
            for st in states:
                merged.merge(st)
            return merged.compute_aps()

        say('[i] Training...')
        for e in range(start_epoch, args.epochs):
            td.epoch = e
            loop.run_epoch(td.train_generator, True, training_loss, training_ap_calc, e > 0)
            loop.run_epoch(td.valid_generator, False, validation_loss, validation_ap_calc, e > 0)
            # ---- summaries (train.py:311-331) -----------------------------------------------------
            tl = training_loss.push(e + 1, rank_sum)
            vl = validation_loss.push(e + 1, rank_sum)
            say('[i] Train {:>2}/{}  total {:.4f}  localization {:.4f}  confidence {:.4f}  l2 {:.4f}'.format(
                e + 1, args.epochs, tl['total'], tl['localization'], tl['confidence'], tl['l2']))
            say('[i] Valid {:>2}/{}  total {:.4f}  localization {:.4f}  confidence {:.4f}  l2 {:.4f}'.format(
                e + 1, args.epochs, vl['total'], vl['localization'], vl['confidence'], vl['l2']))
            # VOC07 11-point AP on the GPU, over all ranks' samples
            APs = gathered_aps(training_ap_calc); mAP = APs2mAP(APs)
            training_ap.push(e + 1, mAP, APs)
            vAPs = gathered_aps(validation_ap_calc); vmAP = APs2mAP(vAPs)
            validation_ap.push(e + 1, vmAP, vAPs)
            if e > 0:
                say('[i] mAP  {:>2}/{}  training {:.4f}  validation {:.4f}'.format(e + 1, args.epochs, mAP, vmAP))
            training_ap_calc.clear(); validation_ap_calc.clear()
            if writer is not None:
                writer.flush()
            # ---- checkpoint (train.py:336-343) -------------------------------------------------
            if (e + 1) % args.checkpoint_interval == 0 and rank == 0:
                path = '{}/e{}.npz'.format(args.name, e + 1)
                net.save_checkpoint(path, lr, args.momentum, args.weight_decay)
                print('[i] Checkpoint saved:', path)
        if rank == 0:
            path = '{}/final.npz'.format(args.name)
            net.save_checkpoint(path, lr, args.momentum, args.weight_decay)
            print('[i] Checkpoint saved:', path)
        if writer is not None:
            writer.close()
        td.close()
        if os.environ.get('SSD_PRINT_CHECKSUM'):      # replica agreement check of the multi-rank tests
            print('[checksum] rank %d step %d params %.12e' % (rank, net.global_step, float(net.params_flat.double().sum())), flush=True)
    if world > 1:
        torch.distributed.barrier()
    return 0


if __name__ == '__main__':
    sys.exit(main())
