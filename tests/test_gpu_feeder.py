"""-m gpu, SURVEY.md 8a A7: the feeder that runs beside the step (the purpose of the reference's worker processes
and slot ring, training_data.py:137-195 / data_queue.py:26-112) and the loss fetch that does not wait for the step.

  * batches prefetched by worker processes into the device slot ring are bit-identical to the serial generator's;
  * a consumer that is slower or faster than the feeder never sees a slot that is being rewritten;
  * the losses read one step late (ssd_get_losses_step) are the losses the synchronous fetch returns;
  * train.py with --num-workers ends with the same weights as without, on one rank and on two."""
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

from ssd_tensorflow_amd import train
from ssd_tensorflow_amd.ssdvgg import SSDVGG, Session
from ssd_tensorflow_amd.training_data import TrainingData

pytestmark = pytest.mark.gpu


def _drain(gen, batch, workers, delay=0.0):
    out = []
    for x, y, gt in gen(batch, workers):
        if delay:
            time.sleep(delay)
        out.append((x.cpu().numpy().copy(), y.cpu().numpy().copy(), gt))
    return out


@pytest.mark.parametrize('augment', [True, False])
def test_prefetched_batches_are_the_serial_batches(augment):
    td = TrainingData(None, 'vgg300', num_train=14, num_valid=6, augment=augment)
    try:
        for epoch in (0, 1):
            td.epoch = epoch
            serial = _drain(td.train_generator, 4, 0)
            assert [len(g) for _, _, g in serial] == [4] * 3 + [2]
            runs = [_drain(td.train_generator, 4, 2)]
            if epoch == 1:
                runs.append(_drain(td.train_generator, 4, 2, delay=0.05))      # the feeder runs ahead and must wait for free slots
            for got in runs:
                assert len(got) == len(serial)
                for (xa, ya, ga), (xb, yb, gb) in zip(serial, got):
                    assert np.array_equal(xa, xb) and np.array_equal(ya, yb) and ga == gb
        v0 = _drain(td.valid_generator, 4, 0); v2 = _drain(td.valid_generator, 4, 2)
        for (xa, ya, ga), (xb, yb, gb) in zip(v0, v2):
            assert np.array_equal(xa, xb) and np.array_equal(ya, yb) and ga == gb
    finally:
        td.close()


def test_train_and_valid_generators_live_at_once():
    """Validating in the middle of a training epoch (the reference's two queues are independent, training_data.py:147-195):
    each data set has its own device slot ring and feeder stream, so interleaved generators hand out exactly the batches they
    hand out one after the other -- and a second generator of the SAME data set is refused."""
    td = TrainingData(None, 'vgg300', num_train=16, num_valid=12, augment=True)
    try:
        t_alone = _drain(td.train_generator, 4, 2)
        v_alone = _drain(td.valid_generator, 4, 2)
        gt_, gv_ = td.train_generator(4, 2), td.valid_generator(4, 2)
        t_mixed, v_mixed = [], []
        for k in range(4):
            x, y, g = next(gt_)
            torch.cuda.synchronize()
            if k < 3:
                xv, yv, gv = next(gv_)
                v_mixed.append((xv.cpu().numpy().copy(), yv.cpu().numpy().copy(), gv))
            t_mixed.append((x.cpu().numpy().copy(), y.cpu().numpy().copy(), g))      # read AFTER the other data set's upload
            if k == 1:
                with pytest.raises(RuntimeError, match='one train generator at a time'):
                    next(td.train_generator(4, 2))
        for a, b in ((t_alone, t_mixed), (v_alone, v_mixed)):
            assert len(a) == len(b)
            for (xa, ya, ga), (xb, yb, gb) in zip(a, b):
                assert np.array_equal(xa, xb) and np.array_equal(ya, yb) and ga == gb
        gt_.close(); gv_.close()
    finally:
        td.close()


def test_slot_is_not_rewritten_under_the_consumer():
    """The consumer enqueues a long kernel sequence that READS the batch (a training step) and only then asks for the next
    batch: the feeder may refill that slot only behind those kernels.  The step's losses must equal the ones of the same
    batches fed from private copies."""
    td = TrainingData(None, 'vgg300', num_train=24, num_valid=4, augment=True)
    sess = Session(0)
    try:
        def run(workers, private):
            net = SSDVGG(sess, td.preset)
            net.build_from_vgg(None, 20, max_batch=4, seed=5)
            net.build_optimizer(learning_rate=1e-4)
            losses = []
            for x, y, gt in td.train_generator(4, workers):
                if private:
                    x, y = x.clone(), y.clone()
                sess.run(net.optimizer, feed_dict={net.image_input: x, net.labels: y})
                losses.append(None)
                if len(losses) > 1:
                    losses[-2] = net.get_losses_step(1)
            losses[-1] = net.get_losses_step(0)
            return losses, float(net.params_flat.double().sum())
        want, wsum = run(0, True)
        got, gsum = run(3, False)
        assert want == got and wsum == gsum
    finally:
        sess.close(); td.close()


def test_losses_one_step_late_equal_the_synchronous_fetch():
    td = TrainingData(None, 'vgg300', num_train=12, num_valid=4)
    sess = Session(0)
    try:
        batches = [(x.clone(), y.clone()) for x, y, _ in td.train_generator(4)]
        nets = []
        for _ in range(2):
            n = SSDVGG(sess, td.preset); n.build_from_vgg(None, 20, max_batch=4, seed=9); n.build_optimizer(learning_rate=1e-4); nets.append(n)
        sync = [sess.run([nets[0].losses, nets[0].optimizer], feed_dict={nets[0].image_input: x, nets[0].labels: y})[0] for x, y in batches]
        late = []
        for k, (x, y) in enumerate(batches):
            sess.run(nets[1].optimizer, feed_dict={nets[1].image_input: x, nets[1].labels: y})
            if k:
                late.append(nets[1].get_losses_step(1))
        late.append(nets[1].get_losses_step(0))
        assert late == sync
        assert nets[1].get_losses_step(1) == sync[-2] and nets[1].get_losses_step(2) == sync[-3]      # the ring keeps three steps
        with pytest.raises(RuntimeError):
            nets[1].get_losses_step(3)
        # a forward pass without update (validation) takes a slot as well
        sess.run(nets[1].eval_op, feed_dict={nets[1].image_input: batches[0][0], nets[1].labels: batches[0][1]})
        ev = sess.run(nets[0].losses, feed_dict={nets[0].image_input: batches[0][0], nets[0].labels: batches[0][1]})
        assert nets[1].get_losses_step(0) == ev and nets[1].get_losses_step(1) == sync[-1]
    finally:
        sess.close(); td.close()


def test_train_driver_with_workers_matches_without(tmp_path, capsys):
    common = ['--epochs', '2', '--batch-size', '4', '--synthetic-train', '14', '--synthetic-valid', '5', '--augment', 'true',
              '--checkpoint-interval', '5', '--lr-values', '0.0001', '--lr-boundaries', '', '--tensorboard-dir', str(tmp_path / 'tb')]
    a, b = str(tmp_path / 'w0'), str(tmp_path / 'w3')
    assert train.main(['--name', a, '--num-workers', '0'] + common) == 0
    out0 = capsys.readouterr().out
    assert train.main(['--name', b, '--num-workers', '3'] + common) == 0
    out3 = capsys.readouterr().out
    assert '[i] Number of workers:     3' in out3
    pick = lambda o: [l for l in o.splitlines() if l.startswith(('[i] Train', '[i] Valid', '[i] mAP'))]
    assert pick(out0) == pick(out3) and len(pick(out0)) == 6          # '[i] Training...' + 2 x (Train, Valid) + mAP
    ca, cb = np.load(a + '/final.npz'), np.load(b + '/final.npz')
    for k in ca.files:
        assert np.array_equal(ca[k], cb[k]), k


def test_two_ranks_with_workers_stay_in_lock_step(tmp_path):
    """Two ranks on GPU 0 over gloo, each with its own worker processes and slot ring, a short last batch whose second
    shard is empty: identical replicas, and the same weights as the two ranks without workers."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sums = {}
    for workers in (0, 2):
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        env = dict(os.environ, SSD_FORCE_DEVICE='0', SSD_DIST_BACKEND='gloo', SSD_PRINT_CHECKSUM='1', PYTHONPATH=root)
        r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                            '--master-port', str(port), '-m', 'ssd_tensorflow_amd.train', '--name', str(tmp_path / ('dp%d' % workers)),
                            '--batch-size', '2', '--epochs', '2', '--synthetic-train', '9', '--synthetic-valid', '3', '--augment', 'true',
                            '--num-workers', str(workers), '--checkpoint-interval', '5', '--lr-values', '0.0001', '--lr-boundaries', '',
                            '--tensorboard-dir', str(tmp_path / 'tb')], env=env, cwd=root, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        got = dict(re.findall(r'\[checksum\] rank (\d) step \d+ params (\S+)', r.stdout))
        assert set(got) == {'0', '1'} and got['0'] == got['1'], r.stdout[-2000:]
        sums[workers] = (got['0'], [l for l in r.stdout.splitlines() if l.startswith(('[i] Train', '[i] Valid', '[i] mAP'))])
    assert sums[0] == sums[2]


def test_batch_larger_than_its_slot_takes_the_serial_upload():
    """A batch that does not fit its host slot comes through the result pipe; the feeder thread then uploads it with the
    serial path (own copies, label encoder with host arrays) into the same device slot ring: same tensors."""
    td = TrainingData(None, 'vgg300', num_train=10, num_valid=4, augment=True)
    try:
        serial = _drain(td.train_generator, 4, 0)
        td._max_image_bytes = 1000
        got = _drain(td.train_generator, 4, 2)
        assert len(got) == len(serial) == 3
        for (xa, ya, ga), (xb, yb, gb) in zip(serial, got):
            assert np.array_equal(xa, xb) and np.array_equal(ya, yb) and ga == gb
    finally:
        td.close()
