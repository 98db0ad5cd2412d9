"""CPU half of the feeder (reference training_data.py:137-195, data_queue.py:26-112):
the worker processes, the shared-memory slot ring and the ordering / reproducibility contract, with the
GPU half of a batch (upload + augmentation + label kernels) replaced by a host stand-in.  The real thing is
tests/test_gpu_feeder.py."""
import random

import numpy as np
import pytest

from oracle import boxes as ob
from ssd_tensorflow_amd import ssdutils
from ssd_tensorflow_amd.data_queue import DataQueue, WorkerError
from ssd_tensorflow_amd.training_data import TrainingData


def _prime(preset_name='vgg300'):
    preset = ssdutils.get_preset_by_name(preset_name)
    ssdutils.prime_anchor_table(preset, ob.anchors_abs(ob.anchors(ob.PRESETS[preset_name])))
    return preset


def _host_upload(arrays, gts, slot):
    """stand-in for TrainingData._upload: keeps a COPY of what would have gone to the GPU"""
    return {k: np.array(v) for k, v in arrays.items()}, [[tuple(b) for b in g] for g in gts]


class _Unreadable:
    """a data set whose file #7 cannot be read (module level: it travels to the worker processes)"""

    def __init__(self, orig):
        self.orig = orig

    def __call__(self, i):
        if i == 7:
            raise OSError('unreadable file %d' % i)
        return self.orig(i)


def _collect(td, gen, batch, workers):
    out = []
    for x, y, gt in gen(batch, workers):
        out.append((x, y, gt, td.global_count))
    return out


def _same(a, b):
    assert len(a) == len(b)
    for (xa, ya, ga, ca), (xb, yb, gb, cb) in zip(a, b):
        assert ca == cb
        if xa is None:
            assert xb is None and ga == gb == []
            continue
        assert xa.keys() == xb.keys()
        for k in xa:
            assert np.array_equal(xa[k], xb[k]), k
        assert ya == yb and ga == gb


def test_data_queue_slots_and_overflow():
    dq = DataQueue(1000, 2)
    a = np.arange(100, dtype=np.float32); b = np.arange(7, dtype=np.uint8)
    dq.put(('g', 0), 1, {'a': a, 'b': b}, ['boxes'])
    tag, slot, arrays, boxes = dq.get(timeout=5)
    assert tag == ('g', 0) and slot == 1 and boxes == ['boxes']
    assert np.array_equal(arrays['a'], a) and np.array_equal(arrays['b'], b)
    assert arrays['a'].base is not None                                  # a view of the slot, not a copy
    big = np.arange(2000, dtype=np.uint8)
    dq.put(('g', 1), 0, {'big': big}, [])                                # does not fit: travels through the pipe
    tag, slot, arrays, _ = dq.get(timeout=5)
    assert tag == ('g', 1) and np.array_equal(arrays['big'], big)
    with pytest.raises(ValueError):
        dq.put(('g', 2), 0, {'x': [1, 2, 3]}, [])                        # data_queue.py:64-65
    with pytest.raises(ValueError):
        dq.put(('g', 2), 0, {'x': np.zeros((4, 4))[:, ::2]}, [])
    dq.put_error(('g', 3), 1, 'boom')
    with pytest.raises(WorkerError) as e:
        dq.get(timeout=5)
    assert e.value.tag == ('g', 3) and e.value.arr_id == 1


@pytest.mark.parametrize('augment', [True, False])
def test_workers_produce_the_serial_generators_batches(augment):
    """num_workers = 0, 1 and 3: the same batches in the same order, twice (two epochs differ, a repeated epoch does
    not), whatever was drawn from `random` in between."""
    _prime()
    td = TrainingData(None, 'vgg300', num_train=22, num_valid=5, augment=augment, device_tensors=False)
    td._upload_hook = _host_upload
    try:
        td.epoch = 0
        random.seed(1)
        serial = _collect(td, td.train_generator, 4, 0)
        assert [len(g) for _, _, g, _ in serial] == [4, 4, 4, 4, 4, 2]
        random.seed(2); random.random()
        one = _collect(td, td.train_generator, 4, 1)
        three = _collect(td, td.train_generator, 4, 3)
        _same(serial, one); _same(serial, three)
        valid0 = _collect(td, td.valid_generator, 4, 0)
        valid2 = _collect(td, td.valid_generator, 4, 2)
        _same(valid0, valid2)
        td.epoch = 1
        other = _collect(td, td.train_generator, 4, 3)
        assert any(not np.array_equal(a[0][k], b[0][k]) if a[0][k].shape == b[0][k].shape else True
                   for a, b in zip(serial, other) for k in a[0])
        _same(other, _collect(td, td.train_generator, 4, 0))
    finally:
        td.close()


def test_rank_shards_with_workers_keep_lock_step():
    """Two ranks, a short last global batch that leaves rank 1 empty: same number of yields on both ranks, the
    global batch size travels with the batch it belongs to (not with the one being prefetched)."""
    _prime()
    tds = [TrainingData(None, 'vgg300', num_train=9, num_valid=2, augment=True, device_tensors=False, rank=r, world=2)
           for r in range(2)]
    try:
        outs = []
        for td in tds:
            td._upload_hook = _host_upload
            outs.append(_collect(td, td.train_generator, 2, 2))
        assert len(outs[0]) == len(outs[1]) == 3
        assert [c for *_, c in outs[0]] == [c for *_, c in outs[1]] == [4, 4, 1]
        assert len(outs[0][2][2]) == 1 and outs[1][2][0] is None and outs[1][2][2] == []
        for td, want in zip(tds, outs):
            _same(want, _collect(td, td.train_generator, 2, 0))
    finally:
        for td in tds:
            td.close()


def test_abandoned_epoch_and_failing_worker():
    _prime()
    td = TrainingData(None, 'vgg300', num_train=40, num_valid=4, augment=True, device_tensors=False)
    td._upload_hook = _host_upload
    try:
        want = _collect(td, td.train_generator, 4, 0)
        g = td.train_generator(4, 2)
        first = next(g)
        g.close()                                              # the consumer walks away mid-epoch
        again = _collect(td, td.train_generator, 4, 2)         # the pool is reused, no stale batch leaks in
        _same(want, again)
        assert np.array_equal(first[0]['packed'], want[0][0]['packed'])
        # a worker that raises: the consumer sees the error, the next epoch still works
        recipe = td._recipes['train']
        recipe.pool.close(); recipe.pool = None
        orig = recipe.sample_at
        recipe.sample_at = _Unreadable(orig)
        with pytest.raises(RuntimeError, match='unreadable file 7'):
            _collect(td, td.train_generator, 4, 2)
        recipe.pool.close(); recipe.pool = None
        recipe.sample_at = orig
        _same(want, _collect(td, td.train_generator, 4, 2))
    finally:
        td.close()


def test_batches_that_do_not_fit_a_slot_travel_through_the_pipe():
    """A slot sized for smaller images than the data set holds (a source whose annotations understate its files): the
    worker sends that batch through the result pipe instead of the slot; nothing changes for the consumer."""
    _prime()
    td = TrainingData(None, 'vgg300', num_train=12, num_valid=4, augment=True, device_tensors=False)
    td._upload_hook = _host_upload
    try:
        want = _collect(td, td.train_generator, 4, 0)
        td._max_image_bytes = 1000                      # every batch overflows its slot now
        got = _collect(td, td.train_generator, 4, 2)
        _same(want, got)
        assert td._recipes['train'].pool.slot_bytes < sum(v.nbytes for v in want[0][0].values())
    finally:
        td.close()


@pytest.mark.parametrize('start', ['forkserver', 'fork'])
def test_both_start_methods_and_a_killed_worker(start, monkeypatch):
    """The workers come from the fork server (default) or from a direct fork: the same batches either way.  A worker that
    is KILLED mid-epoch (out of memory, an operator's kill -9) cannot report anything: the consumer must get a RuntimeError within
    seconds instead of waiting for its batch forever."""
    import os
    import signal
    import time
    monkeypatch.setenv('SSD_FEEDER_START', start)
    _prime()
    td = TrainingData(None, 'vgg300', num_train=64, num_valid=4, augment=True, device_tensors=False)
    td._upload_hook = _host_upload
    try:
        want = _collect(td, td.train_generator, 4, 0)
        _same(want, _collect(td, td.train_generator, 4, 2))
        g = td.train_generator(4, 2)
        next(g)
        pool = td._recipes['train'].pool
        assert len(pool.workers) == 2 and all(w.is_alive() for w in pool.workers)
        for w in pool.workers:
            os.kill(w.pid, signal.SIGKILL)
        t0 = time.perf_counter()
        with pytest.raises(RuntimeError, match='a feeder worker process died'):
            for _ in g:
                pass
        assert time.perf_counter() - t0 < 20.0
    finally:
        for r in td._recipes.values():      # (the pool's workers are dead: nothing to hand back)
            if r.pool is not None:
                r.pool.workers = []
        td.close()


_KILL_DRIVER = r'''
import os, signal, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ['SSD_TEST_ROOT']); sys.path.insert(0, os.path.join(os.environ['SSD_TEST_ROOT'], 'tests'))
from oracle import boxes as ob
from ssd_tensorflow_amd import parallel, ssdutils
from ssd_tensorflow_amd.training_data import TrainingData
rank, local, world = parallel.init('gloo')
preset = ssdutils.get_preset_by_name('vgg300')
ssdutils.prime_anchor_table(preset, ob.anchors_abs(ob.anchors(ob.PRESETS['vgg300'])))
td = TrainingData(None, 'vgg300', num_train=96, num_valid=4, augment=True, device_tensors=False, rank=rank, world=world)
td._upload_hook = lambda arrays, gts, slot: ({k: np.array(v) for k, v in arrays.items()}, [len(g) for g in gts])
for k, (x, y, gt) in enumerate(td.train_generator(4, 2)):
    t = torch.ones(4)
    dist.all_reduce(t)                       # the step's collective: every rank must arrive
    if rank == 1 and k == 2:
        for w in td._recipes['train'].pool.workers:
            os.kill(w.pid, signal.SIGKILL)
print('rank %d finished the epoch' % rank, flush=True)
'''


def test_killed_worker_ends_a_two_rank_job(tmp_path):
    """Two ranks over gloo, a collective per step; rank 1's feeder workers are killed mid-epoch.  Rank 1 must fail with the feeder's
    RuntimeError (not hang waiting for a batch), and the launcher then takes rank 0 -- blocked in the next collective -- down with
    it: the job ends, non-zero, in seconds."""
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'kill_driver.py'
    script.write_text(_KILL_DRIVER)
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), str(script)], env=dict(os.environ, SSD_TEST_ROOT=root, OMP_NUM_THREADS='1'),
                       capture_output=True, text=True, timeout=240, cwd=root)
    took = time.perf_counter() - t0
    assert r.returncode != 0, r.stdout[-1500:]
    assert 'a feeder worker process died' in r.stderr, r.stderr[-3000:]
    assert 'rank 1 finished the epoch' not in r.stdout
    assert took < 120, took


def test_unpicklable_recipe_falls_back_to_a_direct_fork():
    """Something in the recipe that cannot travel to a fork-server worker (here: a lambda as the sample accessor) must not break
    the feeder: it warns and forks the workers directly, as round 3 did."""
    import warnings
    _prime()
    td = TrainingData(None, 'vgg300', num_train=12, num_valid=4, augment=True, device_tensors=False)
    td._upload_hook = _host_upload
    try:
        want = _collect(td, td.train_generator, 4, 0)
        recipe = td._recipes['train']
        orig = recipe.sample_at
        recipe.sample_at = lambda i: orig(i)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter('always')
            got = _collect(td, td.train_generator, 4, 2)
        _same(want, got)
        assert any('forking them directly' in str(w.message) for w in caught)
    finally:
        td.close()
