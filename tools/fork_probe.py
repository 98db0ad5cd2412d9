#!/usr/bin/env python3
"""How long does the feeder's fork take in a process that has initialised the GPU?  (tools/fork_probe.py on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiprocessing as mp


def child(q):
    q.put(os.getpid())


def fork_ms(n=3):
    ctx = mp.get_context('fork')
    out = []
    for _ in range(n):
        q = ctx.Queue()
        t0 = time.perf_counter()
        p = ctx.Process(target=child, args=(q,)); p.start()
        t1 = time.perf_counter()
        q.get(); p.join()
        out.append(((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
    return out


print('before torch / HIP:', fork_ms())
import torch
from ssd_tensorflow_amd.ssdvgg import SSDVGG, Session
x = torch.zeros(1 << 20, device='cuda'); torch.cuda.synchronize()
print('after HIP init:', fork_ms())
sess = Session(0); net = SSDVGG(sess, 'vgg300'); net.build_from_vgg(None, 20, max_batch=32)
torch.cuda.synchronize()
print('with a batch-32 training net (7 GB of HBM mapped):', fork_ms())
from ssd_tensorflow_amd.training_data import TrainingData
td = TrainingData(None, 'vgg300', num_train=64, num_valid=8, augment=True)
t0 = time.perf_counter()
g = td.train_generator(32, 4); next(g)
print('first prefetched batch (pool of 4 forked, slots pinned, ring allocated): %.1f ms' % ((time.perf_counter() - t0) * 1e3))
for _ in g:
    pass
t0 = time.perf_counter()
g = td.train_generator(32, 4); next(g)
print('first batch of the next epoch (pool reused): %.1f ms' % ((time.perf_counter() - t0) * 1e3))
g.close(); td.close(); sess.close()
