"""GPU debugging aid: does the Winograd step compute the DIRECT step's gradients at a TRAINED state (activations and filters of a
real run, not N(0,1) test data)?  Trains the shapes set for a few hundred steps (direct kernels), then runs one forward + backward
of the same batch on two handles -- SSD_WINOGRAD=0 and =7 -- from the checkpoint and prints every variable's gradient error."""
import os, sys, glob, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import boxes as ob, ssdvgg_ref as ref
from ssd_tensorflow_amd import train
from ssd_tensorflow_amd.ssdvgg import SSDVGG, Session

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tmp = tempfile.mkdtemp(prefix='wino_probe_')
os.environ['SSD_WINOGRAD'] = '0'
rc = train.main(['--name', tmp + '/run', '--tensorboard-dir', tmp + '/tb', '--data-dir', 'shapes', '--synthetic-train', '1024', '--synthetic-valid', '128',
                 '--num-workers', '4', '--batch-size', '32', '--checkpoint-interval', '1000', '--lr-values', '0.0003;0.00075;0.0001', '--lr-boundaries', '96;768',
                 '--epochs', str(epochs), '--dtype', 'f32'])
assert rc == 0
ck = sorted(glob.glob(tmp + '/run/*.npz'))[-1]
w = {k: v for k, v in np.load(ck).items()}
print('checkpoint', ck, len(w), 'arrays')
preset = ob.get_preset('vgg300')
x, y, _ = ref.synth_batch(np.random.default_rng(7), 32, preset)
xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
grads = {}
acts = {}
sess = Session(0)
for mode in ('0', '7', '1', '3', '5'):
    os.environ['SSD_WINOGRAD'] = mode
    net = SSDVGG(sess, 'vgg300')
    net.build_from_vgg(None, 20, max_batch=32)
    known = dict(net.variables())
    net.load_variables({k: v for k, v in w.items() if k in known})
    net.build_optimizer(1e-4)
    net.forward_backward_dev(xd, yd)
    torch.cuda.synchronize()
    grads[mode] = net.save_gradients()
    acts[mode] = {n: net.activation(n, 32) for n in ('conv2_1', 'conv3_2', 'conv4_3', 'conv5_3', 'grad:conv4_3', 'grad:conv3_2', 'grad:conv2_1')}
    print('mode', mode, 'losses', net.get_losses())
    net.close()
for mode in ('7', '1', '3', '5'):
    print('---- SSD_WINOGRAD=%s against 0' % mode)
    for n, a in acts[mode].items():
        r = acts['0'][n]
        print('  act %-16s relL2 %.3e  maxrel %.3e  absmax %.3e' % (n, np.sqrt(((a - r) ** 2).sum() / ((r ** 2).sum() + 1e-300)), np.abs(a - r).max() / (np.abs(r).max() + 1e-300), np.abs(r).max()))
    for k, g in grads[mode].items():
        r = grads['0'][k]
        e = np.sqrt(((g.astype(np.float64) - r) ** 2).sum() / ((r.astype(np.float64) ** 2).sum() + 1e-300))
        if e > 1e-5 or 'conv3_2' in k or 'conv5_2' in k:
            print('  grad %-36s relL2 %.3e  maxrel %.3e' % (k, e, np.abs(g - r).max() / (np.abs(r).max() + 1e-300)))
