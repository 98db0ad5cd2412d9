"""GPU debugging aid: per-layer error of d(loss)/d(pre-activation) vs the autograd oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch
from oracle import boxes as ob, ssdvgg_ref as ref
from ssd_tensorflow_amd.ssdvgg import SSDVGG, Session

pname = sys.argv[1] if len(sys.argv) > 1 else 'vgg300'
b = int(sys.argv[2]) if len(sys.argv) > 2 else 2
preset = ob.get_preset(pname)
w = ref.init_params(preset, 20, seed=42, bias_scale=0.01)
m = ref.RefModel(pname, params=w)
sess = Session(0); net = SSDVGG(sess, pname); net.build_from_vgg(None, 20, max_batch=b, weights=w)
net.build_optimizer(0.001)
rng = np.random.default_rng(1234)
x, y, _ = ref.synth_batch(rng, b, preset)
keep = {}
for p in m.params.values():
    p.grad = None
out, result = ref.forward(m.params, torch.as_tensor(x), preset, 20, keep)
L = ref.losses(out, torch.as_tensor(y), m.params, 20, 0.0005)
L['total'].backward()
net.forward_backward_dev(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
torch.cuda.synchronize()


def stats(tag, got, want):
    d = got.astype(np.float64) - want.astype(np.float64)
    rl2 = np.sqrt((d ** 2).sum() / ((want.astype(np.float64) ** 2).sum() + 1e-300))
    mx = np.abs(d).max() / (np.abs(want).max() + 1e-300)
    nbad = int((np.abs(d) > 1e-3 * np.abs(want).max()).sum())
    print(f'{tag:<28s} relL2 {rl2:.3e}  maxrel {mx:.3e}  n(|err|>1e-3 max) {nbad} / {d.size}  nnz got {np.count_nonzero(got)} want {np.count_nonzero(want)}')


names = [k[4:] for k in keep if k.startswith('raw:')]
for n in reversed(names):
    yt = keep['raw:' + n]
    if yt.grad is None:
        continue
    want = (yt.grad * (yt > 0).float()).permute(0, 2, 3, 1).numpy()
    got = net.activation('grad:' + n, b)
    stats('dpre ' + n, got, want)
g = net.save_gradients()
for k, p in m.params.items():
    stats('grad ' + k, g[k], p.grad.numpy()) if ('conv5' in k or 'conv4_3' in k or 'mod_conv6' in k) else None
