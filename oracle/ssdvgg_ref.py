"""
ORACLE (test infrastructure only) -- CPU fp32 restatement of the reference's TF1
graph (ssdvgg.py), written on torch-CPU ops + autograd.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path never does.

PARITY UNPINNED for this half: the arithmetic lives in TensorFlow 1.x (unpinned,
not installable here) and in the Udacity vgg.zip SavedModel (a network artifact,
not under /root/reference); the reference holds no test or golden vector for it
(SURVEY.md 4, 8c).  What this file follows, line by line:

  layer primitives          ssdvgg.py:42-84
  a-trous fc6/fc7           ssdvgg.py:231-292
  extra layers              ssdvgg.py:300-332
  l2 norm + feature maps    ssdvgg.py:335-350
  heads + output layout     ssdvgg.py:353-372
  loss                      ssdvgg.py:375-580
  optimizer                 ssdvgg.py:585-588, train.py:43-47

TF semantics restated: SAME padding pad_total = max((ceil(in/s)-1)*s
+ k_eff - in, 0), before = total//2, the extra at bottom/right; max-pool ignores
padded cells; l2_normalize eps 1e-12 inside max(); l2_loss = sum(x^2)/2; top_k
descending; Momentum without Nesterov; piecewise_constant uses values[i] for
step <= boundaries[i].
Assumed for the external VGG-16 (stated in DESIGN.md): 13x (3x3 s1 SAME conv + bias
+ relu), 4x 2x2 s2 SAME max-pool, input consumed as fed (BGR 0..255, no mean
subtraction), every VGG filter/bias trainable, L2Loss = l2_loss(filter).
The independent cross-check is tests/test_oracle_model.py (explicit numpy loss
restatement + finite differences).
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

from . import boxes as ob

VGG = [('conv1_1', 3, 64), ('conv1_2', 64, 64), 'pool',
       ('conv2_1', 64, 128), ('conv2_2', 128, 128), 'pool',
       ('conv3_1', 128, 256), ('conv3_2', 256, 256), ('conv3_3', 256, 256), 'pool',
       ('conv4_1', 256, 512), ('conv4_2', 512, 512), ('conv4_3', 512, 512), 'pool',
       ('conv5_1', 512, 512), ('conv5_2', 512, 512), ('conv5_3', 512, 512)]


def extra_layers(preset):
    """(name, k, cin, cout, stride, padding) -- ssdvgg.py:300-332."""
    big = len(preset['maps']) >= 7
    L = [('conv8_1', 1, 1024, 256, 1, 'SAME'), ('conv8_2', 3, 256, 512, 2, 'SAME'),
         ('conv9_1', 1, 512, 128, 1, 'SAME'), ('conv9_2', 3, 128, 256, 2, 'SAME'),
         ('conv10_1', 1, 256, 128, 1, 'SAME'),
         ('conv10_2', 3, 128, 256, 2 if big else 1, 'SAME' if big else 'VALID'),
         ('conv11_1', 1, 256, 128, 1, 'SAME'), ('conv11_2', 3, 128, 256, 1, 'VALID')]
    if big:
        L += [('conv12_1', 1, 256, 128, 1, 'SAME'), ('conv12_2', 3, 128, 256, 1, 'VALID')]
    return L


FMAP_CH = [512, 1024, 512, 256, 256, 256, 256]


def param_shapes(preset, num_classes=20):
    """Ordered {tf_variable_name: shape}; filters HWIO.  Order = forward order."""
    nv = num_classes + 5
    P = {}
    for l in VGG:
        if l == 'pool':
            continue
        P[l[0] + '/filter'] = (3, 3, l[1], l[2]); P[l[0] + '/biases'] = (l[2],)
    P['mod_conv6/filter'] = (3, 3, 512, 1024); P['mod_conv6/biases'] = (1024,)
    P['mod_conv7/filter'] = (1, 1, 1024, 1024); P['mod_conv7/biases'] = (1024,)
    for (n, k, ci, co, s, p) in extra_layers(preset):
        P[n + '/filter'] = (k, k, ci, co); P[n + '/biases'] = (co,)
    P['l2_norm_conv4_3/scale'] = (512,)
    for i, (fk, s, ars) in enumerate(preset['maps']):
        for j in range(2 + len(ars)):
            P[f'classifiers/classifier{i}_{j}/filter'] = (3, 3, FMAP_CH[i], nv)
            P[f'classifiers/classifier{i}_{j}/biases'] = (nv,)
    return P


def init_params(preset, num_classes=20, seed=42, bias_scale=0.0, alive=False):
    """Synthetic weights (no vgg.zip offline): Xavier-uniform filters
    (ssdvgg.py:46), zero biases (:47), scale = 20 (:336).  bias_scale > 0 draws
    small random biases instead.
    alive=True is the PARITY-TEST initialisation: with plain Xavier weights and a
    0..255 input the relu stack dies beyond mod_conv7 (every gradient there is exactly
    0 and would test nothing), so: He-uniform filters, conv1_1 scaled by 1/100 for the
    0..255 input, head filters by 0.05, small positive biases.  Every layer then keeps
    ~50 % of its units active at O(1) magnitude."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shp in param_shapes(preset, num_classes).items():
        if name.endswith('/filter'):
            kh, kw, ci, co = shp
            if alive:
                w = rng.uniform(-1, 1, shp) * math.sqrt(6.0 / (kh * kw * ci))
                if name.startswith('conv1_1'):
                    w /= 100.0
                if name.startswith('classifiers'):
                    w *= 0.05
                out[name] = w.astype(np.float32)
            else:
                lim = math.sqrt(6.0 / (kh * kw * ci + kh * kw * co))
                out[name] = rng.uniform(-lim, lim, shp).astype(np.float32)
        elif name.endswith('/scale'):
            out[name] = np.full(shp, 20.0, np.float32)
        elif alive:
            out[name] = rng.normal(0.02, 0.02, shp).astype(np.float32)
        else:
            out[name] = (rng.normal(0, bias_scale, shp) if bias_scale > 0 else np.zeros(shp)).astype(np.float32)
    return out


def init_params_lib(preset, num_classes=20, seed=42):
    """The weights libssdvgg_hip.so itself starts from (csrc/net.hip Net::init_weights: Xavier-uniform filters drawn
    from a splitmix64 stream in arena order, zero biases, scale 20) restated in numpy, so that the oracle can be run
    on exactly the benchmark's configuration.  tests/test_gpu_model.py checks it bit for bit against the library."""
    shapes = param_shapes(preset, num_classes)
    out = {}
    state = (np.uint64(seed) * np.uint64(0x2545F4914F6CDD1D) + np.uint64(1))
    gold = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over='ignore'):
        for name, shp in shapes.items():
            if name.endswith('/filter'):
                kh, kw, ci, co = shp
                n = kh * kw * ci * co
                s = state + gold * np.arange(1, n + 1, dtype=np.uint64)
                state = s[-1]
                z = s.copy()
                z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
                z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
                z ^= z >> np.uint64(31)
                u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
                lim = math.sqrt(6.0 / (float(kh * kw * ci) + float(kh * kw * co)))
                out[name] = ((u * 2.0 - 1.0) * lim).astype(np.float32).reshape(shp)
            elif name.endswith('/scale'):
                out[name] = np.full(shp, 20.0, np.float32)
            else:
                out[name] = np.zeros(shp, np.float32)
    return out


def graph(preset):
    """The layer graph as data, for layer-local checks: list of ops
    ('conv', name, input, k, stride, padding, dilation) | ('pool', name, input, k, stride) |
    ('l2norm', name, input) | ('head', map_index, input); tensors are named by the op that
    produces them ('image_input' is the network input, heads produce 'head<i>')."""
    ops = []
    cur = 'image_input'
    pi = 0
    for l in VGG:
        if l == 'pool':
            pi += 1
            ops.append(('pool', f'pool{pi}', cur, 2, 2)); cur = f'pool{pi}'
        else:
            ops.append(('conv', l[0], cur, 3, 1, 'SAME', 1)); cur = l[0]
    ops.append(('pool', 'mod_pool5', cur, 3, 1)); cur = 'mod_pool5'
    ops.append(('conv', 'mod_conv6', cur, 3, 1, 'SAME', 6)); cur = 'mod_conv6'
    ops.append(('conv', 'mod_conv7', cur, 1, 1, 'SAME', 1)); cur = 'mod_conv7'
    fmaps = ['norm_conv4_3', 'mod_conv7']
    for (n, k, ci, co, s, p) in extra_layers(preset):
        ops.append(('conv', n, cur, k, s, 'BR1' if n == 'conv12_2' else p, 1)); cur = n
        if n.endswith('_2'):
            fmaps.append(n)
    ops.append(('l2norm', 'norm_conv4_3', 'conv4_3'))
    for i in range(len(preset['maps'])):
        ops.append(('head', i, fmaps[i]))
    return ops


# ----------------------------------------------------------------------------
# TF-semantics primitives (NCHW torch tensors)
# ----------------------------------------------------------------------------
def same_pad(n, k, s, d=1):
    keff = (k - 1) * d + 1
    out = -(-n // s)
    tot = max((out - 1) * s + keff - n, 0)
    return tot // 2, tot - tot // 2, out


def conv2d_tf(x, w_hwio, stride=1, padding='SAME', dilation=1):
    """tf.nn.conv2d / atrous_conv2d.  x NCHW, w HWIO."""
    k = w_hwio.shape[0]
    w = w_hwio.permute(3, 2, 0, 1)
    if padding == 'SAME':
        pt, pb, _ = same_pad(x.shape[2], k, stride, dilation)
        pl, pr, _ = same_pad(x.shape[3], k, stride, dilation)
        x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, w, None, stride, 0, dilation)


def maxpool_tf(x, k, s):
    """tf.nn.max_pool SAME: padded cells never win."""
    pt, pb, _ = same_pad(x.shape[2], k, s)
    pl, pr, _ = same_pad(x.shape[3], k, s)
    x = F.pad(x, (pl, pr, pt, pb), value=float('-inf'))
    return F.max_pool2d(x, k, s)


def l2norm_tf(x, scale):
    """scale * x / sqrt(max(sum_c x^2, 1e-12))  -- ssdvgg.py:80-84."""
    ss = (x * x).sum(1, keepdim=True)
    return scale.view(1, -1, 1, 1) * x * torch.rsqrt(torch.clamp(ss, min=1e-12))


def forward(params, x_nhwc, preset, num_classes=20, keep=None):
    """params: {name: torch tensor}. x_nhwc [B,H,W,3].  Returns (out [B,A,C+5] raw
    head outputs, result [B,A,C+5]) -- ssdvgg.py:190-372.  `keep` (dict) collects
    NHWC copies of every intermediate activation when given."""
    nv = num_classes + 5
    x = x_nhwc.permute(0, 3, 1, 2)

    def cbr(x, name, stride=1, padding='SAME', dilation=1, relu=True):
        y = conv2d_tf(x, params[name + '/filter'], stride, padding, dilation)
        y = y + params[name + '/biases'].view(1, -1, 1, 1)
        y = F.relu(y) if relu else y
        if keep is not None:
            keep[name] = y.permute(0, 2, 3, 1)
            if y.requires_grad:
                y.retain_grad()
                keep['raw:' + name] = y
        return y

    pi = 0
    for l in VGG:
        if l == 'pool':
            pi += 1
            x = maxpool_tf(x, 2, 2)
            if keep is not None:
                keep[f'pool{pi}'] = x.permute(0, 2, 3, 1)
        else:
            x = cbr(x, l[0])
            if l[0] == 'conv4_3':
                conv4_3 = x
    x = maxpool_tf(x, 3, 1)                                     # mod_pool5, :234
    if keep is not None:
        keep['mod_pool5'] = x.permute(0, 2, 3, 1)
    x = cbr(x, 'mod_conv6', dilation=6)                          # :260
    x = cbr(x, 'mod_conv7')                                      # :287
    fmaps = [None, x]
    for (n, k, ci, co, s, p) in extra_layers(preset):
        if n == 'conv12_2':
            x = F.pad(x, (0, 1, 0, 1))                           # tf.pad after relu, :328
        x = cbr(x, n, s, p)
        if n.endswith('_2'):
            fmaps.append(x)
    fmaps[0] = l2norm_tf(conv4_3, params['l2_norm_conv4_3/scale'])  # :336
    if keep is not None:
        keep['norm_conv4_3'] = fmaps[0].permute(0, 2, 3, 1)
    outs = []
    for i, (fk, s, ars) in enumerate(preset['maps']):
        for j in range(2 + len(ars)):
            n = f'classifiers/classifier{i}_{j}'
            y = conv2d_tf(fmaps[i], params[n + '/filter']) + params[n + '/biases'].view(1, -1, 1, 1)
            outs.append(y.permute(0, 2, 3, 1).reshape(y.shape[0], fk * fk, nv))   # :63
    out = torch.cat(outs, 1)                                     # :365
    logits = out[:, :, :num_classes + 1]
    result = torch.cat([F.softmax(logits, -1), out[:, :, num_classes + 1:]], -1)  # :369-372
    return out, result


def l2_term(params):
    """sum over filters of sum(w^2)/2 -- filters only (ssdvgg.py:51,64,207,264,292)."""
    return sum((p * p).sum() / 2 for n, p in params.items() if n.endswith('/filter'))


def losses(out, labels, params, num_classes=20, weight_decay=0.0005):
    """ssdvgg.py:375-580.  out: raw head outputs [B,A,C+5]; labels [B,A,C+5]."""
    nc = num_classes + 1
    B, A = out.shape[0], out.shape[1]
    logits, locator = out[:, :, :nc], out[:, :, nc:]
    gt_cl, gt_loc = labels[:, :, :nc], labels[:, :, nc:]
    neg_n = (gt_cl[:, :, -1] != 0).sum(1)                       # count_nonzero, :409
    pos_n = A - neg_n
    pos_safe = torch.where(pos_n == 0, torch.full((B,), 10e-15), pos_n.float())
    pos_mask = gt_cl[:, :, -1] == 0
    ce = -(gt_cl * F.log_softmax(logits, -1)).sum(-1)            # :439
    pos_sum = torch.where(pos_mask, ce, torch.zeros_like(ce)).sum(-1)
    negatives = torch.where(~pos_mask, ce, torch.zeros_like(ce))
    top = torch.topk(negatives, A, dim=1, sorted=True)[0]         # :463 full descending sort
    kmax = torch.minimum(neg_n, 3 * pos_n)
    sel = torch.arange(A).view(1, -1) < kmax.view(-1, 1)
    neg_sum = torch.where(sel, top, torch.zeros_like(top)).sum(-1)
    conf = torch.where(pos_n == 0, torch.zeros(B), (pos_sum + neg_sum) / pos_safe)
    confidence = conf.mean()
    d = locator - gt_loc
    ad = d.abs()
    sl1 = torch.where(ad < 1.0, 0.5 * d * d, ad - 0.5).sum(-1)   # :68-71
    loc = torch.where(pos_mask, sl1, torch.zeros_like(sl1)).sum(-1)
    loc = torch.where(pos_n == 0, torch.zeros(B), loc / pos_safe)
    localization = loc.mean()
    l2 = weight_decay * l2_term(params)
    total = confidence + localization + l2
    return dict(total=total, localization=localization, confidence=confidence, l2=l2)


def piecewise_lr(step, boundaries, values):
    """tf.train.piecewise_constant: values[i] while step <= boundaries[i]."""
    for b, v in zip(boundaries, values):
        if step <= b:
            return v
    return values[len(boundaries)]


class RefModel:
    """A tiny stand-in for (tf.Session + SSDVGG graph): parameters, momentum
    accumulators, global step.  train_step mirrors train.py:262-266."""

    def __init__(self, preset_name, num_classes=20, params=None, seed=42):
        self.preset = ob.get_preset(preset_name)
        self.num_classes = num_classes
        src = params if params is not None else init_params(self.preset, num_classes, seed)
        self.params = {k: torch.tensor(np.asarray(v, np.float32)).requires_grad_(True) for k, v in src.items()}
        self.accum = {k: torch.zeros_like(v) for k, v in self.params.items()}
        self.step = 0
        self.set_optimizer()

    def set_optimizer(self, lr_values=(0.001,), lr_boundaries=(), momentum=0.9, weight_decay=0.0005):
        self.lr_values, self.lr_boundaries = list(lr_values), list(lr_boundaries)
        self.momentum, self.weight_decay = momentum, weight_decay

    def infer(self, x):
        with torch.no_grad():
            _, result = forward(self.params, torch.as_tensor(x, dtype=torch.float32), self.preset, self.num_classes)
        return result.numpy()

    def eval_step(self, x, y, keep=None):
        with torch.no_grad():
            out, result = forward(self.params, torch.as_tensor(x, dtype=torch.float32), self.preset, self.num_classes, keep)
            L = losses(out, torch.as_tensor(y, dtype=torch.float32), self.params, self.num_classes, self.weight_decay)
        return result.numpy(), {k: float(v) for k, v in L.items()}

    def grads(self, x, y):
        """(result, losses, {name: dLoss/dparam}) without touching the weights."""
        for p in self.params.values():
            p.grad = None
        out, result = forward(self.params, torch.as_tensor(x, dtype=torch.float32), self.preset, self.num_classes)
        L = losses(out, torch.as_tensor(y, dtype=torch.float32), self.params, self.num_classes, self.weight_decay)
        L['total'].backward()
        g = {k: (p.grad.detach().numpy().copy() if p.grad is not None else np.zeros(p.shape, np.float32))
             for k, p in self.params.items()}
        return result.detach().numpy(), {k: float(v) for k, v in L.items()}, g

    def train_step(self, x, y):
        """accum = m*accum + g; w -= lr*accum (MomentumOptimizer, no Nesterov)."""
        result, L, g = self.grads(x, y)
        lr = piecewise_lr(self.step, self.lr_boundaries, self.lr_values)
        with torch.no_grad():
            for k, p in self.params.items():
                self.accum[k].mul_(self.momentum).add_(torch.from_numpy(g[k]))
                p.sub_(lr * self.accum[k])
        self.step += 1
        return result, L

    def numpy_params(self):
        return {k: v.detach().numpy().copy() for k, v in self.params.items()}


# ----------------------------------------------------------------------------
# explicit numpy restatement of the loss + its gradient (independent of autograd)
# ----------------------------------------------------------------------------
def loss_numpy(out, labels, num_classes=20):
    """float64 numpy restatement of ssdvgg.py:400-560 and the gradient w.r.t. the
    raw head outputs, with the tie rule stated: among equal cross-entropies at
    the k-th place, the LOWER anchor index is kept (TF top_k).  Returns
    (confidence, localization, dOut [B,A,C+5], selected-negative mask)."""
    out = np.asarray(out, np.float64); y = np.asarray(labels, np.float64)
    nc = num_classes + 1
    B, A, _ = out.shape
    z = out[:, :, :nc]
    m = z.max(-1, keepdims=True)
    lse = m[..., 0] + np.log(np.exp(z - m).sum(-1))
    p = np.exp(z - lse[..., None])
    ce = (y[:, :, :nc] * (lse[..., None] - z)).sum(-1)
    pos = y[:, :, nc - 1] == 0
    neg_n = (~pos).sum(1); pos_n = A - neg_n
    conf = np.zeros(B); loc = np.zeros(B)
    d_out = np.zeros_like(out)
    selmask = np.zeros((B, A), bool)
    d = out[:, :, nc:] - y[:, :, nc:]
    sl1 = np.where(np.abs(d) < 1, 0.5 * d * d, np.abs(d) - 0.5).sum(-1)
    for b in range(B):
        if pos_n[b] == 0:
            continue
        k = min(neg_n[b], 3 * pos_n[b])
        negv = np.where(pos[b], 0.0, ce[b])
        order = np.lexsort((np.arange(A), -negv))[:k]        # value desc, index asc
        picked = np.zeros(A, bool); picked[order] = True
        picked &= ~pos[b]                                       # zeros sitting on positives carry no gradient
        selmask[b] = picked
        conf[b] = (ce[b][pos[b]].sum() + negv[order].sum()) / pos_n[b]
        loc[b] = sl1[b][pos[b]].sum() / pos_n[b]
        w = (pos[b] | picked).astype(np.float64) / pos_n[b] / B
        d_out[b, :, :nc] = (p[b] - y[b, :, :nc]) * w[:, None]
        d_out[b, :, nc:] = np.clip(d[b], -1, 1) * (pos[b].astype(np.float64) / pos_n[b] / B)[:, None]
    return conf.mean(), loc.mean(), d_out, selmask


# ----------------------------------------------------------------------------
# synthetic inputs
# ----------------------------------------------------------------------------
def synth_images(rng, b, preset):
    H, W = preset['image_size'][1], preset['image_size'][0]
    return rng.integers(0, 256, (b, H, W, 3)).astype(np.float32)


def synth_gt(rng):
    n = int(rng.integers(1, 6))
    w = rng.uniform(0.1, 0.6, n); h = rng.uniform(0.1, 0.6, n)
    cx = rng.uniform(w / 2, 1 - w / 2); cy = rng.uniform(h / 2, 1 - h / 2)
    return np.stack([cx, cy, w, h], 1), rng.integers(0, 20, n)


def synth_batch(rng, b, preset, num_classes=20, anch=None, anch_abs=None):
    """x [b,H,W,3] f32 0..255; y [b,A,C+5] via the label oracle, redrawn until
    at least one positive anchor (training_data.py:92-98)."""
    if anch is None:
        anch = ob.anchors(preset); anch_abs = ob.anchors_abs(anch)
    x = synth_images(rng, b, preset)
    ys, gts = [], []
    for _ in range(b):
        for _try in range(50):
            g, c = synth_gt(rng)
            vec = ob.encode_labels(g, c, preset, num_classes, anch, anch_abs)
            if np.count_nonzero(vec[:, num_classes]) < vec.shape[0]:
                break
        ys.append(vec); gts.append((g, c))
    return x, np.stack(ys), gts


def decimate_fc_loops(fc6_w, fc6_b, fc7_w, fc7_b):
    """The weight decimation of ssdvgg.py:245-253 and :273-280, restated with the reference's own loop
    structure (float64 zeros, element-wise copies).  Used to check ssd_tensorflow_amd/weights.py."""
    mod_w6 = np.zeros((3, 3, 512, 1024)); mod_b6 = np.zeros(1024)
    for i in range(1024):
        mod_b6[i] = fc6_b[4 * i]
        for h in range(3):
            for w in range(3):
                mod_w6[h, w, :, i] = fc6_w[3 * h, 3 * w, :, 4 * i]
    mod_w7 = np.zeros((1, 1, 1024, 1024)); mod_b7 = np.zeros(1024)
    for i in range(1024):
        mod_b7[i] = fc7_b[4 * i]
        mod_w7[:, :, :, i] = fc7_w[:, :, 0:4096:4, 4 * i]      # for j: mod_w[:, :, j, i] = orig_w[:, :, 4j, 4i]
    return mod_w6, mod_b6, mod_w7, mod_b7
