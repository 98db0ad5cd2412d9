"""Host-side mirror of the reference's class SSDVGG (ssdvgg.py:87-649) over the C ABI of
libssdvgg_hip.so.  Same method names, argument meaning and attribute names, so the
reference's train.py / infer.py loops read unchanged:

    net = SSDVGG(sess, preset)
    net.build_from_vgg(vgg_dir, num_classes)
    net.build_optimizer(learning_rate=..., weight_decay=..., momentum=..., global_step=...)
    result, loss_batch, _ = sess.run([net.result, net.losses, net.optimizer],
                                     feed_dict={net.image_input: x, net.labels: y})

`Session` stands in for tf.Session: it owns no graph, it routes run() to the handle.
PyTorch appears only as the allocator of the parameter / gradient / momentum arenas
(so torch.distributed can all-reduce the gradient arena in place) and as stream owner.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import lib, check, np_ptr
from .ssdutils import get_preset_by_name, SSD_PRESETS, detect_batch

LOSS_NAMES = ('total', 'localization', 'confidence', 'l2')      # ssdvgg.py:594-599


class _Token:
    """An opaque stand-in for a TF tensor/op handle; only identity matters."""
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return f'<ssd token {self.name}>'


class _DetectionList:
    """The detections of one pass as a sequence of per-image dicts {conf, cls, idx, box} (what detect_last returned as
    a list).  The arrays are VIEWS of the handle's pinned host mirror of that pass: an image's dict is built when it
    is indexed, so collecting a batch costs the host one event wait and no per-image work it does not ask for.  Valid
    until the second-next pass is launched (two slots alternate); copy what must live longer."""

    def __init__(self, count, conf, cls, idx, box, out_cap, net=None, serial=0):
        self.count, self.conf, self.cls, self.idx, self.box, self.out_cap = count, conf, cls, idx, box, out_cap
        self.net, self.serial = net, serial      # the views point into pinned memory the handle owns: checked on every access

    def __len__(self):
        return len(self.count)

    def _check_alive(self):
        """The slot behind these views is rewritten by the second-next pass, moved (freed) when it has to grow and freed with the
        handle: an access after any of these is an error here, not a read of freed memory."""
        net = self.net
        if net is None:
            return
        if net._h is None:
            raise RuntimeError('these detections belong to a closed SSDVGG handle: copy what must outlive it')
        if net._det_serial - self.serial not in (0, 1):
            raise RuntimeError('these detections were overwritten: only the two most recent passes are kept (copy what must live longer)')

    def __getitem__(self, i):
        self._check_alive()
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        n = min(int(self.count[i]), self.out_cap)
        return dict(conf=self.conf[i, :n], cls=self.cls[i, :n], idx=self.idx[i, :n], box=self.box[i, :n])

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class _Detections:
    """Ticket of SSDVGG.detect_last_launch."""
    def __init__(self, net, serial, b, out_cap):
        self.net, self.serial, self.b, self.out_cap = net, serial, b, out_cap

    def get(self):
        which = self.net._det_serial - self.serial
        if which not in (0, 1):
            raise RuntimeError('these detections were overwritten: only the two most recent passes are kept')
        ptrs = [C.c_void_p() for _ in range(5)]
        b = C.c_int(); out_cap = C.c_int()
        check(lib.ssd_detect_host(self.net._h, which, *[C.byref(p) for p in ptrs], C.byref(b), C.byref(out_cap)))
        b, oc = b.value, out_cap.value
        key = (ptrs[0].value, ptrs[4].value, b, oc)
        views = self.net._det_views.get(key)
        if views is None:      # the slots' host arrays move only when a slot grows: the numpy views are built once per layout

            def view(p, ctype, dtype, shape):
                n = int(np.prod(shape))
                return np.frombuffer((ctype * n).from_address(p.value), dtype=dtype).reshape(shape)
            views = (view(ptrs[0], C.c_int, np.int32, (b,)), view(ptrs[1], C.c_float, np.float32, (b, oc)),
                     view(ptrs[2], C.c_int, np.int32, (b, oc)), view(ptrs[3], C.c_int, np.int32, (b, oc)),
                     view(ptrs[4], C.c_int, np.int32, (b, oc, 4)))
            if len(self.net._det_views) > 8:
                self.net._det_views.clear()
            self.net._det_views[key] = views
        return _DetectionList(*views, oc, net=self.net, serial=self.serial)


class LearningRate:
    """compute_lr's result (train.py:43-47): piecewise-constant values over global_step."""
    def __init__(self, values, boundaries):
        values = [float(v) for v in values]; boundaries = [int(b) for b in boundaries]
        if len(values) != len(boundaries) + 1:
            raise ValueError('lr_values must hold one more entry than lr_boundaries')
        self.values, self.boundaries = values, boundaries


class Session:
    """Minimal tf.Session look-alike: `with Session() as sess: sess.run(fetches, feed_dict)`."""
    def __init__(self, device=None):
        self.device = int(os.environ.get('LOCAL_RANK', '0')) if device is None else int(device)
        self.nets = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self):
        for n in self.nets:
            n.close()
        self.nets = []

    def run(self, fetches, feed_dict=None):
        single = not isinstance(fetches, (list, tuple))
        flist = [fetches] if single else list(fetches)
        net = None
        for f in flist:
            owner = getattr(f, 'owner', None) if not isinstance(f, dict) else next(iter(f.values())).owner
            net = owner or net
        if net is None:
            raise ValueError('nothing runnable in fetches')
        out = net._run(flist, feed_dict or {})
        return out[0] if single else out


class SSDVGG:
    def __init__(self, session, preset):
        self.preset = get_preset_by_name(preset) if isinstance(preset, str) else preset
        self.session = session
        self._h = None
        self.__built = False
        self.__build_names()
        if session is not None:
            session.nets.append(self)

    # ------------------------------------------------------------------ construction
    def build_from_vgg(self, vgg_dir, num_classes, a_trous=True, progress_hook='tqdm', max_batch=32,
                       training=True, seed=42, weights=None, dtype='f32'):
        """ssdvgg.py:96-118.  There is no vgg.zip offline: the VGG-16 trunk starts from
        Xavier-uniform synthetic weights (seed) unless `weights` ({tf_name: array}) or
        `<vgg_dir>/vgg16_ssd.npz` supplies them.  dtype 'f32' (default) or 'bf16' (bf16 activations and
        filter mirrors on the bf16 matrix cores; fp32 master weights, loss and optimizer)."""
        if not a_trous:
            raise NotImplementedError('only the default a_trous=True variant (ssdvgg.py:231) is built')
        self.num_classes = num_classes + 1
        self.num_vars = num_classes + 5
        self._create(num_classes, max_batch, training, seed, dtype)
        path = os.path.join(vgg_dir, 'vgg16_ssd.npz') if vgg_dir else None
        if weights is None and path and os.path.exists(path):
            weights = dict(np.load(path))
        if weights:
            self.load_variables(weights)
        self.__built = True

    def build_from_metagraph(self, metagraph_file, checkpoint_file, max_batch=32, training=False, dtype='f32'):
        """ssdvgg.py:120-130: restore a trained net.  checkpoint_file is an .npz written by
        save_checkpoint (variables under the reference's TF names + preset/num_classes)."""
        ck = np.load(checkpoint_file, allow_pickle=False)
        num_classes = int(ck['__num_classes__'])
        self.num_classes = num_classes + 1
        self.num_vars = num_classes + 5
        self._create(num_classes, max_batch, training, 0, dtype)
        self.load_variables({k: ck[k] for k in ck.files if not k.startswith('__')})
        self._ckpt = ck
        self.__built = True

    def _create(self, num_classes, max_batch, training, seed, dtype='f32'):
        import torch
        if dtype not in ('f32', 'bf16'):
            raise ValueError("dtype must be 'f32' or 'bf16', got %r" % (dtype,))
        self.dtype = dtype
        dev = self.session.device if self.session is not None else 0
        self.device = dev
        self.max_batch = int(max_batch)
        self.training = bool(training)
        n = lib.ssd_arena_floats(self.preset.name.encode(), num_classes)
        if n == 0:
            raise RuntimeError(_lib.last_error())
        tdev = torch.device('cuda', dev)
        self.params_flat = torch.empty(n, dtype=torch.float32, device=tdev)
        self.grads_flat = torch.zeros(n, dtype=torch.float32, device=tdev) if training else None
        self.momentum_flat = torch.zeros(n, dtype=torch.float32, device=tdev) if training else None
        h = C.c_void_p()
        check(lib.ssd_create_dtype(self.preset.name.encode(), num_classes, self.max_batch, dev, int(training), seed,
                                   self.params_flat.data_ptr(),
                                   self.grads_flat.data_ptr() if training else None,
                                   self.momentum_flat.data_ptr() if training else None,
                                   1 if dtype == 'bf16' else 0, C.byref(h)))
        self._h = h
        fl = C.c_size_t(); ff = C.c_size_t()
        check(lib.ssd_arenas(h, None, None, None, C.byref(fl), C.byref(ff)))
        self.arena_floats, self.filter_floats = fl.value, ff.value
        self._n_classes = num_classes
        # tokens the callers feed / fetch (ssdvgg.py:128-150,193-194,378,593-599)
        self.image_input = self._tok('image_input:0')
        self.keep_prob = self._tok('keep_prob:0')
        self.result = self._tok('result/result:0')
        self.labels = None
        self.losses = None
        self.optimizer = None
        self.eval_op = None

    def _tok(self, name):
        t = _Token(name)
        t.owner = self
        return t

    def close(self):
        if self._h is not None:
            lib.ssd_destroy(self._h)
            self._h = None
            if hasattr(self, '_det_views'):
                self._det_views.clear()      # (views of pinned memory that ssd_destroy has just freed)
            self._bf16_messages = None       # (parallel.Bf16Message buffers of this handle's gradient arena)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ optimizer
    def build_optimizer(self, learning_rate=0.001, weight_decay=0.0005, momentum=0.9, global_step=None):
        """ssdvgg.py:375-599.  learning_rate: float or LearningRate (train.py:43-47)."""
        if not self.training:
            raise RuntimeError('build_from_* was called with training=False')
        lr = learning_rate if isinstance(learning_rate, LearningRate) else LearningRate([learning_rate], [])
        vals = np.array(lr.values, np.float32)
        bnds = np.array(lr.boundaries + [0], np.int64)
        check(lib.ssd_set_optimizer(self._h, np_ptr(vals), np_ptr(bnds), len(lr.values), float(momentum), float(weight_decay)))
        if global_step is not None:
            check(lib.ssd_set_global_step(self._h, int(global_step)))
        self.labels = self._tok('labels:0')
        self.optimizer = self._tok('optimizer/optimizer')
        self.eval_op = self._tok('total_loss/loss:0 (deferred)')      # forward + loss without the update; read with get_losses_step
        self.losses = {k: self._tok(n) for k, n in zip(
            LOSS_NAMES, ('total_loss/loss:0', 'localization_loss/localization_loss:0',
                         'confidence_loss/confidence_loss:0', 'total_loss/l2_loss:0'))}

    def build_optimizer_from_metagraph(self):
        """ssdvgg.py:133-150: optimizer state (momentum, global_step, schedule) from the checkpoint."""
        ck = getattr(self, '_ckpt', None)
        if ck is None:
            raise RuntimeError('build_from_metagraph first')
        lr = LearningRate(list(ck['__lr_values__']), list(ck['__lr_boundaries__']))
        self.build_optimizer(lr, float(ck['__weight_decay__']), float(ck['__momentum__']), int(ck['__global_step__']))
        for k in ck.files:
            if k.startswith('__momentum__/'):
                a = np.ascontiguousarray(ck[k], np.float32)
                check(lib.ssd_load_momentum(self._h, k[len('__momentum__/'):].encode(), np_ptr(a), a.size))

    def build_summaries(self, restore):
        """ssdvgg.py:625-649 builds TensorBoard histograms; observability is out of scope here."""
        return None

    # ------------------------------------------------------------------ variables
    def variables(self):
        """[(tf_name, shape)] in arena order."""
        out = []
        buf = C.create_string_buffer(128)
        nd = C.c_int(); shp = (C.c_int * 4)()
        for i in range(lib.ssd_num_variables(self._h)):
            check(lib.ssd_variable_info(self._h, i, buf, 128, C.byref(nd), shp))
            out.append((buf.value.decode(), tuple(shp[k] for k in range(nd.value))))
        return out

    def load_variables(self, weights):
        known = dict(self.variables())
        for name, arr in weights.items():
            if name not in known:
                raise RuntimeError('no such variable: ' + name)
            a = np.ascontiguousarray(arr, np.float32)
            if tuple(a.shape) != known[name]:
                raise ValueError(f'{name}: shape {a.shape}, expected {known[name]}')
            check(lib.ssd_load_variable(self._h, name.encode(), np_ptr(a), a.size))

    def _save(self, fn, names=None):
        out = {}
        for name, shape in self.variables():
            if names is not None and name not in names:
                continue
            a = np.empty(shape, np.float32)
            check(fn(self._h, name.encode(), np_ptr(a), a.size))
            out[name] = a
        return out

    def save_variables(self, names=None):
        return self._save(lib.ssd_save_variable, names)

    def save_gradients(self, names=None):
        return self._save(lib.ssd_save_gradient, names)

    def save_momentum(self, names=None):
        return self._save(lib.ssd_save_momentum, names)

    @property
    def global_step(self):
        s = C.c_longlong()
        check(lib.ssd_get_global_step(self._h, C.byref(s)))
        return s.value

    def save_checkpoint(self, path, lr=None, momentum=0.9, weight_decay=0.0005):
        """tf.train.Saver.save counterpart (train.py:336-343): one .npz, reference variable names."""
        d = self.save_variables()
        if self.training:
            d.update({'__momentum__/' + k: v for k, v in self.save_momentum().items()})
        lr = lr or LearningRate([0.001], [])
        d.update(__preset__=np.array(self.preset.name), __num_classes__=np.array(self._n_classes),
                 __global_step__=np.array(self.global_step), __lr_values__=np.array(lr.values, np.float64),
                 __lr_boundaries__=np.array(lr.boundaries, np.int64), __momentum__=np.array(momentum),
                 __weight_decay__=np.array(weight_decay))
        np.savez(path, **d)

    # ------------------------------------------------------------------ steps
    def _check_x(self, x):
        x = np.ascontiguousarray(x, np.float32)
        H, W = self.preset.image_size.h, self.preset.image_size.w
        if x.ndim != 4 or x.shape[1:] != (H, W, 3):
            raise ValueError(f'image_input must be [b, {H}, {W}, 3] float32, got {x.shape}')   # data_queue.py:63-79
        if x.shape[0] < 1 or x.shape[0] > self.max_batch:
            raise ValueError(f'batch {x.shape[0]} outside 1..{self.max_batch} (max_batch)')
        return x

    def _check_y(self, y, b):
        y = np.ascontiguousarray(y, np.float32)
        if y.shape != (b, self.preset.num_anchors, self.num_vars):
            raise ValueError(f'labels must be [{b}, {self.preset.num_anchors}, {self.num_vars}] float32, got {y.shape}')
        return y

    # ---- device-resident feeds: a torch CUDA tensor in the feed goes through the *_dev entry points --------------
    @staticmethod
    def _is_cuda(t):
        return hasattr(t, 'is_cuda') and t.is_cuda

    def _dev_xy(self, x, y):
        import torch
        H, W = self.preset.image_size.h, self.preset.image_size.w
        if x.dtype != torch.float32 or x.dim() != 4 or tuple(x.shape[1:]) != (H, W, 3):
            raise ValueError(f'image_input must be [b, {H}, {W}, 3] float32, got {tuple(x.shape)} {x.dtype}')
        if x.shape[0] < 1 or x.shape[0] > self.max_batch:
            raise ValueError(f'batch {x.shape[0]} outside 1..{self.max_batch} (max_batch)')
        x = x.contiguous()
        if y is not None:
            if not self._is_cuda(y):
                y = torch.from_numpy(self._check_y(y, x.shape[0])).to(x.device)
            if y.dtype != torch.float32 or tuple(y.shape) != (x.shape[0], self.preset.num_anchors, self.num_vars):
                raise ValueError(f'labels must be [{x.shape[0]}, {self.preset.num_anchors}, {self.num_vars}] float32, got {tuple(y.shape)}')
            y = y.contiguous()
        return x, y

    def _dev_result(self, b, want_result):
        if not want_result:
            return None
        res = np.empty((b, self.preset.num_anchors, self.num_vars), np.float32)
        check(lib.ssd_get_result(self._h, b, np_ptr(res)))
        return res

    def train_step(self, x, y, want_result=True, want_losses=True):
        if self._is_cuda(x):
            x, y = self._dev_xy(x, y)
            self.train_step_dev(x, y)
            # neither fetch: nothing waits for the step (the driver reads the losses one step late, get_losses_step)
            return self._dev_result(x.shape[0], want_result), (self.get_losses() if want_losses else None)
        x = self._check_x(x); y = self._check_y(y, x.shape[0])
        res = np.empty(y.shape, np.float32) if want_result else None
        L = np.zeros(4, np.float32)
        check(lib.ssd_train_step(self._h, np_ptr(x), np_ptr(y), x.shape[0], np_ptr(res), np_ptr(L)))
        return res, dict(zip(LOSS_NAMES, (float(v) for v in L)))

    def eval_step(self, x, y, want_result=True, want_losses=True):
        if self._is_cuda(x):
            x, y = self._dev_xy(x, y)
            self.eval_step_dev(x, y)
            return self._dev_result(x.shape[0], want_result), (self.get_losses() if want_losses else None)
        x = self._check_x(x); y = self._check_y(y, x.shape[0])
        res = np.empty(y.shape, np.float32) if want_result else None
        L = np.zeros(4, np.float32)
        check(lib.ssd_eval_step(self._h, np_ptr(x), np_ptr(y), x.shape[0], np_ptr(res), np_ptr(L)))
        return res, dict(zip(LOSS_NAMES, (float(v) for v in L)))

    def infer(self, x):
        if self._is_cuda(x):
            x, _ = self._dev_xy(x, None)
            self.infer_dev(x)
            return self._dev_result(x.shape[0], True)
        x = self._check_x(x)
        res = np.empty((x.shape[0], self.preset.num_anchors, self.num_vars), np.float32)
        check(lib.ssd_infer(self._h, np_ptr(x), x.shape[0], np_ptr(res)))
        return res

    def activation(self, name, b):
        H = C.c_int(); W = C.c_int(); Ch = C.c_int()
        check(lib.ssd_activation_shape(self._h, name.encode(), C.byref(H), C.byref(W), C.byref(Ch)))
        a = np.empty((b, H.value, W.value, Ch.value), np.float32)
        check(lib.ssd_activation(self._h, name.encode(), b, np_ptr(a), a.size))
        return a

    def pool_fusion(self):
        """Per 2x2 stride-2 pool of the graph (pool1 ... in graph order): (fused into its producer's forward epilogue, fused into
        its consumer's data gradient).  A fused pool's input / its output gradient are not materialised (activation() refuses)."""
        out = (C.c_int * 8)()
        n = C.c_int()
        check(lib.ssd_pool_fusion(self._h, out, 8, C.byref(n)))
        return [(bool(out[i] & 1), bool(out[i] & 2)) for i in range(n.value)]

    # device-resident steps (torch tensors on this net's GPU)
    def forward_backward_dev(self, x_t, y_t):
        check(lib.ssd_forward_backward_dev(self._h, x_t.data_ptr(), y_t.data_ptr(), x_t.shape[0]))

    def forward_dev(self, x_t, y_t):
        check(lib.ssd_forward_dev(self._h, x_t.data_ptr(), y_t.data_ptr(), x_t.shape[0]))

    def use_torch_wgrad_stream(self):
        """Run the weight gradients on a torch-owned side stream (returned) so collectives can be enqueued
        behind exactly that stream while the data gradients keep running on the current stream."""
        import torch
        if getattr(self, 'wgrad_stream', None) is None:
            self.wgrad_stream = torch.cuda.Stream(device=self.device)
            check(lib.ssd_set_wgrad_stream(self._h, self.wgrad_stream.cuda_stream))
        return self.wgrad_stream

    def backward_staged(self, y_t, b, min_floats, sync_main=True):
        """Generator over (offset, count) ranges of the gradient arena as backward finishes them."""
        check(lib.ssd_backward_begin_dev(self._h, y_t.data_ptr(), b))
        off = C.c_size_t(); cnt = C.c_size_t(); more = C.c_int(1)
        while more.value:
            check(lib.ssd_backward_next_dev(self._h, int(min_floats), int(sync_main), C.byref(off), C.byref(cnt), C.byref(more)))
            if cnt.value:
                yield off.value, cnt.value

    def backward_ranges(self, min_floats):
        """[(offset, count)] exactly as backward_staged(..., min_floats) yields them, without running backward."""
        cap = 128
        while True:
            offs = (C.c_size_t * cap)(); cnts = (C.c_size_t * cap)(); n = C.c_int()
            check(lib.ssd_backward_ranges(self._h, int(min_floats), offs, cnts, cap, C.byref(n)))
            if n.value <= cap:      # (a truncated list would desynchronise the ranks' collectives: ask again with room for all)
                return [(offs[i], cnts[i]) for i in range(n.value)]
            cap = n.value

    def set_loss_normalizer(self, batch):
        """reduce_mean over `batch` samples instead of the step's own b (<= 0 restores the default): data
        parallel shards of unequal size pass global_samples / world (parallel.train_step_dp)."""
        check(lib.ssd_set_loss_normalizer(self._h, float(batch)))

    def null_gradients_dev(self):
        """The gradient arena of a step without samples: weight_decay * filters, zero elsewhere."""
        check(lib.ssd_null_gradients_dev(self._h))

    def apply_gradients_dev(self, grad_scale=1.0):
        check(lib.ssd_apply_gradients_dev(self._h, float(grad_scale)))

    def train_step_dev(self, x_t, y_t):
        check(lib.ssd_train_step_dev(self._h, x_t.data_ptr(), y_t.data_ptr(), x_t.shape[0]))

    def eval_step_dev(self, x_t, y_t):
        check(lib.ssd_eval_step_dev(self._h, x_t.data_ptr(), y_t.data_ptr(), x_t.shape[0]))

    def infer_dev(self, x_t):
        check(lib.ssd_infer_dev(self._h, x_t.data_ptr(), x_t.shape[0]))

    def get_losses(self):
        L = np.zeros(4, np.float32)
        check(lib.ssd_get_losses(self._h, np_ptr(L)))
        return dict(zip(LOSS_NAMES, (float(v) for v in L)))

    def get_losses_step(self, steps_back=0):
        """The losses of the last step (0) or of the step 1 / 2 before it, waiting only for THAT step's forward pass
        (ssd_get_losses_step): a driver books step k - 1 after launching step k instead of stalling on every step."""
        L = np.zeros(4, np.float32)
        check(lib.ssd_get_losses_step(self._h, int(steps_back), np_ptr(L)))
        return dict(zip(LOSS_NAMES, (float(v) for v in L)))

    def set_stream(self, stream_ptr):
        check(lib.ssd_set_stream(self._h, stream_ptr))

    def _det_caps(self, cap, mo):
        cap = -1 if cap is None else int(cap)
        mo = -1 if mo is None else int(mo)
        out_cap = self.preset.num_anchors if cap < 0 else max(cap, 1)
        if mo >= 0:
            out_cap = max(min(out_cap, mo), 1)
        return cap, mo, out_cap

    def detect_last_launch(self, b, confidence_threshold=0.5, detections_cap=200, max_out=None, nms=True):
        """Enqueue decode + NMS of the last step's result and the copy of its (small) output; returns a
        ticket whose get() yields what detect_last returns.  Two output slots alternate in the handle, so a
        caller may launch the next batch before it collects this one (infer.py:225-235, pipelined)."""
        cap, mo, out_cap = self._det_caps(detections_cap, max_out)
        check(lib.ssd_detect_last_dev(self._h, b, float(confidence_threshold), cap, mo, out_cap, 1 if nms else 0,
                                      None, None, None, None, None))
        self._det_serial = getattr(self, '_det_serial', 0) + 1
        if not hasattr(self, '_det_views'):
            self._det_views = {}
        return _Detections(self, self._det_serial, b, out_cap)

    def detect_last(self, b, confidence_threshold=0.5, detections_cap=200, max_out=None, nms=True):
        """decode + NMS of the last step's result without leaving the GPU (train.py:275-277)."""
        # (copies: a caller of this synchronous form may keep the result across any number of later passes)
        return [{k: v.copy() for k, v in d.items()} for d in self.detect_last_launch(b, confidence_threshold, detections_cap, max_out, nms).get()]

    # ------------------------------------------------------------------ Session.run routing
    def _run(self, fetches, feed):
        x = feed.get(self.image_input)
        if x is None:
            raise ValueError('feed_dict must hold net.image_input')
        y = feed.get(self.labels) if self.labels is not None else None
        want_opt = any(f is self.optimizer for f in fetches if self.optimizer is not None)
        want_loss = any(isinstance(f, dict) or (self.losses and f in self.losses.values()) for f in fetches)
        want_res = any(f is self.result for f in fetches)      # the [b, A, C+5] copy to the host only when fetched
        want_eval = any(f is self.eval_op for f in fetches if getattr(self, 'eval_op', None) is not None)
        if want_opt:
            if y is None:
                raise ValueError('feed_dict must hold net.labels')
            res, L = self.train_step(x, y, want_res, want_loss)
        elif want_loss or want_eval:
            if y is None:
                raise ValueError('feed_dict must hold net.labels')
            res, L = self.eval_step(x, y, want_res, want_loss)
        else:
            res, L = self.infer(x), None
        out = []
        for f in fetches:
            if f is self.result:
                out.append(res)
            elif isinstance(f, dict):
                out.append({k: L[k] for k in f})
            elif self.optimizer is not None and f is self.optimizer:
                out.append(None)
            elif getattr(self, 'eval_op', None) is not None and f is self.eval_op:
                out.append(None)
            elif self.losses and f in self.losses.values():
                out.append(L[[k for k, v in self.losses.items() if v is f][0]])
            else:
                raise ValueError(f'cannot fetch {f!r}')
        return out

    # ------------------------------------------------------------------ names (ssdvgg.py:602-622)
    def __build_names(self):
        self.original_scopes = [
            'conv1_1', 'conv1_2', 'conv2_1', 'conv2_2', 'conv3_1', 'conv3_2', 'conv3_3', 'conv4_1', 'conv4_2',
            'conv4_3', 'conv5_1', 'conv5_2', 'conv5_3', 'mod_conv6', 'mod_conv7']
        self.new_scopes = ['conv8_1', 'conv8_2', 'conv9_1', 'conv9_2', 'conv10_1', 'conv10_2', 'conv11_1', 'conv11_2']
        if len(self.preset.maps) == 7:
            self.new_scopes += ['conv12_1', 'conv12_2']
        for i in range(len(self.preset.maps)):
            for j in range(2 + len(self.preset.maps[i].aspect_ratios)):
                self.new_scopes.append('classifiers/classifier{}_{}'.format(i, j))
