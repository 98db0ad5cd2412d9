"""Batch feeder for the step loop: the counterpart of the reference's TrainingData /
DataQueue (training_data.py:35-206, data_queue.py:26-112) for the path this build covers.

The reference forks N workers that run cv2 transforms and copy finished float32 batches through
shared-memory slots to ONE device.  Here every rank feeds only its own shard (parallel.ShardSampler) and
a batch is born in HBM: the images come out of the batch augmentation kernels (transforms.augment_batch:
the reference's transform recipe, decisions on the host, pixels on the GPU), the label vectors out of the
HIP label encoder (ssd_encode_labels_dev) -- only the images' source bytes and a handful of boxes go up,
nothing of a batch ever comes down.  A generator yields `(images, labels, gt_boxes)` like the
reference's, with `images` / `labels` float32 CUDA tensors (device_tensors=False: numpy arrays).

`gen_batch(batch_size, num_workers)` means what the reference's means (training_data.py:137-195):

  * num_workers == 0: the serial generator -- batch k+1 is prepared when the caller asks for it;
  * num_workers  > 0: that many forked worker processes run the host half of the recipe (file decode, the
    transforms' decisions incl. the <= 50 redraws, packing) into the slots of a data_queue.DataQueue,
    while a feeder thread of the training process uploads finished slots IN ORDER and launches the
    augmentation + label kernels on its own stream into a ring of three device slots.  Batch k+1 is
    therefore ready in HBM while step k runs; the consumer's stream waits for the slot's event, and a
    slot is rewritten only after the kernels the consumer enqueued on it have run.  A yielded batch is
    valid until the next one is requested.

The two are bit-identical: every sample's decisions are drawn from Python's `random` seeded with
(seed, data set, epoch, sample index), so a batch does not depend on which process prepared it or on what
was drawn before (tests/test_feeder_cpu.py, tests/test_gpu_feeder.py).  The reference's workers all start
from the parent's `random` state and its batches arrive in completion order: an epoch there is not
reproducible; that is not mirrored.

Sources:
  * data_dir None / '' / 'synthetic'  -- SURVEY.md 8d: float32 preset-sized images (augment=False) or a
    uint8 "dataset" of variously sized images that goes through the whole recipe (augment=True);
  * any other data_dir                 -- a dataset directory read through a data source module
    (`source_<data_source>.get_source()`, utils.py:44-55 / process_dataset.py:199-204), e.g. the Pascal
    VOC tree.  The reference splits this in two programs: process_dataset.py pickles the sample lists and
    the transform objects, train.py unpickles them (training_data.py:41-69).  Those pickles hold instances
    of the reference's own classes; here the same sample lists and the same recipes
    (transforms.build_train_transforms / build_valid_transforms) are built in process.  Images are
    decoded by transforms.load_image_bgr (.npy arrays or Pillow; decoder parity with OpenCV unpinned).
"""
import multiprocessing as mp
import os
import queue
import random
import threading
import time
import traceback

import numpy as np

from .data_queue import DataQueue, WorkerError
from .parallel import ShardSampler
from .ssdutils import (encode_labels_batch, encode_labels_batch_dev, get_preset_by_name, has_positive_anchor,
                       prime_anchor_table)
from .utils import Box, Point, Size, Sample, load_data_source

VOC_NAMES = ['aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable', 'dog',
             'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor']

DEVICE_SLOTS = 3            # one under the step, one ready, one being filled
CACHE_SYNTHETIC_UP_TO = 4096  # synthetic "files" kept in RAM (a real source reads its files instead)


def _gt_arrays(gts):
    bxs = [np.array([[b.center.x, b.center.y, b.size.w, b.size.h] for b in g], np.float64).reshape(-1, 4) for g in gts]
    cls = [np.array([b.labelid for b in g], np.int32) for g in gts]
    return bxs, cls


def _gt_packed(gts, num_classes):
    """The batch's boxes as the label encoder's three arrays (what ssd_encode_labels* take): [ntot, 4] float64
    proportional (cx, cy, w, h), [ntot] int32 class ids, [b + 1] int32 offsets.  They ride in the batch's slot and
    reach the GPU with the images' bytes; the class range is checked here (the resident encoder does not)."""
    from .ssdutils import _pack_gt
    bxs, cls = _gt_arrays(gts)
    gt, gcls, goff = _pack_gt(bxs, cls)
    if gcls.size and (gcls.min() < 0 or gcls.max() >= num_classes):
        raise ValueError('a ground-truth class id lies outside 0..%d' % (num_classes - 1))
    return {'gt': gt, 'gcls': gcls, 'goff': goff}


class _Recipe:
    """One data set of a TrainingData (train or valid): where its samples come from and how a list of sample
    indices becomes the host half of a batch.  Lives in the parent and, by fork, in the workers."""

    def __init__(self, td, which, total, salt, sample_at=None, transforms=None):
        self.td, self.which, self.total, self.salt = td, which, total, salt
        self.sample_at, self.transforms = sample_at, transforms
        self.pool = None
        self.active = False
        self.dev = None                        # this data set's device-side slot ring + feeder stream (TrainingData._device_ring)

    def __getstate__(self):      # what a worker needs: not the pool it belongs to, nothing that lives on the GPU
        st = dict(self.__dict__)
        st['pool'] = None
        st['dev'] = None
        st['active'] = False
        return st

    # ---- host half of a batch: runs in a worker process or, for num_workers == 0, in the caller --------------------
    def plan(self, epoch, idx):
        """(The transforms draw from the module-level `random`, seeded per sample below.  In the serial generator this runs in the
        caller's process: its generator state is put back afterwards -- the reference never reseeds, and whatever else draws from
        `random` in the driver must not become a function of the last sample index.)"""
        state = random.getstate()
        try:
            return self._plan(epoch, idx)
        finally:
            random.setstate(state)

    def _plan(self, epoch, idx):
        td = self.td
        if self.transforms is None:                              # preset-sized float32 synthetic images
            imgs, gts = [], []
            for i in idx:
                # redrawn (<= 50 times) until at least one anchor is positive (training_data.py:92-98)
                for tries in range(50):
                    img, gt = td._sample(int(i) + 7919 * tries, self.salt)
                    if has_positive_anchor(td.preset, gt):
                        break
                imgs.append(img); gts.append(gt)
            return dict({'images': np.stack(imgs)}, **_gt_packed(gts, td.num_classes)), gts
        from . import transforms as T
        host_tfs = [t for t in self.transforms if not isinstance(t, T.LabelCreatorTransform)]
        loader = host_tfs[0]
        preset_images = getattr(loader, 'images', None)
        plans, gts = [], []
        try:
            for i in idx:
                s = self.sample_at(int(i))
                if not isinstance(s, Sample):                    # synthetic: the pixels travel with the record
                    loader.images, sample = s
                else:
                    loader.images, sample = preset_images, s
                random.seed(td._sample_seed(self.salt, epoch, int(i)))
                # run_transforms until at least one anchor is positive, at most 50 times (training_data.py:88-98);
                # the label of a try is only LOOKED at there (num_bg < rows), so the test runs on the host and the
                # label vectors of the whole batch are encoded once, on the GPU
                for _ in range(50):
                    args = (None, None, sample)
                    for t in host_tfs:
                        args = t(*args)
                    if has_positive_anchor(td.preset, args[2].boxes):
                        break
                plans.append(args[0]); gts.append(args[2].boxes)
        finally:
            loader.images = preset_images
        arr, packed = T.plan_params(plans, td.preset.image_size.w, td.preset.image_size.h)
        return dict({'params': np.frombuffer(arr, np.uint8).copy(), 'packed': packed}, **_gt_packed(gts, td.num_classes)), gts

    def slot_bytes(self, batch_size):
        td = self.td
        W, H = td.preset.image_size.w, td.preset.image_size.h
        if self.transforms is None:
            return batch_size * (H * W * 3 * 4 + 4096) + 8192
        return batch_size * (td._max_image_bytes + 16 + 256 + 4096) + 8192


def _worker_main(recipe, tasks, results, anchors_abs=None):
    """batch_producer of the reference (training_data.py:109-134).  The workers never touch the GPU: the anchor table of
    the redraw test is computed by the training process before they start and handed over (`anchors_abs`; a forked worker
    has inherited it as well)."""
    if anchors_abs is not None:
        prime_anchor_table(recipe.td.preset, anchors_abs)
    try:
        import signal
        signal.signal(signal.SIGINT, signal.SIG_IGN)
    except Exception:
        pass
    parent = os.getppid()
    while True:
        try:
            task = tasks.get(timeout=5.0)
        except queue.Empty:
            if os.getppid() != parent:      # the training process is gone (killed): do not linger
                break
            continue
        if task is None:
            break
        gen, seq, slot, epoch, idx = task
        try:
            arrays, gts = recipe.plan(epoch, idx)
            results.put((gen, seq), slot, arrays, gts)
        except BaseException:
            results.put_error((gen, seq), slot, traceback.format_exc())


class _SharedImageCache:
    """The synthetic data sets' "files" (uint8 images) in shared memory: written once by the training process while it builds
    its sample list, read as views by every worker however it was started (a forked worker used to inherit a dict of arrays;
    one that comes from the fork server inherits nothing).  A real source reads its files in the workers instead."""
    CHUNK = 128 << 20

    def __init__(self, limit):
        self.limit, self.chunks, self.used, self.index, self.records = limit, [], 0, {}, {}

    def get(self, key):
        e = self.index.get(key)
        if e is None:
            return None
        c, off, h, w = e
        name, sample = self.records[key]
        img = np.frombuffer(self.chunks[c], dtype=np.uint8, count=h * w * 3, offset=off).reshape(h, w, 3)
        return {name: img}, sample

    def put(self, key, name, img, sample):
        if len(self.index) >= self.limit:
            return
        n = int(img.nbytes)
        if not self.chunks or self.used + n > len(self.chunks[-1]):
            self.chunks.append(mp.get_context('fork').RawArray('B', max(self.CHUNK, n)))
            self.used = 0
        c, off = len(self.chunks) - 1, self.used
        np.frombuffer(self.chunks[c], dtype=np.uint8, count=n, offset=off)[:] = img.reshape(-1)
        self.used += (n + 255) // 256 * 256
        self.index[key] = (c, off, img.shape[0], img.shape[1])
        self.records[key] = (name, sample)

    def __len__(self):
        return len(self.index)


def _start_context():
    """How the planning workers come to life.  Default `forkserver`: the training process asks a small server process --
    started once with a fresh interpreter, this module preloaded, never a HIP call -- to fork the workers, so what a worker
    costs does not depend on the training process: forking THAT copies the page tables of everything it has ever mapped
    (6-8 ms in a fresh process with 7 GB of HBM mapped, SECONDS in one that has created and destroyed a hundred handles:
    profiles/r03_b_fork_probe.txt -- round 3 ordered its test suite around this).  The recipe, the shared-memory slots and the
    anchor table travel to the worker as pickled Process arguments.  SSD_FEEDER_START=fork restores the direct fork."""
    method = os.environ.get('SSD_FEEDER_START', 'forkserver')
    if method not in ('fork', 'forkserver', 'spawn'):
        raise ValueError('SSD_FEEDER_START must be fork, forkserver or spawn, got %r' % (method,))
    ctx = mp.get_context(method)
    if method == 'forkserver':
        try:
            ctx.set_forkserver_preload(['ssd_tensorflow_amd.training_data', 'ssd_tensorflow_amd.transforms'])
        except Exception:
            pass
    return ctx


class _NoMainReimport:
    """While the workers start: keep multiprocessing from re-importing the launching script in them.  A worker that comes from
    the fork server (or from spawn) is told the parent's `__main__` -- by path or by module name -- and runs it again as
    `__mp_main__` before its target; everything a worker needs (`_worker_main`, the recipe's classes) is importable from this
    package, while a training script WITHOUT an `if __name__ == '__main__':` guard would run a second time in every worker (and
    die at its own first worker start: BrokenPipeError in the parent).  Nothing defined in `__main__` can therefore travel to a
    worker: a data source or transform belongs in a module."""

    def __enter__(self):
        import sys
        self.main = sys.modules.get('__main__')
        self.saved = {}
        for k in ('__spec__', '__file__'):
            if self.main is not None and hasattr(self.main, k):
                self.saved[k] = getattr(self.main, k)
                try:
                    if k == '__spec__':
                        setattr(self.main, k, None)
                    else:
                        delattr(self.main, k)
                except Exception:
                    self.saved.pop(k)
        return self

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            setattr(self.main, k, v)
        return False


class _SampleAt:
    """sample #i of a data set (picklable: it travels to the workers)."""

    def __init__(self, td, which, salt):
        self.td, self.which, self.salt = td, which, salt

    def __call__(self, i):
        td = self.td
        samples = td.train_samples if self.which == 'train' else td.valid_samples
        if td._real_dataset:
            return samples[i]
        return td._dataset_sample(i, self.salt)


class _WorkerPool:
    def __init__(self, recipe, num_workers, slot_bytes):
        ctx = _start_context()
        self.num_workers, self.slot_bytes = num_workers, slot_bytes
        self.nslots = num_workers + 2
        self.tasks = ctx.Queue()
        self.results = DataQueue(slot_bytes, self.nslots, ctx)
        self.free_slots = list(range(self.nslots))
        self.outstanding = {}                 # (gen, seq) -> slot of tasks whose result has not been seen
        self.generation = 0
        self.pinned = []
        self.workers = []
        anchors_abs = prime_anchor_table(recipe.td.preset)      # (computed once per process, on the GPU unless a test installed one)
        try:
            with _NoMainReimport():
                for _ in range(num_workers):
                    w = ctx.Process(target=_worker_main, args=(recipe, self.tasks, self.results, anchors_abs), daemon=True)
                    w.start()
                    self.workers.append(w)
        except Exception as e:      # no fork server to be had (e.g. nothing picklable about a custom source): the direct fork still works
            if ctx.get_start_method() == 'fork':
                raise
            import warnings
            warnings.warn('feeder workers could not be started through %s (%s: %s); forking them directly' %
                          (ctx.get_start_method(), type(e).__name__, e), RuntimeWarning)
            for w in self.workers:
                try:
                    w.terminate()
                except Exception:
                    pass
            self.workers = []
            fork = mp.get_context('fork')
            for _ in range(num_workers):
                w = fork.Process(target=_worker_main, args=(recipe, self.tasks, self.results, anchors_abs), daemon=True)
                w.start()
                self.workers.append(w)

    def pin(self, device):
        """Page-lock the slots so the uploads are asynchronous DMA transfers (a refusal is harmless: the copies are
        then staged by the runtime)."""
        import torch
        if self.pinned or not torch.cuda.is_available():
            return
        rt = torch.cuda.cudart()
        for a in self.results.array_pool:
            try:
                rc = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)
                self.pinned.append(int(rc) == 0)
            except Exception:
                self.pinned.append(False)

    def check_alive(self):
        for w in self.workers:
            if not w.is_alive():
                raise RuntimeError('a feeder worker process died (exit code %s)' % (w.exitcode,))

    def unpin(self):
        """Before the slots' memory goes away: a registration that outlives its pages poisons whatever the allocator
        maps at that address next (a later hipMemcpy from an ordinary buffer there fails with "invalid argument")."""
        if not any(self.pinned):
            self.pinned = []
            return
        import torch
        rt = torch.cuda.cudart()
        for a, ok in zip(self.results.array_pool, self.pinned):
            if ok:
                try:
                    rt.cudaHostUnregister(a.ctypes.data)
                except Exception:
                    pass
        self.pinned = []

    def close(self):
        self.unpin()
        for _ in self.workers:
            try:
                self.tasks.put(None)
            except Exception:
                pass
        for w in self.workers:
            w.join(timeout=2)
            if w.is_alive():
                w.terminate()
        self.workers = []


class TrainingData:
    """Same attributes the drivers read from the reference's TrainingData: preset, num_classes,
    lid2name, lname2id, num_train, num_valid, train_samples, valid_samples, train_generator,
    valid_generator.  `global_count` holds the size of the global batch the generator yielded last
    (what parallel.train_step_dp needs when the last batch of an epoch is short)."""

    def __init__(self, data_dir=None, preset='vgg300', num_train=64, num_valid=16, seed=1234, rank=0, world=1,
                 augment=False, sampler_trials=50, expand_prob=0.5, device=0, device_tensors=True,
                 data_source='pascal_voc', valid_fraction=0.025, images=None):
        self.preset = get_preset_by_name(preset) if isinstance(preset, str) else preset
        self.seed, self.rank, self.world = seed, rank, world
        self.epoch = 0
        self.device, self.device_tensors = device, bool(device_tensors)
        self.global_count = 0
        self._upload_hook = None               # tests: replaces the GPU half of a batch
        self._synthetic_cache = _SharedImageCache(CACHE_SYNTHETIC_UP_TO)
        # where a prefetched epoch's time went (seconds, reset by every gen_batch call with workers): the consumer waiting
        # for a batch, the feeder thread waiting for the workers / for a free device slot / uploading
        self.feeder_stats = dict(consumer_wait=0.0, worker_wait=0.0, slot_wait=0.0, upload=0.0, batches=0)
        from . import transforms as T
        # 'shapes': the learnable synthetic set (textured rectangles, class = texture; _shapes_canvas) -- same plumbing as
        # 'synthetic', whose uniform-noise images carry nothing a detector could learn
        self._shapes = data_dir == 'shapes'
        self._real_dataset = data_dir not in (None, '', 'synthetic', 'shapes')
        if data_dir not in (None, '', 'synthetic', 'shapes'):
            # ---- a real dataset directory (training_data.py:41-69 + process_dataset.py:199-252) ----
            try:
                source = load_data_source(data_source)
                source.load_trainval_data(data_dir, valid_fraction)
            except (ImportError, AttributeError, OSError) as e:
                raise RuntimeError(str(e))                                              # training_data.py:48-49
            self.num_classes = source.num_classes
            self.label_colors = source.colors
            self.lid2name, self.lname2id = source.lid2name, source.lname2id
            self.train_samples, self.valid_samples = list(source.train_samples), list(source.valid_samples)
            self.num_train, self.num_valid = len(self.train_samples), len(self.valid_samples)
            self.augment = True
            self.train_transforms = T.build_train_transforms(self.preset, self.num_classes, sampler_trials, expand_prob, images)
            self.valid_transforms = T.build_valid_transforms(self.preset, self.num_classes, images)
            self._max_image_bytes = max([s.imgsize.w * s.imgsize.h * 3 for s in self.train_samples + self.valid_samples] + [1])
            self._recipes = {
                'train': _Recipe(self, 'train', self.num_train, 0, _SampleAt(self, 'train', 0), self.train_transforms),
                'valid': _Recipe(self, 'valid', self.num_valid, 1 << 20, _SampleAt(self, 'valid', 1 << 20), self.valid_transforms)}
        else:
            self.num_classes = 20
            self.label_colors = {}
            self.lid2name = dict(enumerate(VOC_NAMES))
            self.lname2id = {n: i for i, n in self.lid2name.items()}
            self.num_train, self.num_valid = num_train, num_valid
            self.augment = bool(augment)
            if self.augment:
                self.train_transforms = T.build_train_transforms(self.preset, self.num_classes, sampler_trials, expand_prob)
                self.valid_transforms = T.build_valid_transforms(self.preset, self.num_classes)
                self.train_samples = [self._dataset_sample(i, 0)[1] for i in range(num_train)]
                self.valid_samples = [self._dataset_sample(i, 1 << 20)[1] for i in range(num_valid)]
                self._max_image_bytes = 640 * 640 * 3
                self._recipes = {
                    'train': _Recipe(self, 'train', num_train, 0, _SampleAt(self, 'train', 0), self.train_transforms),
                    'valid': _Recipe(self, 'valid', num_valid, 1 << 20, _SampleAt(self, 'valid', 1 << 20), self.valid_transforms)}
            else:
                self.train_samples = self.valid_samples = None
                self._recipes = {'train': _Recipe(self, 'train', num_train, 0), 'valid': _Recipe(self, 'valid', num_valid, 1 << 20)}
        self.train_generator = self._make_generator(self._recipes['train'])
        self.valid_generator = self._make_generator(self._recipes['valid'])

    # ---- what travels to a worker process (forkserver / spawn start: _start_context) ------------------------------
    def __getstate__(self):
        st = dict(self.__dict__)
        for k in ('_upload_hook', 'train_generator', 'valid_generator'):
            st[k] = None
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)

    # ---- lifetime of the worker processes ----------------------------------------------------------------------
    def close(self):
        for r in self._recipes.values():
            if r.pool is not None:
                r.pool.close()
                r.pool = None
            r.dev = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _sample_seed(self, salt, epoch, index):
        """The seed of sample #index's decisions in `epoch` of data set `salt`: one integer whose 32-bit words all
        enter random.seed's init_by_array."""
        return ((int(self.seed) & 0xffffffff) << 96) | (int(salt) << 64) | (int(epoch) << 32) | int(index)

    # ---- labels: encoded once per batch on the GPU, left there ---------------------------------------------------
    def _labels(self, gts, out=None):
        bxs, cls = _gt_arrays(gts)
        if self.device_tensors:
            return encode_labels_batch_dev(self.preset, self.num_classes, bxs, cls, self.device, out=out)
        return encode_labels_batch(self.preset, self.num_classes, bxs, cls)

    # ---- the learnable synthetic set ('shapes') --------------------------------------------------------------------
    SHAPE_CLASSES = 4      # textures, booked under the first four of the 20 class ids (the net keeps the benchmark's shape)

    def _shapes_canvas(self, rng, W, H):
        """uint8 BGR image [H, W, 3] + boxes: 1..3 non-overlapping rectangles of 20..50 % of the frame on low-contrast
        noise, each filled with a two-colour texture that IS its class -- 0 horizontal stripes, 1 vertical stripes,
        2 checkerboard, 3 diagonal stripes (either direction: a horizontal flip keeps every class).  Texture, not colour,
        because the reference's training recipe permutes channels and shifts hue / saturation / contrast
        (process_dataset.py:66-140, transforms.py:117-240): a colour code would not survive it."""
        img = rng.integers(104, 152, (H, W, 3), dtype=np.uint8)
        n = int(rng.integers(1, 4))
        placed = []
        for _ in range(n):
            for _try in range(20):
                w, h = rng.uniform(0.2, 0.5, 2)
                cx, cy = rng.uniform(w / 2, 1 - w / 2), rng.uniform(h / 2, 1 - h / 2)
                if all(abs(cx - q[0]) >= (w + q[2]) / 2 or abs(cy - q[1]) >= (h + q[3]) / 2 for q in placed):
                    placed.append((float(cx), float(cy), float(w), float(h)))
                    break
        boxes = []
        for cx, cy, w, h in placed:
            c = int(rng.integers(0, self.SHAPE_CLASSES))
            x0, x1 = int(round((cx - w / 2) * W)), int(round((cx + w / 2) * W))
            y0, y1 = int(round((cy - h / 2) * H)), int(round((cy + h / 2) * H))
            p = int(rng.integers(5, 12)) * max(1, min(W, H) // 300)          # stripe width in pixels
            yy, xx = np.mgrid[y0:y1, x0:x1]
            sgn = 1 if rng.integers(0, 2) else -1
            pat = [(yy // p) % 2, (xx // p) % 2, (yy // p + xx // p) % 2, ((yy + sgn * xx) // p) % 2][c].astype(bool)
            lo, hi = rng.integers(0, 64, 3), rng.integers(192, 256, 3)
            img[y0:y1, x0:x1] = np.where(pat[..., None], hi, lo).astype(np.uint8)
            boxes.append(Box(self.lid2name[c], c, Point(cx, cy), Size(w, h)))
        return img, boxes

    # ---- a dataset of variously sized uint8 images through the reference's transform recipe ---------------------
    def _dataset_sample(self, index, salt):
        """Deterministic synthetic "file" #index: ({name: uint8 BGR image}, Sample record with 1..5 boxes)."""
        key = (salt, index)
        hit = self._synthetic_cache.get(key)
        if hit is not None:
            return hit
        rng = np.random.default_rng([self.seed, salt, index, 77])
        W, H = int(rng.integers(200, 640)), int(rng.integers(200, 640))
        if self._shapes:
            img, boxes = self._shapes_canvas(rng, W, H)
            name = 'shapes/%d/%d' % (salt, index)
            out = ({name: img}, Sample(name, boxes, Size(W, H)))
            self._synthetic_cache.put(key, name, img, out[1])
            return out
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        n = int(rng.integers(1, 6))
        w = rng.uniform(0.1, 0.6, n); h = rng.uniform(0.1, 0.6, n)
        cx = rng.uniform(w / 2, 1 - w / 2); cy = rng.uniform(h / 2, 1 - h / 2)
        cls = rng.integers(0, self.num_classes, n)
        boxes = [Box(self.lid2name[int(c)], int(c), Point(float(x), float(y)), Size(float(ww), float(hh)))
                 for x, y, ww, hh, c in zip(cx, cy, w, h, cls)]
        name = 'synthetic/%d/%d' % (salt, index)
        out = ({name: img}, Sample(name, boxes, Size(W, H)))
        self._synthetic_cache.put(key, name, img, out[1])
        return out

    # ---- preset-sized float32 synthetic images (no transform recipe) ------------------------------------------------
    def _sample(self, index, salt):
        """Deterministic synthetic sample #index: image + 1..5 GT boxes."""
        rng = np.random.default_rng([self.seed, salt, index])
        H, W = self.preset.image_size.h, self.preset.image_size.w
        if self._shapes:
            img, gt = self._shapes_canvas(rng, W, H)
            return img.astype(np.float32), gt
        img = rng.integers(0, 256, (H, W, 3)).astype(np.float32)        # BGR 0..255, training_data.py:100
        n = int(rng.integers(1, 6))
        w = rng.uniform(0.1, 0.6, n); h = rng.uniform(0.1, 0.6, n)
        boxes = np.stack([rng.uniform(w / 2, 1 - w / 2), rng.uniform(h / 2, 1 - h / 2), w, h], 1)
        cls = rng.integers(0, self.num_classes, n)
        gt = [Box(self.lid2name[int(ci)], int(ci), Point(*map(float, bi[:2])), Size(*map(float, bi[2:]))) for bi, ci in zip(boxes, cls)]
        return img, gt

    # ---- GPU half of a batch ------------------------------------------------------------------------------------------
    def _device_ring(self, recipe, batch_size):
        """Three device slots (images, labels, staging for the source bytes, tap-table workspace) + the feeder stream, PER DATA
        SET: a training and a validation generator may be live at the same time (the reference's queues are independent,
        training_data.py:147-195) and must not fill each other's slots.  Never replaced under a live generator (_prefetched
        refuses a second generator of the same data set)."""
        import torch
        from ._lib import lib
        d = recipe.dev
        if d is not None and d['batch'] >= batch_size:
            return d
        dev = torch.device('cuda', self.device)
        W, H = self.preset.image_size.w, self.preset.image_size.h
        A, nv = self.preset.num_anchors, self.num_classes + 5
        d = {'batch': batch_size, 'stream': torch.cuda.Stream(device=dev),
             'images': [torch.empty((batch_size, H, W, 3), dtype=torch.float32, device=dev) for _ in range(DEVICE_SLOTS)],
             'labels': [torch.empty((batch_size, A, nv), dtype=torch.float32, device=dev) for _ in range(DEVICE_SLOTS)],
             'packed': [None] * DEVICE_SLOTS, 'enc_ws': [None] * DEVICE_SLOTS,
             'ws': [torch.empty((lib.ssd_augment_ws_bytes(batch_size, W, H),), dtype=torch.uint8, device=dev) for _ in range(DEVICE_SLOTS)]}
        recipe.dev = d
        return d

    def _upload(self, arrays, gts, slot=None, ring=None):
        """Source bytes up, augmentation + label kernels on torch's current stream -> (images, labels) of this batch.
        slot None: fresh tensors (the serial generator); else the device slot to fill."""
        if self._upload_hook is not None:
            return self._upload_hook(arrays, gts, slot)
        import ctypes as C
        import torch
        from ._lib import lib, check
        b = len(gts)
        dev = torch.device('cuda', self.device)
        W, H = self.preset.image_size.w, self.preset.image_size.h
        ring = ring if slot is not None else None
        if 'images' in arrays:
            src = torch.from_numpy(arrays['images'])
            if ring is not None:
                images = ring['images'][slot][:b]
                images.copy_(src, non_blocking=True)
            else:
                images = src.to(dev, non_blocking=True) if self.device_tensors else arrays['images']
        else:
            packed, params = arrays['packed'], arrays['params']
            if ring is not None:
                if ring['packed'][slot] is None or ring['packed'][slot].numel() < packed.size:
                    ring['packed'][slot] = torch.empty((max(packed.size, ring['batch'] * (self._max_image_bytes + 16)),), dtype=torch.uint8, device=dev)
                staged = ring['packed'][slot][:packed.size]
                staged.copy_(torch.from_numpy(packed), non_blocking=True)
                images, ws = ring['images'][slot][:b], ring['ws'][slot]
            else:
                staged = torch.from_numpy(packed).to(dev)
                images = torch.empty((b, H, W, 3), dtype=torch.float32, device=dev)
                ws = torch.empty((lib.ssd_augment_ws_bytes(b, W, H),), dtype=torch.uint8, device=dev)
            check(lib.ssd_augment_batch_dev(staged.data_ptr(), C.c_void_p(params.ctypes.data), b, W, H, images.data_ptr(), ws.data_ptr(),
                                            torch.cuda.current_stream(dev).cuda_stream))
            if not self.device_tensors:
                images = images.cpu().numpy()
        labels = self._labels(gts, out=ring['labels'][slot][:b] if ring is not None else None)
        return images, labels

    def _upload_async(self, ring, slot_arr, arrays, gts, dslot):
        """The prefetching feeder's upload: ONE host-to-device transfer of the slot's used prefix (source bytes, parameter
        records, boxes), then the augmentation and label kernels, all enqueued on torch's current stream; nothing is waited
        for.  `arrays` are views into slot_arr (pinned when the registration succeeded): the slot must stay untouched until
        the returned work has run."""
        import ctypes as C
        import torch
        from ._lib import lib, check
        b = len(gts)
        dev = torch.device('cuda', self.device)
        W, H = self.preset.image_size.w, self.preset.image_size.h
        stream = torch.cuda.current_stream(dev).cuda_stream
        base = slot_arr.ctypes.data
        offs = {k: v.ctypes.data - base for k, v in arrays.items()}
        big = 'images' if 'images' in arrays else None
        small = [k for k in arrays if k != big]
        lo = min(offs[k] for k in small)
        hi = max(offs[k] + arrays[k].nbytes for k in small)
        need = hi - lo
        staged = ring['packed'][dslot]
        if staged is None or staged.numel() < need:
            cap = max(need, ring['batch'] * (getattr(self, '_max_image_bytes', 0) + 16 + 256 + 4096) + 8192)
            staged = ring['packed'][dslot] = torch.empty((cap,), dtype=torch.uint8, device=dev)
        staged[:need].copy_(torch.from_numpy(slot_arr[lo:hi]), non_blocking=True)
        sp = staged.data_ptr() - lo
        images = ring['images'][dslot][:b]
        if big:
            images.copy_(torch.from_numpy(arrays['images']), non_blocking=True)
        else:
            check(lib.ssd_augment_batch_dev(sp + offs['packed'], C.c_void_p(arrays['params'].ctypes.data), b, W, H, images.data_ptr(),
                                            ring['ws'][dslot].data_ptr(), stream))
        ntot = int(arrays['gcls'].shape[0])
        ws_need = lib.ssd_encode_labels_ws_bytes(ntot)
        if ring['enc_ws'][dslot] is None or ring['enc_ws'][dslot].numel() < ws_need:
            ring['enc_ws'][dslot] = torch.empty((max(ws_need, lib.ssd_encode_labels_ws_bytes(ring['batch'] * 16)),), dtype=torch.uint8, device=dev)
        labels = ring['labels'][dslot][:b]
        check(lib.ssd_encode_labels_resident(self.preset.name.encode(), self.num_classes, self.device, sp + offs['gt'], sp + offs['gcls'],
                                             sp + offs['goff'], b, ntot, labels.data_ptr(), ring['enc_ws'][dslot].data_ptr(), stream))
        return images, labels

    # ---- the generators ----------------------------------------------------------------------------------------------
    def _make_generator(self, recipe):
        def gen_batch(batch_size, num_workers=0):
            sampler = ShardSampler(recipe.total, batch_size, self.rank, self.world, self.seed + recipe.salt)
            batches = list(sampler.batches_with_count(self.epoch))
            if num_workers and num_workers > 0 and (self.device_tensors or self._upload_hook is not None):
                yield from self._prefetched(recipe, batches, self.epoch, batch_size, int(num_workers))
                return
            for idx, count in batches:
                if len(idx) == 0:              # an empty shard of a short last batch: the rank still takes the step
                    self.global_count = count
                    yield None, None, []
                    continue
                arrays, gts = recipe.plan(self.epoch, idx)
                images, labels = self._upload(arrays, gts)
                self.global_count = count
                yield images, labels, gts
        return gen_batch

    def _prefetched(self, recipe, batches, epoch, batch_size, num_workers):
        import torch
        if recipe.active:
            raise RuntimeError('one %s generator at a time: finish or close the previous one first' % recipe.which)
        slot_bytes = recipe.slot_bytes(batch_size)
        pool = recipe.pool
        if pool is not None and (pool.num_workers != num_workers or pool.slot_bytes < slot_bytes):
            pool.close()
            pool = None
        if pool is None:
            prime_anchor_table(self.preset)              # the workers must find it: they never call the GPU
            pool = recipe.pool = _WorkerPool(recipe, num_workers, slot_bytes)
        use_gpu = self._upload_hook is None
        if use_gpu:
            pool.pin(self.device)
            ring = self._device_ring(recipe, batch_size)
            fstream = ring['stream']
            dev = torch.device('cuda', self.device)
            # the device slots start out free: whatever the caller still has enqueued on batches of an earlier generator
            # (the last steps of the previous epoch) reads them and must have run before the first uploads land
            fstream.wait_stream(torch.cuda.current_stream(dev))
        pool.generation += 1
        gen = pool.generation
        nb = len(batches)
        ready = queue.Queue()
        dev_free = queue.Queue()
        for s in range(DEVICE_SLOTS):
            dev_free.put((s, None))
        cancel = threading.Event()
        stats = self.feeder_stats = dict(consumer_wait=0.0, worker_wait=0.0, slot_wait=0.0, upload=0.0, host_slot_wait=0.0, batches=0)
        clock = time.perf_counter
        inflight = []          # (host slot, event behind its upload): the slot returns to the pool when the event has passed

        def reap(block):
            """host slots whose upload has run; block: wait for the oldest one when none is free"""
            while inflight and (inflight[0][1] is None or inflight[0][1].query()):
                pool.free_slots.append(inflight.pop(0)[0])
            if block and inflight and not pool.free_slots:
                t0 = clock()
                inflight[0][1].synchronize()
                stats['host_slot_wait'] += clock() - t0
                pool.free_slots.append(inflight.pop(0)[0])

        def feeder():
            try:
                next_submit = next_upload = 0
                done = {}
                while next_upload < nb and not cancel.is_set():
                    reap(block=next_submit < nb and next_upload not in done and len(pool.outstanding) == 0)
                    while next_submit < nb and (len(batches[next_submit][0]) == 0 or pool.free_slots):
                        idx = batches[next_submit][0]
                        if len(idx) == 0:
                            done[next_submit] = None
                        else:
                            slot = pool.free_slots.pop()
                            pool.outstanding[(gen, next_submit)] = slot
                            pool.tasks.put((gen, next_submit, slot, epoch, np.asarray(idx)))
                        next_submit += 1
                    if next_upload in done:
                        item = done.pop(next_upload)
                        count = batches[next_upload][1]
                        if item is None:
                            ready.put((next_upload, None, 0, [], count, None))
                            next_upload += 1
                            continue
                        hslot, arrays, gts = item
                        t0 = clock()
                        while True:
                            try:
                                dslot, released = dev_free.get(timeout=0.2)
                                break
                            except queue.Empty:
                                if cancel.is_set():
                                    return
                        t1 = clock()
                        stats['slot_wait'] += t1 - t0
                        if use_gpu:
                            slot_arr = pool.results.array_pool[hslot]
                            in_slot = all(isinstance(v, np.ndarray) and v.base is not None and
                                          slot_arr.ctypes.data <= v.ctypes.data < slot_arr.ctypes.data + slot_arr.nbytes for v in arrays.values())
                            with torch.cuda.stream(fstream):
                                if released is not None:
                                    fstream.wait_event(released)
                                if in_slot:      # enqueue and move on: the consumer's stream waits for the event, not this thread
                                    images, labels = self._upload_async(ring, slot_arr, arrays, gts, dslot)
                                else:            # a batch that did not fit its slot came through the pipe: the serial upload
                                    images, labels = self._upload(arrays, gts, dslot, ring)
                                ev = torch.cuda.Event()
                                ev.record(fstream)
                            inflight.append((hslot, ev))
                        else:
                            images, labels = self._upload(arrays, gts, dslot)
                            ev = None
                            pool.free_slots.append(hslot)
                        stats['upload'] += clock() - t1
                        stats['batches'] += 1
                        ready.put((next_upload, dslot, len(gts), (images, labels, gts), count, ev))
                        next_upload += 1
                        continue
                    t0 = clock()
                    try:
                        tag, hslot, arrays, gts = pool.results.get(timeout=0.5)
                    except queue.Empty:
                        stats['worker_wait'] += clock() - t0
                        pool.check_alive()
                        continue
                    except WorkerError as e:
                        pool.outstanding.pop(e.tag, None)
                        if e.tag[0] != gen:
                            continue
                        raise
                    stats['worker_wait'] += clock() - t0
                    pool.outstanding.pop(tag, None)
                    if tag[0] != gen:                      # a batch of an abandoned epoch
                        pool.free_slots.append(hslot)
                        continue
                    done[tag[1]] = (hslot, arrays, gts)
            except BaseException as e:
                ready.put(e)
            finally:
                for hslot, ev in inflight:
                    if ev is not None:
                        try:
                            ev.synchronize()
                        except Exception:
                            pass
                del inflight[:]

        recipe.active = True
        th = threading.Thread(target=feeder, name='ssd-feeder-' + recipe.which, daemon=True)
        th.start()
        try:
            for k in range(nb):
                t0 = clock()
                item = ready.get()
                stats['consumer_wait'] += clock() - t0
                if isinstance(item, BaseException):
                    raise item
                seq, dslot, n, payload, count, ev = item
                assert seq == k
                if dslot is None:
                    self.global_count = count
                    yield None, None, []
                    continue
                images, labels, gts = payload
                if ev is not None:
                    torch.cuda.current_stream(dev).wait_event(ev)
                self.global_count = count
                yield images, labels, gts
                # the caller is back for the next batch: what it enqueued on its stream reads this slot
                released = None
                if use_gpu:
                    released = torch.cuda.Event()
                    released.record(torch.cuda.current_stream(dev))
                dev_free.put((dslot, released))
        finally:
            cancel.set()
            th.join()
            recipe.active = False
            # slots of results that were received but never uploaded go back; tasks still in flight are
            # recognised by their generation when they arrive
            while True:
                try:
                    item = ready.get_nowait()
                except queue.Empty:
                    break
            self._reclaim(pool, gen)

    @staticmethod
    def _reclaim(pool, gen):
        """After a generator ended (exhausted or abandoned): wait for the tasks of generation `gen` that are still with
        the workers and take their slots back."""
        while pool.outstanding:
            try:
                tag, hslot, _, _ = pool.results.get(timeout=5.0)
            except queue.Empty:
                pool.check_alive()
                continue
            except WorkerError as e:
                tag = e.tag
            pool.outstanding.pop(tag, None)
        pool.outstanding.clear()
        pool.free_slots = list(range(pool.nslots))
