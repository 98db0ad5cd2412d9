"""Batch feeder for the step loop: the counterpart of the reference's TrainingData /
DataQueue (training_data.py:35-206, data_queue.py:26-112) for the path this build covers.

The reference forks N workers that run cv2 transforms and copy batches through shared-memory
slots to ONE device.  Here every rank feeds only its own shard (parallel.ShardSampler) and a batch is
born in HBM: the images come out of the batch augmentation kernels (transforms.augment_batch: the
reference's transform recipe, decisions on the host, pixels on the GPU), the label vectors out of the
HIP label encoder (ssd_encode_labels_dev) -- only the images' source bytes and a handful of boxes go up,
nothing of a batch ever comes down.  A generator yields `(images, labels, gt_boxes)` like the
reference's, with `images` / `labels` float32 CUDA tensors (device_tensors=False: numpy arrays).

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
import numpy as np

from .parallel import ShardSampler
from .ssdutils import encode_labels_batch, encode_labels_batch_dev, get_preset_by_name, has_positive_anchor
from .utils import Box, Point, Size, Sample, load_data_source

VOC_NAMES = ['aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable', 'dog',
             'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor']


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
        from . import transforms as T
        if data_dir not in (None, '', 'synthetic'):
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
            self.train_generator = self._augmented_generator(lambda i: self.train_samples[i], self.num_train, 0, self.train_transforms)
            self.valid_generator = self._augmented_generator(lambda i: self.valid_samples[i], self.num_valid, 1 << 20, self.valid_transforms)
            return
        self.num_classes = 20
        self.label_colors = {}
        self.lid2name = dict(enumerate(VOC_NAMES))
        self.lname2id = {n: i for i, n in self.lid2name.items()}
        self.num_train, self.num_valid = num_train, num_valid
        self.augment = bool(augment)
        if self.augment:
            self.train_transforms = T.build_train_transforms(self.preset, self.num_classes, sampler_trials, expand_prob)
            self.valid_transforms = T.build_valid_transforms(self.preset, self.num_classes)
            self.train_generator = self._augmented_generator(lambda i: self._dataset_sample(i, 0), num_train, 0, self.train_transforms)
            self.valid_generator = self._augmented_generator(lambda i: self._dataset_sample(i, 1 << 20), num_valid, 1 << 20,
                                                             self.valid_transforms)
            self.train_samples = [self._dataset_sample(i, 0)[1] for i in range(num_train)]
            self.valid_samples = [self._dataset_sample(i, 1 << 20)[1] for i in range(num_valid)]
            return
        self.train_generator = self._generator(num_train, 0)
        self.valid_generator = self._generator(num_valid, 1 << 20)
        self.train_samples = self.valid_samples = None

    # ---- labels: encoded once per batch on the GPU, left there ---------------------------------------------------
    def _labels(self, gts):
        bxs = [np.array([[b.center.x, b.center.y, b.size.w, b.size.h] for b in g], np.float64).reshape(-1, 4) for g in gts]
        cls = [np.array([b.labelid for b in g], np.int32) for g in gts]
        if self.device_tensors:
            return encode_labels_batch_dev(self.preset, self.num_classes, bxs, cls, self.device)
        return encode_labels_batch(self.preset, self.num_classes, bxs, cls)

    # ---- a dataset of variously sized uint8 images through the reference's transform recipe ---------------------
    def _dataset_sample(self, index, salt):
        """Deterministic synthetic "file" #index: ({name: uint8 BGR image}, Sample record with 1..5 boxes)."""
        rng = np.random.default_rng([self.seed, salt, index, 77])
        W, H = int(rng.integers(200, 640)), int(rng.integers(200, 640))
        img = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
        n = int(rng.integers(1, 6))
        w = rng.uniform(0.1, 0.6, n); h = rng.uniform(0.1, 0.6, n)
        cx = rng.uniform(w / 2, 1 - w / 2); cy = rng.uniform(h / 2, 1 - h / 2)
        cls = rng.integers(0, self.num_classes, n)
        boxes = [Box(self.lid2name[int(c)], int(c), Point(float(x), float(y)), Size(float(ww), float(hh)))
                 for x, y, ww, hh, c in zip(cx, cy, w, h, cls)]
        name = 'synthetic/%d/%d' % (salt, index)
        return {name: img}, Sample(name, boxes, Size(W, H))

    def _augmented_generator(self, sample_at, total, salt, transforms):
        from . import transforms as T
        host_tfs = [t for t in transforms if not isinstance(t, T.LabelCreatorTransform)]
        loader = host_tfs[0]
        preset_images = getattr(loader, 'images', None)

        def gen_batch(batch_size, num_workers=0):
            sampler = ShardSampler(total, batch_size, self.rank, self.world, self.seed + salt)
            W, H = self.preset.image_size.w, self.preset.image_size.h
            for idx, count in sampler.batches_with_count(self.epoch):
                self.global_count = count
                if len(idx) == 0:              # an empty shard of a short last batch: the rank still takes the step
                    yield None, None, []
                    continue
                plans, gts = [], []
                for i in idx:
                    s = sample_at(int(i))
                    if not isinstance(s, Sample):                # synthetic: the pixels travel with the record
                        loader.images, sample = s
                    else:
                        loader.images, sample = preset_images, s
                    # run_transforms until at least one anchor is positive, at most 50 times (training_data.py:88-98);
                    # the label of a try is only LOOKED at there (num_bg < rows), so the test runs on the host and the
                    # label vectors of the whole batch are encoded once, on the GPU, below
                    for _ in range(50):
                        args = (None, None, sample)
                        for t in host_tfs:
                            args = t(*args)
                        if has_positive_anchor(self.preset, args[2].boxes):
                            break
                    plans.append(args[0]); gts.append(args[2].boxes)
                loader.images = preset_images
                images = T.augment_batch(plans, W, H, device=self.device)
                labels = self._labels(gts)
                if not self.device_tensors:
                    images = images.cpu().numpy()
                yield images, labels, gts
        return gen_batch

    # ---- preset-sized float32 synthetic images (no transform recipe) ------------------------------------------------
    def _sample(self, index, salt):
        """Deterministic synthetic sample #index: image + 1..5 GT boxes."""
        rng = np.random.default_rng([self.seed, salt, index])
        H, W = self.preset.image_size.h, self.preset.image_size.w
        img = rng.integers(0, 256, (H, W, 3)).astype(np.float32)        # BGR 0..255, training_data.py:100
        n = int(rng.integers(1, 6))
        w = rng.uniform(0.1, 0.6, n); h = rng.uniform(0.1, 0.6, n)
        boxes = np.stack([rng.uniform(w / 2, 1 - w / 2), rng.uniform(h / 2, 1 - h / 2), w, h], 1)
        cls = rng.integers(0, self.num_classes, n)
        gt = [Box(self.lid2name[int(ci)], int(ci), Point(*map(float, bi[:2])), Size(*map(float, bi[2:]))) for bi, ci in zip(boxes, cls)]
        return img, gt

    def _generator(self, total, salt):
        def gen_batch(batch_size, num_workers=0):
            import torch
            sampler = ShardSampler(total, batch_size, self.rank, self.world, self.seed + salt)
            for idx, count in sampler.batches_with_count(self.epoch):
                self.global_count = count
                if len(idx) == 0:
                    yield None, None, []
                    continue
                imgs, gts = [], []
                for i in idx:
                    # redrawn (<= 50 times) until at least one anchor is positive (training_data.py:92-98)
                    for tries in range(50):
                        img, gt = self._sample(int(i) + 7919 * tries, salt)
                        if has_positive_anchor(self.preset, gt):
                            break
                    imgs.append(img); gts.append(gt)
                images = np.stack(imgs)
                labels = self._labels(gts)
                if self.device_tensors:
                    images = torch.from_numpy(images).to(torch.device('cuda', self.device), non_blocking=True)
                yield images, labels, gts
        return gen_batch
