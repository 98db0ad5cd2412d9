"""Batch feeder for the step loop: the counterpart of the reference's TrainingData /
DataQueue (training_data.py:35-206, data_queue.py:26-112) for the path this build covers.

The reference forks N workers that run cv2 transforms and copy batches through shared-memory
slots to ONE device.  Here every rank feeds only its own shard (parallel.ShardSampler); the label
vectors come from the HIP label encoder and, with augment=True, the images from the batch
augmentation kernels (transforms.augment_batch: the reference's train recipe, decisions on the host,
pixels on the GPU), so there is no queue to cross: a batch is born in HBM.  Dataset pickles hold
instances of the reference's own classes and there is no OpenCV here to decode files, so the data
source is synthetic: float32 preset-sized images (augment=False) or a uint8
"dataset" of variously sized images that goes through the whole recipe (augment=True).
"""
import random

import numpy as np

from .parallel import ShardSampler
from .ssdutils import encode_labels_batch, get_preset_by_name, has_positive_anchor
from .utils import Box, Point, Size, Sample

VOC_NAMES = ['aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable', 'dog',
             'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor']


class TrainingData:
    """Same attributes the drivers read from the reference's TrainingData: preset, num_classes,
    lid2name, lname2id, num_train, num_valid, train_generator, valid_generator."""

    def __init__(self, data_dir=None, preset='vgg300', num_train=64, num_valid=16, seed=1234, rank=0, world=1,
                 augment=False, sampler_trials=50, expand_prob=0.5, device=0):
        if data_dir not in (None, '', 'synthetic'):
            raise RuntimeError(f"[Errno 2] No such file or directory: '{data_dir}/training-data.pkl' "
                               '(only the synthetic source is built; SURVEY.md 8f)')   # training_data.py:49
        self.preset = get_preset_by_name(preset) if isinstance(preset, str) else preset
        self.num_classes = 20
        self.lid2name = dict(enumerate(VOC_NAMES))
        self.lname2id = {n: i for i, n in self.lid2name.items()}
        self.num_train, self.num_valid = num_train, num_valid
        self.seed, self.rank, self.world = seed, rank, world
        self.epoch = 0
        self.augment, self.device = bool(augment), device
        if self.augment:
            from . import transforms as T
            self.train_transforms = T.build_train_transforms(self.preset, self.num_classes, sampler_trials, expand_prob)
            self.valid_transforms = T.build_valid_transforms(self.preset, self.num_classes)
            self.train_generator = self._augmented_generator(num_train, 0, self.train_transforms)
            self.valid_generator = self._augmented_generator(num_valid, 1 << 20, self.valid_transforms)
            return
        self.train_generator = self._generator(num_train, 0)
        self.valid_generator = self._generator(num_valid, 1 << 20)

    # ---- augment=True: a synthetic uint8 dataset through the reference's transform recipe --------------------
    def _dataset_sample(self, index, salt):
        """Deterministic "file" #index: a uint8 BGR image of its own size + 1..5 boxes (a Sample record)."""
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
        return name, img, Sample(name, boxes, Size(W, H))

    def _augmented_generator(self, total, salt, transforms):
        from . import transforms as T
        host_tfs = [t for t in transforms if not isinstance(t, T.LabelCreatorTransform)]

        def gen_batch(batch_size, num_workers=0):
            sampler = ShardSampler(total, batch_size, self.rank, self.world, self.seed + salt)
            W, H = self.preset.image_size.w, self.preset.image_size.h
            for idx in sampler.batches(self.epoch):
                if len(idx) == 0:
                    continue
                plans, gts = [], []
                for i in idx:
                    name, img, sample = self._dataset_sample(int(i), salt)
                    host_tfs[0].images = {name: img}
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
                images = T.augment_batch(plans, W, H, device=self.device)
                bxs = [np.array([[b.center.x, b.center.y, b.size.w, b.size.h] for b in g], np.float64).reshape(-1, 4) for g in gts]
                cls = [np.array([b.labelid for b in g], np.int32) for g in gts]
                labels = encode_labels_batch(self.preset, self.num_classes, bxs, cls)
                yield images, labels, gts
        return gen_batch

    def _sample(self, index, salt):
        """Deterministic synthetic sample #index: image + 1..5 GT boxes, redrawn (<= 50 times) until
        at least one anchor is positive (training_data.py:92-98) -- checked on the encoded label."""
        rng = np.random.default_rng([self.seed, salt, index])
        H, W = self.preset.image_size.h, self.preset.image_size.w
        img = rng.integers(0, 256, (H, W, 3)).astype(np.float32)        # BGR 0..255, training_data.py:100
        n = int(rng.integers(1, 6))
        w = rng.uniform(0.1, 0.6, n); h = rng.uniform(0.1, 0.6, n)
        boxes = np.stack([rng.uniform(w / 2, 1 - w / 2), rng.uniform(h / 2, 1 - h / 2), w, h], 1)
        cls = rng.integers(0, self.num_classes, n)
        return img, boxes, cls

    def _generator(self, total, salt):
        def gen_batch(batch_size, num_workers=0):
            sampler = ShardSampler(total, batch_size, self.rank, self.world, self.seed + salt)
            for idx in sampler.batches(self.epoch):
                if len(idx) == 0:
                    continue
                imgs, bxs, cls = zip(*[self._sample(int(i), salt) for i in idx])
                labels = encode_labels_batch(self.preset, self.num_classes, list(bxs), list(cls))
                for k in range(len(idx)):          # redraw samples without a positive anchor
                    tries = 0
                    while np.count_nonzero(labels[k][:, self.num_classes]) == labels[k].shape[0] and tries < 50:
                        tries += 1
                        img, b, c = self._sample(int(idx[k]) + 7919 * tries, salt)
                        imgs = imgs[:k] + (img,) + imgs[k + 1:]; bxs = bxs[:k] + (b,) + bxs[k + 1:]; cls = cls[:k] + (c,) + cls[k + 1:]
                        labels[k] = encode_labels_batch(self.preset, self.num_classes, [b], [c])[0]
                gt = [[Box(self.lid2name[int(ci)], int(ci), Point(*map(float, bi[:2])), Size(*map(float, bi[2:])))
                       for bi, ci in zip(b, c)] for b, c in zip(bxs, cls)]
                yield np.stack(imgs), labels, gt
        return gen_batch
