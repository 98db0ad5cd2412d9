"""Batch feeder for the step loop: the counterpart of the reference's TrainingData /
DataQueue (training_data.py:35-206, data_queue.py:26-112) for the path this build covers.

The reference forks N workers that run cv2 transforms and copy batches through shared-memory
slots to ONE device.  Here every rank feeds only its own shard (parallel.ShardSampler) and the
label vectors come from the HIP label encoder, so there is no queue to cross.  Dataset pickles
hold instances of the reference's own classes and the cv2 augmentation pipeline is out of
scope: the data source here is synthetic.
"""
import numpy as np

from .parallel import ShardSampler
from .ssdutils import encode_labels_batch, get_preset_by_name
from .utils import Box, Point, Size, Sample

VOC_NAMES = ['aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable', 'dog',
             'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor']


class TrainingData:
    """Same attributes the drivers read from the reference's TrainingData: preset, num_classes,
    lid2name, lname2id, num_train, num_valid, train_generator, valid_generator."""

    def __init__(self, data_dir=None, preset='vgg300', num_train=64, num_valid=16, seed=1234, rank=0, world=1):
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
        self.train_generator = self._generator(num_train, 0)
        self.valid_generator = self._generator(num_valid, 1 << 20)

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
