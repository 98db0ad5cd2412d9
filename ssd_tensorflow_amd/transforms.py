"""LabelCreatorTransform of the reference's transforms.py:57-114, backed by the HIP label
encoder (ssd_encode_labels).  Same constructor keywords (preset, num_classes) and the same
(data, label, gt) -> (data, label, gt) call convention."""
import numpy as np

from .ssdutils import encode_labels_batch


class Transform:
    """transforms.py:31-35"""
    def __init__(self, **kwargs):
        for arg, val in kwargs.items():
            setattr(self, arg, val)
        self.initialized = False


class LabelCreatorTransform(Transform):
    """Parameters: preset, num_classes"""
    def __call__(self, data, label, gt):
        boxes = np.array([[b.center.x, b.center.y, b.size.w, b.size.h] for b in gt.boxes], np.float64).reshape(-1, 4)
        cls = np.array([b.labelid for b in gt.boxes], np.int32)
        vec = encode_labels_batch(self.preset, self.num_classes, [boxes], [cls])[0]
        return data, vec, gt
