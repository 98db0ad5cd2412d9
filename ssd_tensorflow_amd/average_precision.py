"""Mirror of the reference's average_precision.py (APs2mAP :30-42, APCalculator :45-192) over the HIP
kernel behind ssd_average_precision.  Same class, same methods, same dict-of-label results."""
from collections import defaultdict

import numpy as np

from . import _lib
from ._lib import lib, check, np_ptr
from .utils import Size, prop2abs

IMG_SIZE = Size(1000, 1000)


def APs2mAP(aps):
    """Mean of the APs over all classes (summed in the dict's order)."""
    num_classes = 0.
    sum_ap = 0.
    for _, v in aps.items():
        sum_ap += v
        num_classes += 1
    if num_classes == 0:
        return 0
    return sum_ap / num_classes


class APCalculator:
    """VOC07 11-point average precision (see the reference's docstring for its peculiarities)."""

    def __init__(self, minoverlap=0.5):
        self.minoverlap = minoverlap
        self.clear()

    def add_detections(self, gt_boxes, boxes):
        """gt_boxes: list of Box; boxes: list of (confidence, Box) with a correctly set label."""
        sample_id = len(self.gt_boxes)
        self.gt_boxes.append(gt_boxes)
        for conf, box in boxes:
            self.det_params.append(prop2abs(box.center, box.size, IMG_SIZE))
            self.det_confidence.append(conf)
            self.det_labels.append(box.label)
            self.det_sample_ids.append(sample_id)

    def compute_aps(self):
        """{label: AP} for every label that has ground truth, in first-appearance order."""
        label_id = {}
        gb, gk, gs = [], [], []
        for sample_id, boxes in enumerate(self.gt_boxes):
            for box in boxes:
                k = label_id.setdefault(box.label, len(label_id))
                gb.append(prop2abs(box.center, box.size, IMG_SIZE)); gk.append(k); gs.append(sample_id)
        if not label_id:
            return {}
        ncls = len(label_id)
        keep = [i for i, l in enumerate(self.det_labels) if l in label_id]      # detections of classes without ground truth are ignored
        db = np.ascontiguousarray([self.det_params[i] for i in keep], np.float32).reshape(-1, 4)
        dc = np.ascontiguousarray([self.det_confidence[i] for i in keep], np.float32)
        dk = np.ascontiguousarray([label_id[self.det_labels[i]] for i in keep], np.int32)
        ds = np.ascontiguousarray([self.det_sample_ids[i] for i in keep], np.int32)
        gb = np.ascontiguousarray(gb, np.float64).reshape(-1, 4)
        gk = np.ascontiguousarray(gk, np.int32); gs = np.ascontiguousarray(gs, np.int32)
        ap = np.zeros(ncls, np.float64); present = np.zeros(ncls, np.int32)
        check(lib.ssd_average_precision(_lib.device(), len(dc), np_ptr(db), np_ptr(dc), np_ptr(dk), np_ptr(ds), len(gk), np_ptr(gb),
                                        np_ptr(gk), np_ptr(gs), ncls, float(self.minoverlap), np_ptr(ap), np_ptr(present)))
        return {label: float(ap[k]) for label, k in label_id.items() if present[k]}

    def state(self):
        """Everything add_detections has collected, as plain lists (picklable: one rank's share of a data-parallel run)."""
        return dict(det_params=self.det_params, det_confidence=self.det_confidence, det_labels=self.det_labels,
                    det_sample_ids=self.det_sample_ids, gt_boxes=self.gt_boxes)

    def merge(self, state):
        """Append another calculator's state(); its sample ids are shifted behind the samples already held."""
        base = len(self.gt_boxes)
        self.gt_boxes.extend(state['gt_boxes'])
        self.det_params.extend(state['det_params'])
        self.det_confidence.extend(state['det_confidence'])
        self.det_labels.extend(state['det_labels'])
        self.det_sample_ids.extend(i + base for i in state['det_sample_ids'])

    def clear(self):
        self.det_params = []
        self.det_confidence = []
        self.det_labels = []
        self.det_sample_ids = []
        self.gt_boxes = []
