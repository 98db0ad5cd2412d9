"""Host-side mirror of the reference's ssdutils.py: same names, arguments and error
behaviour; the arithmetic runs in libssdvgg_hip.so (HIP kernels on the MI355X).

  SSD_PRESETS / get_preset_by_name   ssdutils.py:32-73
  get_anchors_for_preset             ssdutils.py:76-117
  anchors2array                      ssdutils.py:120-130
  decode_boxes                       ssdutils.py:192-229
  suppress_overlaps                  ssdutils.py:310-318
plus the batched entry points the drivers use (detect_batch, encode_labels_batch).
"""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import _lib
from ._lib import lib, check, np_ptr
from .utils import Size, Point, Box, abs2prop

SSDMap = namedtuple('SSDMap', ['size', 'scale', 'aspect_ratios'])
SSDPreset = namedtuple('SSDPreset', ['name', 'image_size', 'maps', 'extra_scale', 'num_anchors'])
Anchor = namedtuple('Anchor', ['center', 'size', 'x', 'y', 'scale', 'map'])

_AR2 = [2, 0.5]
_AR4 = [2, 3, 0.5, 1. / 3.]
SSD_PRESETS = {
    'vgg300': SSDPreset('vgg300', Size(300, 300),
                        [SSDMap(Size(38, 38), 0.1, _AR2), SSDMap(Size(19, 19), 0.2, _AR4),
                         SSDMap(Size(10, 10), 0.375, _AR4), SSDMap(Size(5, 5), 0.55, _AR4),
                         SSDMap(Size(3, 3), 0.725, _AR2), SSDMap(Size(1, 1), 0.9, _AR2)], 1.075, 8732),
    'vgg512': SSDPreset('vgg512', Size(512, 512),
                        [SSDMap(Size(64, 64), 0.07, _AR2), SSDMap(Size(32, 32), 0.15, _AR4),
                         SSDMap(Size(16, 16), 0.3, _AR4), SSDMap(Size(8, 8), 0.45, _AR4),
                         SSDMap(Size(4, 4), 0.6, _AR4), SSDMap(Size(2, 2), 0.75, _AR2),
                         SSDMap(Size(1, 1), 0.9, _AR2)], 1.05, 24564),
}
IMG1000 = Size(1000, 1000)


def set_device(device):
    """GPU ordinal the free functions of this package run on (a handle-owning SSDVGG has its own)."""
    _lib.set_device(device)


def get_preset_by_name(pname):
    if pname not in SSD_PRESETS:
        raise RuntimeError('No such preset: ' + pname)
    return SSD_PRESETS[pname]


def _pname(preset):
    return (preset if isinstance(preset, str) else preset.name).encode()


def anchors_array(preset):
    """[A,4] float64 (cx, cy, w, h) from the HIP anchor kernel."""
    p = get_preset_by_name(preset if isinstance(preset, str) else preset.name)
    out = np.empty((p.num_anchors, 4), np.float64)
    check(lib.ssd_anchors(_pname(p), _lib.device(), np_ptr(out)))
    return out


def get_anchors_for_preset(preset):
    """List of Anchor records in the reference's order (map -> box type -> row -> col)."""
    arr = anchors_array(preset)
    out = []
    k = 0
    for m, mp in enumerate(preset.maps):
        fk = mp.size[0]
        for _t in range(2 + len(mp.aspect_ratios)):
            for j in range(fk):
                for i in range(fk):
                    a = arr[k]
                    out.append(Anchor(Point(float(a[0]), float(a[1])), Size(float(a[2]), float(a[3])), i, j, mp.scale, m))
                    k += 1
    return out


def anchors2array(anchors, img_size):
    """[A,4] float64 (xmin, xmax, ymin, ymax) truncated ints.  The HIP kernel serves the
    reference's only call (img_size == Size(1000, 1000), a full preset); anything else is
    converted with the scalar helper."""
    from .utils import prop2abs
    if tuple(img_size) == (1000, 1000):
        for p in SSD_PRESETS.values():
            if len(anchors) == p.num_anchors:
                out = np.empty((p.num_anchors, 4), np.int32)
                check(lib.ssd_anchors_abs(_pname(p), _lib.device(), np_ptr(out)))
                return out.astype(np.float64)
    return np.array([prop2abs(a.center, a.size, img_size) for a in anchors], np.float64).reshape(-1, 4)


def _preset_for(num_anchors):
    for p in SSD_PRESETS.values():
        if p.num_anchors == num_anchors:
            return p
    raise ValueError(f'no preset with {num_anchors} anchors')


def detect_batch(pred, preset, confidence_threshold=0.01, detections_cap=200, max_out=None, nms=True, out_cap=None):
    """decode_boxes (+ suppress_overlaps)[:max_out] for pred [b, A, C+5] on the GPU.
    Returns a list (one per image) of dicts: conf f32 [n], cls i32 [n], idx i32 [n], box i32 [n,4]."""
    pred = np.ascontiguousarray(pred, np.float32)
    if pred.ndim == 2:
        pred = pred[None]
    b, A, nv = pred.shape
    p = _preset_for(A) if preset is None else preset
    if p.num_anchors != A:
        raise ValueError(f'pred has {A} anchors, preset {p.name} has {p.num_anchors}')
    cap = -1 if detections_cap is None else int(detections_cap)
    mo = -1 if max_out is None else int(max_out)
    if out_cap is None:
        out_cap = A if cap < 0 else max(cap, 1)
        if mo >= 0:
            out_cap = max(min(out_cap, mo), 1)
    count = np.zeros(b, np.int32)
    conf = np.zeros((b, out_cap), np.float32)
    cls = np.zeros((b, out_cap), np.int32)
    idx = np.zeros((b, out_cap), np.int32)
    box = np.zeros((b, out_cap, 4), np.int32)
    check(lib.ssd_decode_nms(_pname(p), nv - 5, _lib.device(), np_ptr(pred), b, float(confidence_threshold), cap, mo, out_cap,
                             1 if nms else 0, np_ptr(count), np_ptr(conf), np_ptr(cls), np_ptr(idx), np_ptr(box)))
    out = []
    for i in range(b):
        n = min(int(count[i]), out_cap)
        out.append(dict(conf=conf[i, :n], cls=cls[i, :n], idx=idx[i, :n], box=box[i, :n]))
    return out


def boxes_from_detection(det, lid2name={}):
    """dict from detect_batch -> the reference's list of (confidence, Box)."""
    res = []
    for c, k, b in zip(det['conf'], det['cls'], det['box']):
        center, size = abs2prop(int(b[0]), int(b[1]), int(b[2]), int(b[3]), IMG1000)
        k = int(k)
        res.append((np.float32(c), Box(lid2name.get(k), k, center, size)))
    return res


class DecodedBoxes(list):
    """decode_boxes' list of (confidence, Box); remembers the prediction it came from so that
    suppress_overlaps can run decode + NMS as ONE fused GPU pass instead of a second upload."""
    source = None


def decode_boxes(pred, anchors, confidence_threshold=0.01, lid2name={}, detections_cap=200):
    """ssdutils.py:192-229 for one image.  `anchors` only selects the preset (its length).
    pred is not modified (the reference clamps offsets > 100 in place)."""
    pred = np.asarray(pred, np.float32)
    p = _preset_for(len(anchors))
    det = detect_batch(pred, p, confidence_threshold, detections_cap, None, nms=False)[0]
    out = DecodedBoxes(boxes_from_detection(det, lid2name))
    out.source = (pred, p, confidence_threshold, detections_cap, lid2name, tuple(out))
    return out


def _nms_list(boxes, groups, overlap_threshold):
    """ssd_nms_boxes on a list of (confidence, Box): the reference's prop2abs on the 1000 grid
    (ssdutils.py:244-249) is scalar host arithmetic, the sort and the suppression run on the GPU."""
    from .utils import prop2abs
    n = len(boxes)
    if n == 0:
        return []
    if n > 65535:
        raise ValueError('at most 65535 boxes per call')
    arr = np.array([prop2abs(b[1].center, b[1].size, IMG1000) for b in boxes], np.int64).reshape(n, 4)
    arr = np.ascontiguousarray(np.clip(arr, -(1 << 30), 1 << 30), np.int32)
    conf = np.ascontiguousarray([b[0] for b in boxes], np.float32)
    grp = None if groups is None else np.ascontiguousarray(groups, np.int32)
    keep = np.empty(n, np.int32)
    nk = C.c_int(0)
    check(lib.ssd_nms_boxes(_lib.device(), n, np_ptr(arr), np_ptr(conf), np_ptr(grp), float(overlap_threshold), np_ptr(keep),
                            C.byref(nk)))
    return [boxes[int(i)] for i in keep[:nk.value]]


def suppress_overlaps(boxes):
    """ssdutils.py:310-318: per-class NMS at 0.45, classes in first-appearance order.  An untouched list
    from decode_boxes goes back through the fused GPU decode + NMS (same result, no second upload);
    any other list of (confidence, Box) runs through ssd_nms_boxes."""
    src = getattr(boxes, 'source', None)
    if src is not None and len(boxes) == len(src[5]) and all(a is b for a, b in zip(boxes, src[5])):
        pred, p, thr, cap, lid2name = src[:5]
        det = detect_batch(pred, p, thr, cap, None, nms=True)[0]
        return boxes_from_detection(det, lid2name)
    rank = {}
    groups = [rank.setdefault(b[1].labelid, len(rank)) for b in boxes]
    return _nms_list(list(boxes), groups, 0.45)


# ---- scalar / per-box helpers of the reference kept under their names (ssdutils.py:133-189) ----
def box2array(box, img_size):
    """ssdutils.py:133-135"""
    from .utils import prop2abs
    return np.array(prop2abs(box.center, box.size, img_size))


def jaccard_overlap(box_arr, anchors_arr):
    """ssdutils.py:138-152 (+1 pixel convention) on the GPU (ssd_jaccard_overlap)."""
    a = np.ascontiguousarray(anchors_arr, np.float64).reshape(-1, 4)
    box = np.ascontiguousarray(box_arr, np.float64).reshape(4)
    iou = np.empty(a.shape[0], np.float64)
    check(lib.ssd_jaccard_overlap(_lib.device(), np_ptr(box), np_ptr(a), a.shape[0], np_ptr(iou)))
    return iou


def compute_overlap(box_arr, anchors_arr, threshold):
    """ssdutils.py:155-170: Overlap(best, good) with Score(idx, score) records."""
    from .utils import Score, Overlap
    iou = jaccard_overlap(box_arr, anchors_arr)
    best_idx = int(np.argmax(iou))
    best = Score(best_idx, iou[best_idx]) if iou[best_idx] > threshold else None
    return Overlap(best, [Score(int(i), iou[i]) for i in np.nonzero(iou > threshold)[0]])


def compute_location(box, anchor):
    """ssdutils.py:173-179"""
    from math import log
    arr = np.zeros((4))
    arr[0] = (box.center.x - anchor.center.x) / anchor.size.w * 10
    arr[1] = (box.center.y - anchor.center.y) / anchor.size.h * 10
    arr[2] = log(box.size.w / anchor.size.w) * 5
    arr[3] = log(box.size.h / anchor.size.h) * 5
    return arr


def decode_location(box, anchor):
    """ssdutils.py:182-189, without the in-place clamp of the caller's array."""
    from math import exp
    box = np.where(box > 100, 100, box).astype(box.dtype)
    x = box[0] / 10 * anchor.size.w + anchor.center.x
    y = box[1] / 10 * anchor.size.h + anchor.center.y
    return Point(x, y), Size(exp(box[2] / 5) * anchor.size.w, exp(box[3] / 5) * anchor.size.h)


def non_maximum_suppression(boxes, overlap_threshold):
    """ssdutils.py:232-307 for one list of (confidence, Box) (the caller has grouped by class): greedy NMS
    with intersection/union > overlap_threshold (any threshold), selected boxes in descending confidence.
    Equal confidences: the earlier box first (the reference's argsort leaves that unspecified)."""
    return _nms_list(list(boxes), None, overlap_threshold)


_ANCHORS_ABS = {}


def prime_anchor_table(preset, table=None):
    """The integer anchor table (xmin, xmax, ymin, ymax on the 1000-pixel grid, ssdutils.py:120-130) of the redraw
    test, computed once per process by the anchor kernel.  The feeder calls this before it forks its workers: they
    inherit the table and never call the GPU.  `table` installs a given [A, 4] array instead (CPU tests)."""
    key = preset.name
    if table is not None:
        _ANCHORS_ABS[key] = np.ascontiguousarray(table, np.float64).reshape(preset.num_anchors, 4)
    if key not in _ANCHORS_ABS:
        out = np.empty((preset.num_anchors, 4), np.int32)
        check(lib.ssd_anchors_abs(_pname(preset), _lib.device(), np_ptr(out)))
        _ANCHORS_ABS[key] = out.astype(np.float64)
    return _ANCHORS_ABS[key]


def has_positive_anchor(preset, boxes):
    """Would LabelCreatorTransform mark at least one anchor positive for these Box records?  That is the only
    thing the reference's redraw loop looks at (training_data.py:92-95: num_bg < rows), and an anchor turns
    positive iff its IoU (+1 pixel, 1000-pixel grid) with some box exceeds 0.5 (transforms.py:76-107,
    ssdutils.py:152-169).  A handful of boxes against the cached integer anchor table, on the host."""
    from .utils import prop2abs, Size
    an = prime_anchor_table(preset)
    area_a = (an[:, 1] - an[:, 0] + 1) * (an[:, 3] - an[:, 2] + 1)
    grid = Size(1000, 1000)
    for b in boxes:
        xmin, xmax, ymin, ymax = prop2abs(b.center, b.size, grid)
        w = np.maximum(0, np.minimum(xmax, an[:, 1]) - np.maximum(xmin, an[:, 0]) + 1)
        h = np.maximum(0, np.minimum(ymax, an[:, 3]) - np.maximum(ymin, an[:, 2]) + 1)
        inter = w * h
        iou = inter / ((xmax - xmin + 1) * (ymax - ymin + 1) + area_a - inter)
        if (iou > 0.5).any():
            return True
    return False


def _pack_gt(gt_boxes_list, gt_cls_list):
    b = len(gt_boxes_list)
    offs = np.zeros(b + 1, np.int32)
    for i, g in enumerate(gt_boxes_list):
        offs[i + 1] = offs[i] + len(g)
    gt = np.concatenate([np.asarray(g, np.float64).reshape(-1, 4) for g in gt_boxes_list] + [np.zeros((0, 4))], 0)
    cls = np.concatenate([np.asarray(c, np.int32).reshape(-1) for c in gt_cls_list] + [np.zeros((0,), np.int32)], 0)
    return np.ascontiguousarray(gt, np.float64), np.ascontiguousarray(cls, np.int32), offs


def encode_labels_batch(preset, num_classes, gt_boxes_list, gt_cls_list):
    """LabelCreatorTransform for a batch on the GPU: lists (one per image) of [n,4] float64
    proportional (cx, cy, w, h) and [n] class ids -> [b, A, num_classes+5] float32 (host)."""
    b = len(gt_boxes_list)
    gt, cls, offs = _pack_gt(gt_boxes_list, gt_cls_list)
    vec = np.empty((b, preset.num_anchors, num_classes + 5), np.float32)
    check(lib.ssd_encode_labels(_pname(preset), int(num_classes), _lib.device(), np_ptr(gt), np_ptr(cls), np_ptr(offs), b, np_ptr(vec)))
    return vec


def encode_labels_batch_dev(preset, num_classes, gt_boxes_list, gt_cls_list, device=None, out=None):
    """The same, left in HBM: a float32 CUDA tensor [b, A, num_classes+5] on `device` (default: this module's
    device).  The few ground-truth boxes go up, 873 KB per image never come down (ssd_encode_labels_dev)."""
    import torch
    device = _lib.device() if device is None else int(device)
    b = len(gt_boxes_list)
    gt, cls, offs = _pack_gt(gt_boxes_list, gt_cls_list)
    if out is None:
        out = torch.empty((b, preset.num_anchors, num_classes + 5), dtype=torch.float32, device=torch.device('cuda', device))
    check(lib.ssd_encode_labels_dev(_pname(preset), int(num_classes), device, np_ptr(gt), np_ptr(cls), np_ptr(offs), b,
                                    out.data_ptr(), torch.cuda.current_stream(device).cuda_stream))
    return out
