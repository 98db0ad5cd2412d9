"""The reference's transforms.py (:31-392) for the GPU feed path, same class names, same keyword
parameters, same `(data, label, gt) -> (data, label, gt)` call convention, same draws from Python's
`random` in the same order -- but no transform touches a pixel.  Between ImageLoaderTransform and the
batch kernel `data` is an ImagePlan: the loaded uint8 image plus the list of decisions taken so far
(brightness delta, distort chain, channel order, expand offsets, crop window, flip, resize algorithm).
`augment_batch()` then executes a whole batch of plans in two HIP launches (csrc/augment.hip,
ssd_augment_batch_dev): every photometric step is pointwise and every geometric step an index map, so
the chain collapses into one gather per output pixel.  The ground-truth boxes are transformed here, on
the host, exactly as the reference does (transform_box / transform_gt, transforms.py:236-270).

LabelCreatorTransform (transforms.py:57-114) is backed by the HIP label encoder.

The plan's canonical form is the reference's own recipe order (process_dataset.py:126-136): brightness -> distort chain ->
channel reorder -> expand -> crop -> flip -> resize.  Round 5: transforms composed in OTHER orders are rewritten into that
form wherever the composition is one (tests/test_augment.py runs them against the free composition of the oracle's pixel
operations): a second photometric pass -- any brightness / contrast / saturation / hue / reorder after the canonical slots
are taken -- becomes a list of up to 8 extra pointwise steps the kernel runs behind the canonical chain; photometric steps
after a flip commute with it; a crop, an expand or another flip after a flip are mirrored index maps; a second expand adds
its offsets.

Round 6: the rest of the orders a user can write.  A pointwise step behind an expand acts on the image AND on the canvas: it
joins the extra steps, and the plan remembers from which extra step on the canvas's mean value is transformed too (a Brightness /
Contrast there turns the float64 canvas back into uint8, transforms.py:168-175, so the resize rounds again).  Hue / Saturation
behind a crop or an expand index ROWS 0 / 1 of the array they are handed: the step carries the source row that is that array's
row 0.  An expand behind a crop shows canvas where the crop cut: the plan keeps the window of the source image that is still
visible.  ExpandTransform's mean_value travels with the plan.  Pointwise steps behind ResizeTransform run on the resized pixel,
a flip behind it mirrors the output columns.  What still raises: Hue / Saturation on a floating-point array (cv2.cvtColor has no
CV_64F path: the reference raises cv2.error there, this mirror RuntimeError); a crop, an expand or a second resize behind
ResizeTransform, a second expand behind steps that followed the first, two expands with different mean values, more than 16
extra / 4 post steps (NotImplementedError: the batch kernel has no intermediate image to run them on).
"""
import ctypes as C
import os
import random
from math import sqrt

import numpy as np

from . import _lib
from ._lib import lib, check
from .ssdutils import encode_labels_batch
from .utils import Size, Sample, Point, Box, abs2prop, prop2abs

MAX_EXTRA = 16       # ssd_augment_params.extra_kind / extra_val
MAX_POST = 4         # ssd_augment_params.post_kind / post_val

# cv2's interpolation enum values (the mirror does not import cv2)
INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4 = 0, 1, 2, 3, 4


class ImagePlan:
    """Stands in for the ndarray `data` of the reference between loading and the batch kernel."""

    def __init__(self, image):
        image = np.ascontiguousarray(image)
        if image.dtype != np.uint8 or image.ndim != 3 or image.shape[2] != 3:
            raise ValueError('ImagePlan needs a uint8 [H, W, 3] BGR image, got %s %s' % (image.dtype, image.shape))
        self.image = image
        self.src = Size(image.shape[1], image.shape[0])
        self.brightness = None
        self.distort = []              # [(kind, value)] kind: 0 contrast, 1 saturation, 2 hue
        self.reorder = [0, 1, 2]
        self.extra = []                # further pointwise steps behind the canonical chain, in order: (kind, value) with kind
                                       # 0..2 as above, 3 brightness (value = delta), 4 reorder (value = [c0, c1, c2])
        self.extra_r0 = []             # per extra step: the source row that was row 0 of the array the step was handed (Hue / Saturation)
        self.expand = None             # (Size new, h_off, w_off); offsets of the SOURCE image in the canvas (negative behind a crop)
        self.crop = None               # (x0, y0, w, h) in the (expanded) frame, before the flip
        self.flip = False
        self.resize = None             # (width, height, algorithm)
        # round 6 (any order of the reference's transforms):
        self.fill_from = None          # canvas pixels exist (an expand happened) and receive the extra steps [fill_from:]
        self.mean = [104, 117, 123]    # ExpandTransform.mean_value
        self.is_float = False          # the array is float64 at this point of the chain (expand; Brightness / Contrast make it uint8 again)
        self.is_float_at_resize = None     # ... when cv2.resize was handed it: a uint8 array is rounded and saturated (None: a plan built by hand -- float iff expanded)
        self.clip = None               # (x0, y0, x1, y1): the window of the source image still visible (an expand behind a crop)
        self.post = []                 # pointwise steps behind the resize
        self.out_flip = False          # a flip behind the resize

    # the frame the next geometric transform sees
    @property
    def size(self):
        if self.crop is not None:
            return Size(self.crop[2], self.crop[3])
        return self.expand[0] if self.expand is not None else self.src

    @property
    def shape(self):
        s = Size(self.resize[0], self.resize[1]) if self.resize is not None else self.size
        return (s.h, s.w, 3)

    def _row0(self):
        """the source row that is row 0 of the array a transform is handed at this point of the chain"""
        return (self.crop[1] if self.crop is not None else 0) - (self.expand[1] if self.expand is not None else 0)

    def _pointwise(self, kind, value, what):
        """Route one pointwise step (0 contrast, 1 saturation, 2 hue, 3 brightness, 4 reorder): a canonical slot while the order is
        the recipe's, the extra list behind it, the post list behind the resize.  Returns True when the caller should fill the
        canonical slot itself."""
        if kind in (1, 2) and self.is_float:
            # cv2.cvtColor(float64 image, COLOR_BGR2HSV): "Unsupported depth of input image ... 'depth' is 6 (CV_64F)"
            raise RuntimeError(what + ' on a floating-point image (behind ExpandTransform): cv2.cvtColor supports CV_8U / CV_16U / CV_32F only, '
                               'the reference raises cv2.error here')
        if self.resize is not None:
            if len(self.post) >= MAX_POST:
                raise NotImplementedError(what + ': the batch kernel runs at most %d steps behind the resize' % MAX_POST)
            self.post.append((kind, value))
        else:
            # (a flip is a pure permutation of columns and the Hue / Saturation row quirk indexes ROWS: pointwise steps commute with it;
            # brightness, contrast and the channel order act on every pixel alike, so they commute with a crop as well; hue / saturation
            # act on image ROWS 0 / 1 of the current array, transforms.py:201-203,218-220, which a crop or an expand moves)
            canonical = self.expand is None and (kind in (0, 3, 4) or self.crop is None)
            if canonical:
                return True
            self._extra_step(kind, value, what)
        if kind in (0, 3):
            self.is_float = False          # data.astype(np.uint8), transforms.py:172,186
        return False

    def _extra_step(self, kind, value, what):
        if len(self.extra) >= MAX_EXTRA:
            raise NotImplementedError(what + ': the batch kernel runs at most %d photometric steps behind the canonical chain' % MAX_EXTRA)
        self.extra.append((kind, value))
        self.extra_r0.append(self._row0() if kind in (1, 2) else 0)

    def _geometric_ok(self, what):
        if self.resize is not None:
            raise NotImplementedError(what + ' after ResizeTransform is outside the order the batch kernel runs (it has no intermediate image)')


class Transform:
    """transforms.py:31-35"""
    def __init__(self, **kwargs):
        for arg, val in kwargs.items():
            setattr(self, arg, val)
        self.initialized = False


def load_image_bgr(filename):
    """cv2.imread(filename) without OpenCV: a .npy file holds the uint8 BGR array itself; anything else is
    decoded by Pillow (RGB -> BGR).  Decoder parity with OpenCV is unpinned (no cv2 in the build container)."""
    if filename.endswith('.npy'):
        return np.load(filename)
    if os.path.exists(filename + '.npy'):
        return np.load(filename + '.npy')
    try:
        from PIL import Image
    except ImportError:
        raise RuntimeError('cannot decode %r: neither OpenCV nor Pillow is available (use .npy images)' % (filename,))
    with Image.open(filename) as im:
        return np.ascontiguousarray(np.asarray(im.convert('RGB'))[:, :, ::-1])


class ImageLoaderTransform(Transform):
    """transforms.py:38-43 reads gt.filename with cv2.imread.  Here the pixels come from `images` (a mapping
    filename -> uint8 BGR array) when given, else from load_image_bgr (a .npy array or a Pillow decode)."""
    def __call__(self, data, label, gt):
        images = getattr(self, 'images', None)
        if images is not None and gt.filename in images:
            return ImagePlan(images[gt.filename]), label, gt
        if isinstance(gt.filename, str):
            return ImagePlan(load_image_bgr(gt.filename)), label, gt
        raise RuntimeError('cannot load %r' % (gt.filename,))


class LabelCreatorTransform(Transform):
    """Parameters: preset, num_classes"""
    def __call__(self, data, label, gt):
        boxes = np.array([[b.center.x, b.center.y, b.size.w, b.size.h] for b in gt.boxes], np.float64).reshape(-1, 4)
        cls = np.array([b.labelid for b in gt.boxes], np.int32)
        vec = encode_labels_batch(self.preset, self.num_classes, [boxes], [cls])[0]
        return data, vec, gt


class ResizeTransform(Transform):
    """Parameters: width, height, algorithms (transforms.py:117-126)"""
    def __call__(self, data, label, gt):
        alg = random.choice(self.algorithms)
        data._geometric_ok('a second ResizeTransform')
        data = _copy_plan(data)
        data.resize = (int(self.width), int(self.height), int(alg))
        data.is_float_at_resize = data.is_float
        return data, label, gt


class RandomTransform(Transform):
    """Parameters: prob, transform (transforms.py:129-139)"""
    def __call__(self, data, label, gt):
        p = random.uniform(0, 1)
        if p < self.prob:
            return self.transform(data, label, gt)
        return data, label, gt


class ComposeTransform(Transform):
    """Parameters: transforms (transforms.py:142-151)"""
    def __call__(self, data, label, gt):
        args = (data, label, gt)
        for t in self.transforms:
            args = t(*args)
        return args


class TransformPickerTransform(Transform):
    """Parameters: transforms (transforms.py:154-161)"""
    def __call__(self, data, label, gt):
        pick = random.randint(0, len(self.transforms) - 1)
        return self.transforms[pick](data, label, gt)


class BrightnessTransform(Transform):
    """Parameters: delta (transforms.py:164-176)"""
    def __call__(self, data, label, gt):
        delta = random.randint(-self.delta, self.delta)
        data = _copy_plan(data)
        if data._pointwise(3, int(delta), 'BrightnessTransform'):
            if data.brightness is not None or data.distort or data.reorder != [0, 1, 2] or data.extra:
                data._extra_step(3, int(delta), 'BrightnessTransform')      # the canonical slot is taken or behind us
            else:
                data.brightness = int(delta)
        return data, label, gt


def _distort(data, kind, value, what):
    data = _copy_plan(data)
    if data._pointwise(kind, float(value), what):
        if data.reorder != [0, 1, 2] or data.extra or len(data.distort) >= 3:
            data._extra_step(kind, float(value), what)
        else:
            data.distort.append((kind, float(value)))
    return data


class ContrastTransform(Transform):
    """Parameters: lower, upper (transforms.py:179-191)"""
    def __call__(self, data, label, gt):
        return _distort(data, 0, random.uniform(self.lower, self.upper), 'ContrastTransform'), label, gt


class HueTransform(Transform):
    """Parameters: delta (transforms.py:194-208; shifts image ROW 0 of the HSV image, as the reference does)"""
    def __call__(self, data, label, gt):
        return _distort(data, 2, random.randint(-self.delta, self.delta), 'HueTransform'), label, gt


class SaturationTransform(Transform):
    """Parameters: lower, upper (transforms.py:211-225; scales image ROW 1 of the HSV image, as the reference does)"""
    def __call__(self, data, label, gt):
        return _distort(data, 1, random.uniform(self.lower, self.upper), 'SaturationTransform'), label, gt


class ReorderChannelsTransform(Transform):
    """transforms.py:228-234"""
    def __call__(self, data, label, gt):
        channels = [0, 1, 2]
        random.shuffle(channels)
        data = _copy_plan(data)
        if data._pointwise(4, list(channels), 'ReorderChannelsTransform'):
            if data.extra:
                data._extra_step(4, list(channels), 'ReorderChannelsTransform')
            else:
                data.reorder = [data.reorder[c] for c in channels]
        return data, label, gt


def transform_box(box, orig_size, new_size, h_off, w_off):
    """A ground-truth box after the image was shifted by (w_off, h_off) pixels into a canvas of new_size; None when
    the box's centre pixel leaves the canvas (transforms.py:236-259: integer pixel box, centre = corner + int(extent / 2))."""
    x0, x1, y0, y1 = prop2abs(box.center, box.size, orig_size)
    x0, x1, y0, y1 = x0 + w_off, x1 + w_off, y0 + h_off, y1 + h_off
    centre_x = x0 + int((x1 - x0) / 2)
    centre_y = y0 + int((y1 - y0) / 2)
    if not (0 <= centre_x < new_size.w and 0 <= centre_y < new_size.h):
        return None
    return Box(box.label, box.labelid, *abs2prop(x0, x1, y0, y1, new_size))


def transform_gt(gt, new_size, h_off, w_off):
    """The sample record on the new canvas, boxes whose centre left it dropped (transforms.py:262-270)."""
    moved = (transform_box(b, gt.imgsize, new_size, h_off, w_off) for b in gt.boxes)
    return Sample(gt.filename, [b for b in moved if b is not None], new_size)


class ExpandTransform(Transform):
    """Parameters: max_ratio, mean_value (transforms.py:273-301).  The kernel fills with 104, 117, 123."""
    def __call__(self, data, label, gt):
        ratio = random.uniform(1, self.max_ratio)
        orig_size = gt.imgsize
        new_size = Size(int(orig_size.w * ratio), int(orig_size.h * ratio))
        h_off = random.randint(0, new_size.h - orig_size.h)
        w_off = random.randint(0, new_size.w - orig_size.w)
        data._geometric_ok('ExpandTransform')
        data = _copy_plan(data)
        mean = [float(v) for v in getattr(self, 'mean_value', [104, 117, 123])]
        if (orig_size.w, orig_size.h) != tuple(data.size):
            raise ValueError('gt.imgsize %s does not match the image at this point of the chain %s' % (orig_size, data.size))
        if data.fill_from is not None:      # a canvas exists already
            if len(data.extra) > data.fill_from:
                raise NotImplementedError('a second expand behind photometric steps that followed the first: two canvases with different histories')
            if mean != [float(v) for v in data.mean]:
                raise NotImplementedError('two expands with different mean values')
        else:
            data.fill_from = len(data.extra)
        data.mean = mean
        # the plan expands BEFORE it crops and flips: an expand behind a flip places the image at the mirrored column offset; behind
        # another expand it adds its offsets (the same fill value surrounds both canvases); behind a crop the offsets are taken
        # relative to the crop's origin and only the cropped window of the source stays visible
        w_plan = new_size.w - orig_size.w - w_off if data.flip else w_off
        eh, ew = (data.expand[1], data.expand[2]) if data.expand is not None else (0, 0)
        cx0, cy0 = (data.crop[0], data.crop[1]) if data.crop is not None else (0, 0)
        if data.crop is not None:
            win = (cx0 - ew, cy0 - eh, cx0 - ew + data.crop[2], cy0 - eh + data.crop[3])       # the crop window in source coordinates
            old = data.clip if data.clip is not None else (0, 0, data.src.w, data.src.h)
            x0, y0, x1, y1 = max(old[0], win[0]), max(old[1], win[1]), min(old[2], win[2]), min(old[3], win[3])
            data.clip = (x0, y0, max(x0, x1), max(y0, y1))
            data.crop = None
        data.expand = (new_size, eh - cy0 + int(h_off), ew - cx0 + int(w_plan))
        data.is_float = True           # np.zeros(...) canvas: float64 (transforms.py:290)
        gt = transform_gt(gt, new_size, h_off, w_off)
        return data, label, gt


def _jaccard_plus1(box_arr, others):
    """ssdutils.py:139-149 on the host: a handful of boxes per image"""
    areaa = (others[:, 1] - others[:, 0] + 1) * (others[:, 3] - others[:, 2] + 1)
    areab = (box_arr[1] - box_arr[0] + 1) * (box_arr[3] - box_arr[2] + 1)
    xxmin = np.maximum(box_arr[0], others[:, 0]); xxmax = np.minimum(box_arr[1], others[:, 1])
    yymin = np.maximum(box_arr[2], others[:, 2]); yymax = np.minimum(box_arr[3], others[:, 3])
    w = np.maximum(0, xxmax - xxmin + 1); h = np.maximum(0, yymax - yymin + 1)
    inter = w * h
    return inter / (areab + areaa - inter)


class SamplerTransform(Transform):
    """Params: sample, min_scale, max_scale, min_aspect_ratio, max_aspect_ratio, min_jaccard_overlap, max_trials
    (transforms.py:304-359).  Returns None when no window satisfies the overlap constraint.

    A trial consumes four uniform draws in the reference's order (scale, aspect ratio, then the window's x and y
    slack); the window counts when its best IoU (+1 pixel convention, ssdutils.py:139-149) with a ground-truth box is
    positive and at least min_jaccard_overlap."""

    def _draw_window(self, imgsize):
        scale = random.uniform(self.min_scale, self.max_scale)
        ratio = random.uniform(self.min_aspect_ratio, self.max_aspect_ratio)
        ratio = min(max(ratio, scale ** 2), 1 / (scale ** 2))          # keeps both sides of the window inside the unit square
        width, height = scale * sqrt(ratio), scale / sqrt(ratio)
        cx = 0.5 * width + random.uniform(0, 1 - width)
        cy = 0.5 * height + random.uniform(0, 1 - height)
        return np.array(prop2abs(Point(cx, cy), Size(width, height), imgsize))

    def __call__(self, data, label, gt):
        if not self.sample:
            return data, label, gt
        gt_px = np.array([prop2abs(b.center, b.size, gt.imgsize) for b in gt.boxes], dtype=np.float64).reshape(-1, 4)
        for _ in range(self.max_trials):
            window = self._draw_window(gt.imgsize)
            best = _jaccard_plus1(window, gt_px).max()
            if best > 0 and best >= self.min_jaccard_overlap:
                break
        else:
            return None
        return self.crop(data, label, gt, window)

    def crop(self, data, label, gt, window):
        """The accepted window (xmin, xmax, ymin, ymax in pixels of gt.imgsize) applied to the plan and the boxes."""
        left, top = int(window[0]), int(window[2])
        new_size = Size(int(window[1] - window[0]), int(window[3] - window[2]))
        out = _copy_plan(data)
        out._geometric_ok('SamplerTransform')
        # the plan crops BEFORE it flips: a window of the flipped frame is the mirrored window of the unflipped one
        left_plan = out.size.w - (left + new_size.w) if out.flip else left
        x0, y0 = (out.crop[0], out.crop[1]) if out.crop is not None else (0, 0)
        out.crop = (x0 + left_plan, y0 + top, new_size.w, new_size.h)
        return out, label, transform_gt(gt, new_size, -top, -left)


def _copy_plan(p):
    q = ImagePlan.__new__(ImagePlan)
    q.__dict__.update(p.__dict__)
    q.distort = list(p.distort)
    q.reorder = list(p.reorder)
    q.extra = list(p.extra)
    q.extra_r0 = list(p.extra_r0)
    q.post = list(p.post)
    q.mean = list(p.mean)
    return q


NATIVE_SAMPLER = os.environ.get('SSD_NATIVE_SAMPLER', '1') != '0'      # A/B switch: '0' runs the trial loops in Python


class SamplePickerTransform(Transform):
    """Parameters: samplers (transforms.py:362-376).

    The samplers' trial loops (<= 50 draws of a window each, ~210 per image: the whole cost of planning an image in
    Python) run in ONE native call, ssd_sampler_trials, on the state of Python's `random`: the same draws in the same
    order, the generator left where the Python loops leave it (tests/test_augment.py pins that against the loops)."""

    def __call__(self, data, label, gt):
        if NATIVE_SAMPLER and all(type(s) is SamplerTransform for s in self.samplers):
            return self._native(data, label, gt)
        samples = []
        for sampler in self.samplers:
            sample = sampler(data, label, gt)
            if sample is not None:
                samples.append(sample)
        return random.choice(samples)

    def _native(self, data, label, gt):
        active = [s for s in self.samplers if s.sample]
        n = len(active)
        found = (C.c_int * max(n, 1))()
        windows = (C.c_longlong * (4 * max(n, 1)))()
        if n:
            if not gt.boxes:
                raise ValueError('zero-size array to reduction operation maximum which has no identity')      # what the loop's .max() raises
            st = random.getstate()
            mt = (C.c_uint32 * 625)(*st[1])
            params = (C.c_double * (5 * n))(*[v for s in active for v in (s.min_scale, s.max_scale, s.min_aspect_ratio, s.max_aspect_ratio,
                                                                         s.min_jaccard_overlap)])
            trials = (C.c_int * n)(*[int(s.max_trials) for s in active])
            gt_px = np.array([prop2abs(b.center, b.size, gt.imgsize) for b in gt.boxes], dtype=np.float64).reshape(-1, 4)
            check(lib.ssd_sampler_trials(mt, n, params, trials, int(gt.imgsize.w), int(gt.imgsize.h), gt_px.ctypes.data, len(gt.boxes),
                                         windows, found))
            random.setstate((st[0], tuple(mt), st[2]))
        # the candidates in the samplers' order; only the chosen one is materialised (building a candidate draws nothing)
        cands, k = [], 0
        for s in self.samplers:
            if not s.sample:
                cands.append((s, None))
            else:
                if found[k]:
                    cands.append((s, [windows[4 * k + i] for i in range(4)]))
                k += 1
        s, window = random.choice(cands)
        return (data, label, gt) if window is None else s.crop(data, label, gt, window)


class HorizontalFlipTransform(Transform):
    """transforms.py:379-392"""
    def __call__(self, data, label, gt):
        data = _copy_plan(data)
        if data.resize is not None:
            data.out_flip = not data.out_flip      # behind the resize: the output's columns
        else:
            data.flip = not data.flip
        boxes = []
        for box in gt.boxes:
            center = Point(1 - box.center.x, box.center.y)
            boxes.append(Box(box.label, box.labelid, center, box.size))
        return data, label, Sample(gt.filename, boxes, gt.imgsize)


# ------------------------------------------------------------------------------------------------
# the batch kernel
# ------------------------------------------------------------------------------------------------
class _Params(C.Structure):
    """ssd_augment_params (include/ssdvgg_hip.h)"""
    _fields_ = [('src_off', C.c_ulonglong), ('src_w', C.c_int), ('src_h', C.c_int),
                ('brightness_on', C.c_int), ('brightness_delta', C.c_int),
                ('n_distort', C.c_int), ('distort_kind', C.c_int * 3), ('distort_val', C.c_float * 3),
                ('reorder', C.c_int * 3),
                ('expand_on', C.c_int), ('exp_w', C.c_int), ('exp_h', C.c_int), ('exp_hoff', C.c_int), ('exp_woff', C.c_int),
                ('crop_x0', C.c_int), ('crop_y0', C.c_int), ('crop_w', C.c_int), ('crop_h', C.c_int),
                ('flip', C.c_int), ('resize_alg', C.c_int),
                ('n_extra', C.c_int), ('extra_kind', C.c_int * MAX_EXTRA), ('extra_val', C.c_float * MAX_EXTRA),
                ('extra_r0', C.c_int * MAX_EXTRA), ('fill_from', C.c_int), ('is_float', C.c_int), ('mean', C.c_double * 3),
                ('clip_x0', C.c_int), ('clip_y0', C.c_int), ('clip_x1', C.c_int), ('clip_y1', C.c_int),
                ('n_post', C.c_int), ('post_kind', C.c_int * MAX_POST), ('post_val', C.c_float * MAX_POST), ('out_flip', C.c_int)]


def _step_val(kind, val):
    return float(val[0] + 4 * val[1] + 16 * val[2]) if kind == 4 else float(val)      # a permutation travels as a base-4 code


def plan_params(plans, width, height):
    """(ctypes array of ssd_augment_params, packed uint8 image bytes) for a list of ImagePlans"""
    arr = (_Params * len(plans))()
    offs, off = [], 0
    for p in plans:
        offs.append(off)
        off += (p.image.size + 15) // 16 * 16
    packed = np.empty(off, np.uint8)
    for i, p in enumerate(plans):
        if p.resize is None:
            raise ValueError('plan %d was not resized: the batch needs one output size (ResizeTransform last)' % i)
        if (p.resize[0], p.resize[1]) != (width, height):
            raise ValueError('plan %d resizes to %s, the batch is %s' % (i, p.resize[:2], (width, height)))
        q = arr[i]
        q.src_off = offs[i]; q.src_w = p.src.w; q.src_h = p.src.h
        q.brightness_on = int(p.brightness is not None); q.brightness_delta = p.brightness or 0
        q.n_distort = len(p.distort)
        for k, (kind, val) in enumerate(p.distort):
            q.distort_kind[k] = kind; q.distort_val[k] = val
        for c in range(3):
            q.reorder[c] = p.reorder[c]
        if p.expand is not None:
            q.expand_on = 1; q.exp_w = p.expand[0].w; q.exp_h = p.expand[0].h; q.exp_hoff = p.expand[1]; q.exp_woff = p.expand[2]
        frame = p.expand[0] if p.expand is not None else p.src
        x0, y0, cw, ch = p.crop if p.crop is not None else (0, 0, frame.w, frame.h)
        q.crop_x0 = x0; q.crop_y0 = y0; q.crop_w = cw; q.crop_h = ch
        q.flip = int(p.flip); q.resize_alg = p.resize[2]
        q.n_extra = len(p.extra)
        for k, (kind, val) in enumerate(p.extra):
            q.extra_kind[k] = kind
            q.extra_val[k] = _step_val(kind, val)
            q.extra_r0[k] = p.extra_r0[k]
        q.fill_from = len(p.extra) if p.fill_from is None else p.fill_from
        q.is_float = int(p.expand is not None if p.is_float_at_resize is None else p.is_float_at_resize)
        for c in range(3):
            q.mean[c] = float(p.mean[c])
        q.clip_x0, q.clip_y0, q.clip_x1, q.clip_y1 = p.clip if p.clip is not None else (0, 0, p.src.w, p.src.h)
        q.n_post = len(p.post)
        for k, (kind, val) in enumerate(p.post):
            q.post_kind[k] = kind
            q.post_val[k] = _step_val(kind, val)
        q.out_flip = int(p.out_flip)
        n = p.image.size
        packed[offs[i]:offs[i] + n] = p.image.reshape(-1)
        packed[offs[i] + n:offs[i] + (n + 15) // 16 * 16] = 0
    return arr, packed


def augment_batch(plans, width, height, device=0, out=None):
    """Run a batch of ImagePlans on the GPU, on torch's current stream.  Returns a torch float32 tensor
    [b, height, width, 3] on `device` (what training_data.py:100-104 stacks on the host)."""
    import torch
    arr, packed = plan_params(plans, width, height)
    dev = torch.device('cuda', device)
    images = torch.from_numpy(packed).to(dev)
    b = len(plans)
    if out is None:
        out = torch.empty((b, height, width, 3), dtype=torch.float32, device=dev)
    ws = torch.empty((lib.ssd_augment_ws_bytes(b, width, height),), dtype=torch.uint8, device=dev)
    check(lib.ssd_augment_batch_dev(images.data_ptr(), C.cast(arr, C.c_void_p), b, width, height, out.data_ptr(), ws.data_ptr(),
                                    torch.cuda.current_stream(dev).cuda_stream))
    # the scratch tensors are released in stream order by torch's allocator; the parameter structs were copied by the call
    return out


# ------------------------------------------------------------------------------------------------
# the recipes of process_dataset.py:60-147
# ------------------------------------------------------------------------------------------------
def build_sampler(overlap, trials):
    """process_dataset.py:60-63"""
    return SamplerTransform(sample=True, min_scale=0.3, max_scale=1.0, min_aspect_ratio=0.5, max_aspect_ratio=2.0,
                            min_jaccard_overlap=overlap, max_trials=trials)


def build_train_transforms(preset, num_classes, sampler_trials, expand_prob, images=None):
    """process_dataset.py:66-140"""
    tf_resize = ResizeTransform(width=preset.image_size.w, height=preset.image_size.h,
                                algorithms=[INTER_LINEAR, INTER_AREA, INTER_NEAREST, INTER_CUBIC, INTER_LANCZOS4])
    tf_rnd_brightness = RandomTransform(prob=0.5, transform=BrightnessTransform(delta=32))
    tf_rnd_contrast = RandomTransform(prob=0.5, transform=ContrastTransform(lower=0.5, upper=1.5))
    tf_rnd_hue = RandomTransform(prob=0.5, transform=HueTransform(delta=18))
    tf_rnd_saturation = RandomTransform(prob=0.5, transform=SaturationTransform(lower=0.5, upper=1.5))
    tf_rnd_reorder_channels = RandomTransform(prob=0.5, transform=ReorderChannelsTransform())
    tf_distort_lst = [tf_rnd_contrast, tf_rnd_saturation, tf_rnd_hue, tf_rnd_contrast]
    tf_distort = TransformPickerTransform(transforms=[ComposeTransform(transforms=tf_distort_lst[:-1]),
                                                      ComposeTransform(transforms=tf_distort_lst[1:])])
    tf_rnd_expand = RandomTransform(prob=expand_prob, transform=ExpandTransform(max_ratio=4.0, mean_value=[104, 117, 123]))
    samplers = [SamplerTransform(sample=False)] + [build_sampler(o, sampler_trials) for o in (0.1, 0.3, 0.5, 0.7, 0.9, 1.0)]
    tf_sample_picker = SamplePickerTransform(samplers=samplers)
    tf_rnd_flip = RandomTransform(prob=0.5, transform=HorizontalFlipTransform())
    return [ImageLoaderTransform(images=images), tf_rnd_brightness, tf_distort, tf_rnd_reorder_channels, tf_rnd_expand,
            tf_sample_picker, tf_rnd_flip, LabelCreatorTransform(preset=preset, num_classes=num_classes), tf_resize]


def build_valid_transforms(preset, num_classes, images=None):
    """process_dataset.py:143-153"""
    return [ImageLoaderTransform(images=images), LabelCreatorTransform(preset=preset, num_classes=num_classes),
            ResizeTransform(width=preset.image_size.w, height=preset.image_size.h, algorithms=[INTER_LINEAR])]
