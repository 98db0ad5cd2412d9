"""Test helper (round 5): freely composed transform lists.

The reference's transforms are independent callables on pixels (transforms.py:117-391) that a user may compose in any order.
The product's mirror (ssd_tensorflow_amd/transforms.py) touches no pixel: it rewrites the list into ONE plan of the batch
kernel's canonical form.  Here a list runs twice from the same `random` seed: through the mirror (-> a plan) and through
"pixel twins" that draw the SAME numbers in the same order and apply the oracle's pixel operations in the order given.
`run_plan` executes a plan with the oracle's pixel operations in the plan's own order (brightness -> distort -> reorder ->
extra steps -> expand -> crop -> flip -> resize), so the CPU test checks the rewriting alone; the GPU test compares the batch
kernel with the free composition directly."""
import random

import numpy as np

from oracle import augment as oa
from ssd_tensorflow_amd import transforms as T
from ssd_tensorflow_amd.utils import Sample, Box, Point, Size

KINDS = {0: oa.contrast, 1: oa.saturation, 2: oa.hue, 3: oa.brightness}


def sample(size, boxes=((0.5, 0.5, 0.4, 0.4),), cls=(1,)):
    return Sample('im', [Box('c%d' % c, int(c), Point(b[0], b[1]), Size(b[2], b[3])) for b, c in zip(boxes, cls)], Size(*size))


# ---- a step = (name, params); the mirror's transform and the pixel twin draw alike
def mirror_step(step, args):
    name, kw = step
    if name == 'brightness':
        return T.BrightnessTransform(delta=kw.get('delta', 32))(*args)
    if name == 'contrast':
        return T.ContrastTransform(lower=0.5, upper=1.5)(*args)
    if name == 'saturation':
        return T.SaturationTransform(lower=0.5, upper=1.5)(*args)
    if name == 'hue':
        return T.HueTransform(delta=18)(*args)
    if name == 'reorder':
        return T.ReorderChannelsTransform()(*args)
    if name == 'expand':
        return T.ExpandTransform(max_ratio=kw.get('max_ratio', 2.0), mean_value=kw.get('mean_value', [104, 117, 123]))(*args)
    if name == 'crop':       # an accepted sampler window, as fractions of the current frame
        data, label, gt = args
        w, h = gt.imgsize.w, gt.imgsize.h
        fx0, fx1, fy0, fy1 = kw['window']
        win = np.array([int(fx0 * w), int(fx1 * w), int(fy0 * h), int(fy1 * h)])
        return T.SamplerTransform(sample=True).crop(data, label, gt, win)
    if name == 'flip':
        return T.HorizontalFlipTransform()(*args)
    if name == 'resize':
        return T.ResizeTransform(width=kw['size'][0], height=kw['size'][1], algorithms=kw.get('algorithms', [T.INTER_LINEAR]))(*args)
    raise KeyError(name)


def pixel_step(step, img):
    """the reference's transform on pixels (oracle primitives), drawing what mirror_step's transform draws"""
    name, kw = step
    if name == 'brightness':
        d = kw.get('delta', 32)
        return oa.brightness(img, random.randint(-d, d))
    if name == 'contrast':
        return oa.contrast(img, random.uniform(0.5, 1.5))
    if name == 'saturation':
        return oa.saturation(img, random.uniform(0.5, 1.5))
    if name == 'hue':
        return oa.hue(img, random.randint(-18, 18))
    if name == 'reorder':
        ch = [0, 1, 2]
        random.shuffle(ch)
        return img[:, :, ch]
    if name == 'expand':
        ratio = random.uniform(1, kw.get('max_ratio', 2.0))
        h, w = img.shape[:2]
        nw, nh = int(w * ratio), int(h * ratio)
        h_off = random.randint(0, nh - h)
        w_off = random.randint(0, nw - w)
        return oa.expand(img, (nw, nh), h_off, w_off, kw.get('mean_value'))
    if name == 'crop':
        h, w = img.shape[:2]
        fx0, fx1, fy0, fy1 = kw['window']
        x0, x1, y0, y1 = int(fx0 * w), int(fx1 * w), int(fy0 * h), int(fy1 * h)
        return img[y0:y1, x0:x1]
    if name == 'flip':
        return img[:, ::-1]
    if name == 'resize':
        alg = random.choice(kw.get('algorithms', [T.INTER_LINEAR]))
        return oa.resize(np.ascontiguousarray(img), kw['size'][0], kw['size'][1], alg)
    raise KeyError(name)


def compose_mirror(steps, img, seed):
    random.seed(seed)
    args = (T.ImagePlan(img), None, sample((img.shape[1], img.shape[0])))
    for st in steps:
        args = mirror_step(st, args)
    return args[0], args[2]


def compose_pixels(steps, img, seed):
    random.seed(seed)
    d = img
    for st in steps:
        d = pixel_step(st, d)
    return np.asarray(d, np.float32)


def _step_at(kind, v, d, r0):
    """one pointwise step on array d whose Hue / Saturation rows 0 / 1 are rows r0 / r0 + 1 of d (oracle primitives)"""
    if kind == 4:
        return d[:, :, list(v)]
    if kind in (0, 3) or r0 == 0:
        return KINDS[kind](d, v)
    # Hue / Saturation: the HSV round trip on every pixel, the shift / scale on one row (oa.hue / oa.saturation with the row moved)
    h = oa.bgr2hsv_u8(d).astype(np.float32)
    r = r0 + (0 if kind == 2 else 1)
    if 0 <= r < h.shape[0]:
        if kind == 2:
            h[r] += v
            h[r][h[r] > 180] -= 180
            h[r][h[r] < 0] += 180
        else:
            h[r] *= v
            h[r][h[r] > 255] = 255
            h[r][h[r] < 0] = 0
    return oa.hsv2bgr_u8(h.astype(np.uint8))


def run_plan(p):
    """A plan executed with the oracle's pixel operations in the plan's canonical order: the pointwise chain on the source image
    (and, from the first expand's steps on, on one canvas ROW PROFILE of the mean value) -> canvas + visible window -> crop -> flip
    -> resize -> steps behind the resize -> output flip."""
    d = p.image
    if p.brightness is not None:
        d = oa.brightness(d, p.brightness)
    for kind, v in p.distort:
        d = KINDS[kind](d, v)
    d = d[:, :, p.reorder]
    for (kind, v), r0 in zip(p.extra, p.extra_r0):
        d = _step_at(kind, v, d, r0)
    if p.expand is not None:
        new, h_off, w_off = p.expand
        H, W = new.h, new.w
        # the canvas: the mean value, transformed row by row by the steps taken behind the expand (canvas row y = source row y - h_off)
        fill = np.zeros((H, 1, 3))
        fill[:, :] = np.array(p.mean, np.float64)
        steps = list(zip(p.extra, p.extra_r0))[p.fill_from:] if p.fill_from is not None else []
        if steps:
            f = fill.astype(np.float32)
            for (kind, v), r0 in steps:
                f = _step_at(kind, v, f.astype(np.uint8) if f.dtype != np.uint8 and kind in (1, 2) else f, r0 + h_off)
            fill = f
        is_u8 = steps and fill.dtype == np.uint8
        canvas = np.repeat(np.asarray(fill, np.float64), W, axis=1)
        x0, y0, x1, y1 = p.clip if p.clip is not None else (0, 0, p.src.w, p.src.h)
        # the visible window of the source, clipped to the canvas
        cx0, cy0 = max(x0 + w_off, 0), max(y0 + h_off, 0)
        cx1, cy1 = min(x1 + w_off, W), min(y1 + h_off, H)
        if cx1 > cx0 and cy1 > cy0:
            canvas[cy0:cy1, cx0:cx1] = d[cy0 - h_off:cy1 - h_off, cx0 - w_off:cx1 - w_off]
        d = canvas if p.is_float_at_resize else canvas.astype(np.uint8)
        assert p.is_float_at_resize or is_u8 or d.dtype == np.uint8
    if p.crop is not None:
        x0, y0, w, h = p.crop
        d = d[y0:y0 + h, x0:x0 + w]
    if p.flip:
        d = d[:, ::-1]
    d = oa.resize(np.ascontiguousarray(d), p.resize[0], p.resize[1], p.resize[2])
    for kind, v in p.post:
        d = _step_at(kind, v, d, 0)
    if p.out_flip:
        d = d[:, ::-1]
    return np.asarray(d, np.float32)


RS = ('resize', dict(size=(96, 80), algorithms=[T.INTER_LINEAR, T.INTER_AREA, T.INTER_NEAREST, T.INTER_CUBIC, T.INTER_LANCZOS4]))
WIN = dict(window=(0.15, 0.85, 0.2, 0.9))
# lists OUTSIDE the recipe order that are compositions of the plan's form
FREE_LISTS = [
    [('flip', {}), ('crop', WIN), RS],                                                           # crop after flip
    [('flip', {}), ('expand', {}), ('crop', WIN), ('flip', {}), RS],                             # expand after flip, flip twice
    [('expand', {}), ('expand', dict(max_ratio=1.5)), ('crop', WIN), RS],                        # two expands
    [('contrast', {}), ('brightness', {}), ('reorder', {}), ('hue', {}), ('brightness', dict(delta=20)), RS],   # brightness behind the chain, steps behind a reorder
    [('brightness', {}), ('contrast', {}), ('saturation', {}), ('hue', {}), ('reorder', {}),
     ('brightness', {}), ('saturation', {}), ('hue', {}), ('contrast', {}), ('reorder', {}), ('expand', {}), ('crop', WIN), ('flip', {}), RS],   # the recipe's photometric pass twice
    [('flip', {}), ('hue', {}), ('saturation', {}), ('crop', WIN), ('flip', {}), ('crop', dict(window=(0.1, 0.7, 0.0, 1.0))), RS],   # photometric after a flip, crops either side of a flip
    [('reorder', {}), ('reorder', {}), ('contrast', {}), ('reorder', {}), RS],
    [('hue', {}), ('crop', WIN), ('brightness', {}), ('flip', {}), ('contrast', {}), ('reorder', {}), ('brightness', {}), RS],   # per-pixel steps after a crop
]
# round 6: orders that rounds 1-5 refused
ROUND6_LISTS = [
    [('expand', {}), ('brightness', {}), RS],                                                    # pointwise behind an expand: image AND canvas, uint8 again
    [('expand', {}), ('contrast', {}), ('hue', {}), ('saturation', {}), ('crop', WIN), ('flip', {}), RS],   # hue / saturation on the canvas's rows 0 / 1
    [('expand', {}), ('reorder', {}), RS],                                                       # the canvas's channels are permuted too (still float)
    [('crop', WIN), ('hue', {}), ('saturation', {}), RS],                                        # rows 0 / 1 of the CROPPED array
    [('crop', dict(window=(0.0, 1.0, 0.3, 0.9))), ('saturation', {}), ('crop', dict(window=(0.2, 0.8, 0.0, 0.5))), ('hue', {}), RS],
    [('crop', WIN), ('expand', {}), RS],                                                         # an expand behind a crop
    [('flip', {}), ('crop', WIN), ('expand', {}), ('crop', dict(window=(0.1, 0.9, 0.1, 0.95))), ('expand', dict(max_ratio=1.5)), ('flip', {}), RS],
    [('expand', dict(mean_value=[10.5, 200.25, 33.0])), ('crop', WIN), RS],                       # another mean value
    [RS, ('flip', {})],                                                                          # behind the resize
    [RS, ('contrast', {}), ('hue', {}), ('flip', {}), ('brightness', {})],
    [('expand', {}), RS, ('brightness', {}), ('saturation', {})],                                # a float64 resize result, made uint8 by the brightness
    [('hue', {}), ('crop', WIN), ('expand', {}), ('brightness', {}), ('hue', {}), ('crop', dict(window=(0.05, 0.95, 0.0, 0.9))), ('flip', {}), RS, ('reorder', {})],
]
# lists that are NOT plans: refused, loudly (the first two as the reference itself refuses them: cv2.cvtColor on float64)
REFUSED_LISTS = [
    [('expand', {}), ('hue', {})],
    [('expand', {}), RS, ('saturation', {})],
    [RS, ('crop', WIN)],
    [RS, ('expand', {})],
    [RS, RS],
    [('expand', {}), ('brightness', {}), ('expand', {})],
    [('expand', {}), ('expand', dict(mean_value=[1, 2, 3]))],
    [('contrast', {})] * 20,
    [RS] + [('contrast', {})] * 5,
]


def test_image(seed, size=(150, 110)):
    nrng = np.random.default_rng(seed)
    base = nrng.integers(0, 256, (size[1] // 8 + 2, size[0] // 8 + 2, 3)).astype(np.float32)
    img = np.kron(base, np.ones((8, 8, 1), np.float32))[:size[1], :size[0]] * 0.8 + nrng.integers(0, 52, (size[1], size[0], 3))
    return np.clip(img, 0, 255).astype(np.uint8)
