"""Geometry helpers and record types of the reference's utils.py (:64-135), host side.

These are the scalar conversions a caller uses to turn the GPU's integer boxes back into the
reference's Box(center, size) records; the batched arithmetic itself runs in the HIP library.
"""
import argparse
import math
from collections import namedtuple

Label = namedtuple('Label', ['name', 'color'])
Size = namedtuple('Size', ['w', 'h'])
Point = namedtuple('Point', ['x', 'y'])
Sample = namedtuple('Sample', ['filename', 'boxes', 'imgsize'])
Box = namedtuple('Box', ['label', 'labelid', 'center', 'size'])
Score = namedtuple('Score', ['idx', 'score'])
Overlap = namedtuple('Overlap', ['best', 'good'])


def str2bool(v):
    """utils.py:73-82"""
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


def abs2prop(xmin, xmax, ymin, ymax, imgsize):
    """Absolute min/max bounds -> proportional centre/size (utils.py:85-97)."""
    width = float(xmax - xmin)
    height = float(ymax - ymin)
    cx = float(xmin) + width / 2
    cy = float(ymin) + height / 2
    return Point(cx / imgsize.w, cy / imgsize.h), Size(width / imgsize.w, height / imgsize.h)


def prop2abs(center, size, imgsize):
    """Proportional centre/size -> absolute bounds; int() truncates toward zero (utils.py:100-108)."""
    width2 = size.w * imgsize.w / 2
    height2 = size.h * imgsize.h / 2
    cx = center.x * imgsize.w
    cy = center.y * imgsize.h
    return int(cx - width2), int(cx + width2), int(cy - height2), int(cy + height2)


def box_is_valid(box):
    """utils.py:111-115"""
    return not any(math.isnan(v) or math.isinf(v) for v in (box.center.x, box.center.y, box.size.w, box.size.h))


def normalize_box(box):
    """Clip to the 1000x1000 integer grid (utils.py:118-135)."""
    if not box_is_valid(box):
        return box
    img = Size(1000, 1000)
    xmin, xmax, ymin, ymax = prop2abs(box.center, box.size, img)
    xmin = max(xmin, 0); xmax = min(xmax, img.w - 1)
    ymin = max(ymin, 0); ymax = min(ymax, img.h - 1)
    xmin = min(xmin, xmax); ymin = min(ymin, ymax)
    center, size = abs2prop(xmin, xmax, ymin, ymax, img)
    return Box(box.label, box.labelid, center, size)


def load_data_source(data_source):
    """utils.py:44-55: the module `source_<name>` must provide get_source()"""
    import importlib
    try:
        module = importlib.import_module('ssd_tensorflow_amd.source_' + data_source)
    except ImportError:
        module = importlib.import_module('source_' + data_source)
    return module.get_source()
