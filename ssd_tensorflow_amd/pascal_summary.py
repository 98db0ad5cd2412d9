"""VOC "comp4" detection files, the counterpart of the reference's PascalSummary (pascal_summary.py:30-65).

One text file per class, `comp4_det_test_<class>.txt`, one line per detection:
`<image id> <confidence> <left> <top> <right> <bottom>`, six decimals, pixel corners 1-based.
The corners are the reference's arithmetic: prop2abs (truncation toward zero) on the image's own size,
clamped into the image, plus one.  The reference learns the image size by decoding the file with
cv2.imread (pascal_summary.py:41-42); here the caller passes it (it has just loaded the image, or holds
the dataset record) -- nothing is decoded twice.  Unpinned against the reference (it cannot run without
cv2); tests/test_pascal.py checks the arithmetic by hand."""
import os

from .utils import prop2abs, Size


def _image_id(filename):
    """basename without its LAST extension; further dots disappear like in the reference ('a.b.jpg' -> 'ab')"""
    parts = os.path.basename(filename).split('.')
    return ''.join(parts[:-1])


def _corners(box, img_size):
    lo_hi = prop2abs(box.center, box.size, img_size)
    limits = (img_size.w - 1, img_size.w - 1, img_size.h - 1, img_size.h - 1)
    xmin, xmax, ymin, ymax = (min(max(v, 0), lim) for v, lim in zip(lo_hi, limits))
    return xmin + 1.0, ymin + 1.0, xmax + 1.0, ymax + 1.0      # left, top, right, bottom


class PascalSummary:
    def __init__(self):
        self.rows = {}              # class name -> [(image id, confidence, left, top, right, bottom)], insertion ordered

    def add_detections(self, filename, boxes, img_size=None):
        """boxes: [(confidence, Box)] as decode_boxes / suppress_overlaps return them; img_size: Size(w, h) of the file"""
        if img_size is None:
            raise ValueError('img_size is required: pass the size of the image the detections belong to')
        img_size = Size(*img_size)
        image_id = _image_id(filename)
        for confidence, box in boxes:
            self.rows.setdefault(box.label, []).append((image_id, confidence) + _corners(box, img_size))

    def write_summary(self, target_dir):
        for label, rows in self.rows.items():
            with open(os.path.join(target_dir, 'comp4_det_test_%s.txt' % label), 'w') as f:
                f.writelines('%s %.6f %.6f %.6f %.6f %.6f\n' % row for row in rows)
