"""PascalSummary of the reference's pascal_summary.py (:30-65): the VOC comp4 submission files, one per
class, `<image id> <confidence> <left> <top> <right> <bottom>` with +1-pixel corners and 6 decimals.

The reference opens every image with cv2.imread only to learn its size (pascal_summary.py:41-42); there
is no OpenCV here, so the size is passed in (the caller knows it: Sample.imgsize, or the array it fed).
Parity with the reference's output is by construction of the same arithmetic (prop2abs truncation, the
clamp to the image, the +1): nothing could be generated without cv2 (unpinned)."""
import os
from collections import defaultdict, namedtuple

from .utils import prop2abs, Size

Detection = namedtuple('Detection', ['fileid', 'confidence', 'left', 'top', 'right', 'bottom'])


class PascalSummary:
    def __init__(self):
        self.boxes = defaultdict(list)

    def add_detections(self, filename, boxes, img_size=None):
        """boxes: [(confidence, Box)] as decode_boxes / suppress_overlaps return them; img_size: Size(w, h) of the file"""
        if img_size is None:
            raise ValueError('img_size is required: this build cannot read the image file to learn its size (no OpenCV)')
        img_size = Size(*img_size)
        fileid = os.path.basename(filename)
        fileid = ''.join(fileid.split('.')[:-1])
        for conf, box in boxes:
            xmin, xmax, ymin, ymax = prop2abs(box.center, box.size, img_size)
            xmin = min(max(xmin, 0), img_size.w - 1); xmax = min(max(xmax, 0), img_size.w - 1)
            ymin = min(max(ymin, 0), img_size.h - 1); ymax = min(max(ymax, 0), img_size.h - 1)
            self.boxes[box.label].append(Detection(fileid, conf, float(xmin + 1), float(ymin + 1), float(xmax + 1), float(ymax + 1)))

    def write_summary(self, target_dir):
        for k, v in self.boxes.items():
            with open(target_dir + '/comp4_det_test_' + k + '.txt', 'w') as f:
                for det in v:
                    f.write('{} {:.6f} {:.6f} {:.6f} {:.6f} {:.6f}\n'.format(det.fileid, det.confidence, det.left, det.top, det.right, det.bottom))
