"""PascalVOCSource of the reference's source_pascal_voc.py (:61-210): VOC annotation XML -> Sample records.

Differences forced by the environment: xml.etree instead of lxml (same documents, same fields), and the
image size comes from the annotation's <size> element instead of decoding the JPEG with cv2.imread
(source_pascal_voc.py:108-109) -- VOC annotations carry it and it is what the JPEG holds.  Images themselves
are not decoded here (no OpenCV): a Sample's filename is what ImageLoaderTransform is asked for."""
import os
import xml.etree.ElementTree as ET
from glob import glob

from .utils import Label, Box, Sample, Size, abs2prop

VOC_NAMES = ['aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable', 'dog',
             'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor']
_RGB = [(0, 0, 0), (111, 74, 0), (81, 0, 81), (128, 64, 128), (244, 35, 232), (230, 150, 140), (70, 70, 70), (102, 102, 156),
        (190, 153, 153), (150, 120, 90), (153, 153, 153), (250, 170, 30), (220, 220, 0), (107, 142, 35), (52, 151, 52),
        (70, 130, 180), (220, 20, 60), (0, 0, 142), (0, 0, 230), (119, 11, 32)]
label_defs = [Label(n, (c[2], c[1], c[0])) for n, c in zip(VOC_NAMES, _RGB)]       # rgb2bgr (utils.py:58-62)


class PascalVOCSource:
    def __init__(self):
        self.num_classes = len(label_defs)
        self.colors = {l.name: l.color for l in label_defs}
        self.lid2name = {i: l.name for i, l in enumerate(label_defs)}
        self.lname2id = {l.name: i for i, l in enumerate(label_defs)}
        self.num_train = self.num_valid = self.num_test = 0
        self.train_samples, self.valid_samples, self.test_samples = [], [], []

    @staticmethod
    def _annotation_list(root, dataset_type):
        """source_pascal_voc.py:76-88"""
        out = []
        with open(root + '/ImageSets/Main/' + dataset_type + '.txt') as f:
            for line in f:
                fn = root + '/Annotations/' + line.strip() + '.xml'
                if os.path.exists(fn):
                    out.append(fn)
        return out

    def _sample_list(self, root, annot_files, require_image=True):
        """source_pascal_voc.py:91-137"""
        samples = []
        for fn in annot_files:
            doc = ET.parse(fn).getroot()
            filename = root + '/JPEGImages/' + doc.findtext('filename')
            if require_image and not os.path.exists(filename):
                continue
            imgsize = Size(int(doc.findtext('size/width')), int(doc.findtext('size/height')))
            boxes = []
            for obj in doc.findall('object'):
                label = obj.findtext('name')
                xmin = int(float(obj.findtext('bndbox/xmin'))); xmax = int(float(obj.findtext('bndbox/xmax')))
                ymin = int(float(obj.findtext('bndbox/ymin'))); ymax = int(float(obj.findtext('bndbox/ymax')))
                center, size = abs2prop(xmin, xmax, ymin, ymax, imgsize)
                boxes.append(Box(label, self.lname2id[label], center, size))
            if boxes:
                samples.append(Sample(filename, boxes, imgsize))
        return samples

    def load_trainval_data(self, data_dir, valid_fraction, require_image=True):
        """source_pascal_voc.py:140-187: VOC2007 + VOC2012 trainval + VOC2007 test train; the VOC2012 annotations on no
        list validate"""
        train_annot, train = [], []
        for vocid in ('VOC2007', 'VOC2012'):
            root = data_dir + '/trainval/VOCdevkit/' + vocid
            annot = self._annotation_list(root, 'trainval')
            train_annot += annot
            train += self._sample_list(root, annot, require_image)
        root = data_dir + '/test/VOCdevkit/VOC2007'
        train += self._sample_list(root, self._annotation_list(root, 'test'), require_image)
        root = data_dir + '/trainval/VOCdevkit/VOC2012'
        valid = self._sample_list(root, sorted(set(glob(root + '/Annotations/*.xml')) - set(train_annot)), require_image)
        self.valid_samples, self.train_samples = valid, train
        if not train:
            raise RuntimeError('No training samples found in ' + data_dir)
        if valid_fraction > 0 and not valid:
            raise RuntimeError('No validation samples found in ' + data_dir)
        self.num_train, self.num_valid = len(train), len(valid)

    def load_test_data(self, data_dir, require_image=True):
        """source_pascal_voc.py:190-203"""
        root = data_dir + '/test/VOCdevkit/VOC2012'
        self.test_samples = self._sample_list(root, self._annotation_list(root, 'test'), require_image)
        if not self.test_samples:
            raise RuntimeError('No testing samples found in ' + data_dir)
        self.num_test = len(self.test_samples)


def get_source():
    """source_pascal_voc.py:207-209"""
    return PascalVOCSource()
