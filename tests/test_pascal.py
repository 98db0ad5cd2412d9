"""SURVEY.md 8f N4 (host-side dataset / reporting edges, no GPU): the VOC comp4 summary writer and the VOC XML reader.
Unpinned (both need cv2 in the reference); the expectations below are the reference's arithmetic done by hand:
pascal_summary.py:43-53 (prop2abs truncation, clamp into the image, +1) and source_pascal_voc.py:118-126."""
import os

from ssd_tensorflow_amd.pascal_summary import PascalSummary
from ssd_tensorflow_amd.source_pascal_voc import PascalVOCSource
from ssd_tensorflow_amd.utils import Box, Point, Size, load_data_source


def test_pascal_summary_format(tmp_path):
    ps = PascalSummary()
    boxes = [(0.875, Box('dog', 11, Point(0.5, 0.5), Size(0.5, 0.25))),          # 500x400: x 125..375, y 150..250
             (0.5, Box('dog', 11, Point(0.02, 0.98), Size(0.2, 0.2))),           # x -40..60 -> 0..60, y 352..432 -> 352..399
             (0.25, Box('cat', 7, Point(0.9, 0.1), Size(0.4, 0.1)))]            # x 350..550 -> 499, y 20..60
    ps.add_detections('/data/VOC/JPEGImages/2008_000123.jpg', boxes, img_size=(500, 400))
    ps.write_summary(str(tmp_path))
    dog = open(tmp_path / 'comp4_det_test_dog.txt').read().splitlines()
    cat = open(tmp_path / 'comp4_det_test_cat.txt').read().splitlines()
    assert dog == ['2008_000123 0.875000 126.000000 151.000000 376.000000 251.000000',
                   '2008_000123 0.500000 1.000000 353.000000 61.000000 400.000000']
    assert cat == ['2008_000123 0.250000 351.000000 21.000000 500.000000 61.000000']


VOC_XML = """<annotation><folder>VOC2012</folder><filename>{name}.jpg</filename>
<size><width>500</width><height>375</height><depth>3</depth></size>
<object><name>person</name><bndbox><xmin>48.0</xmin><ymin>240</ymin><xmax>195</xmax><ymax>371</ymax></bndbox></object>
<object><name>horse</name><bndbox><xmin>8</xmin><ymin>12</ymin><xmax>352</xmax><ymax>498</ymax></bndbox></object>
</annotation>"""


def test_voc_source_reads_annotations(tmp_path):
    for part, vocid, lst in (('trainval', 'VOC2007', 'trainval'), ('trainval', 'VOC2012', 'trainval'), ('test', 'VOC2007', 'test')):
        root = tmp_path / part / 'VOCdevkit' / vocid
        os.makedirs(root / 'Annotations'); os.makedirs(root / 'ImageSets' / 'Main', exist_ok=True); os.makedirs(root / 'JPEGImages', exist_ok=True)
        names = ['%s_%s_%d' % (vocid, lst, i) for i in range(2)]
        for n in names:
            (root / 'Annotations' / (n + '.xml')).write_text(VOC_XML.format(name=n))
        (root / 'ImageSets' / 'Main' / (lst + '.txt')).write_text('\n'.join(names) + '\n')
    # one VOC2012 annotation on no list -> validation sample
    (tmp_path / 'trainval' / 'VOCdevkit' / 'VOC2012' / 'Annotations' / 'extra_0.xml').write_text(VOC_XML.format(name='extra_0'))
    src = load_data_source('pascal_voc')
    assert isinstance(src, PascalVOCSource) and src.num_classes == 20 and src.lname2id['person'] == 14
    src.load_trainval_data(str(tmp_path), 0.025, require_image=False)
    assert (src.num_train, src.num_valid) == (6, 1)
    s = src.train_samples[0]
    assert s.imgsize == Size(500, 375) and [b.label for b in s.boxes] == ['person', 'horse']
    b = s.boxes[0]          # abs2prop(48, 195, 240, 371, 500x375)
    assert (b.center.x, b.center.y, b.size.w, b.size.h) == ((48 + 147 / 2) / 500, (240 + 131 / 2) / 375, 147 / 500, 131 / 375)
    assert src.colors['bicycle'] == (0, 74, 111)                 # rgb2bgr of (111, 74, 0)
