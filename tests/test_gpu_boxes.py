"""-m gpu: anchor / label / decode / NMS kernels through the C ABI, bit-exact against the golden
vectors captured from the imported reference (tests/golden) and against oracle/boxes.py on
fresh seeded inputs."""
import numpy as np
import pytest

from oracle import boxes as ob
from golden_util import load, dense_pred, detect_cases
from ssd_tensorflow_amd import ssdutils as su
from ssd_tensorflow_amd import transforms as tfm
from ssd_tensorflow_amd.utils import Box, Point, Size, Sample

pytestmark = pytest.mark.gpu
PRESETS = ['vgg300', 'vgg512']


@pytest.mark.parametrize('pname', PRESETS)
def test_anchors_bit_exact(pname):
    g = load(f'g1_anchors_{pname}.npz')
    a = su.anchors_array(pname)
    assert np.array_equal(a, g['anchors'])
    preset = su.get_preset_by_name(pname)
    anchors = su.get_anchors_for_preset(preset)
    assert len(anchors) == preset.num_anchors
    assert anchors[0].x == 0 and anchors[0].map == 0 and anchors[-1].map == len(preset.maps) - 1
    arr = su.anchors2array(anchors, Size(1000, 1000))
    assert arr.dtype == np.float64 and np.array_equal(arr, g['anchors_abs'].astype(np.float64))


def test_unknown_preset_raises_runtimeerror():
    with pytest.raises(RuntimeError, match='No such preset'):
        su.get_preset_by_name('vgg999')
    with pytest.raises(RuntimeError, match='No such preset'):
        su.anchors_array('vgg999')


@pytest.mark.parametrize('pname', PRESETS)
def test_labels_golden(pname):
    g = load(f'g23_labels_{pname}.npz')
    preset = su.get_preset_by_name(pname)
    n = int(g['ncases'][0])
    vec = su.encode_labels_batch(preset, 20, [g[f'gt_{i}'] for i in range(n)], [g[f'cls_{i}'] for i in range(n)])
    assert vec.shape == (n, preset.num_anchors, 25) and vec.dtype == np.float32
    for ci in range(n):
        pos = np.nonzero(vec[ci, :, 20] == 0)[0]
        assert np.array_equal(pos, g[f'pos_{ci}']), f'case {ci}: positive anchor set'
        rows, ref = vec[ci][pos], g[f'rows_{ci}']
        assert np.array_equal(rows[:, :21], ref[:, :21]), f'case {ci}: classes'
        # offsets: f64 log on the device may differ from libm in the last ulp before the f32 cast
        assert np.allclose(rows[:, 21:], ref[:, 21:], rtol=2e-7, atol=1e-7), f'case {ci}: offsets'
        neg = np.ones(preset.num_anchors, bool); neg[pos] = False
        assert np.all(vec[ci][neg, 20] == 1) and not vec[ci][neg, :20].any() and not vec[ci][neg, 21:].any()


@pytest.mark.parametrize('pname', PRESETS)
def test_labels_random_vs_oracle(pname):
    rng = np.random.default_rng(99)
    preset = su.get_preset_by_name(pname); op = ob.get_preset(pname)
    anch = ob.anchors(op); aabs = ob.anchors_abs(anch)
    gts, cls = [], []
    for i in range(16):
        n = int(rng.integers(0, 9))           # includes images with no GT box at all
        w = rng.uniform(0.02, 0.9, n); h = rng.uniform(0.02, 0.9, n)
        g = np.stack([rng.uniform(w / 2, 1 - w / 2), rng.uniform(h / 2, 1 - h / 2), w, h], 1).reshape(-1, 4)
        gts.append(g); cls.append(rng.integers(0, 20, n))
    vec = su.encode_labels_batch(preset, 20, gts, cls)
    for i in range(16):
        ref = ob.encode_labels(gts[i], cls[i], op, 20, anch, aabs)
        assert np.array_equal(vec[i][:, :21], ref[:, :21]), f'image {i}: class columns'
        assert np.allclose(vec[i][:, 21:], ref[:, 21:], rtol=2e-7, atol=1e-7)


def test_label_creator_transform_mirror():
    preset = su.get_preset_by_name('vgg300')
    t = tfm.LabelCreatorTransform(preset=preset, num_classes=20)
    gt = Sample('f', [Box('dog', 11, Point(0.5, 0.5), Size(0.4, 0.5))], Size(1000, 1000))
    data, vec, gt2 = t('img', None, gt)
    assert data == 'img' and gt2 is gt and vec.shape == (8732, 25)
    ref = ob.encode_labels(np.array([[0.5, 0.5, 0.4, 0.5]]), np.array([11]), ob.get_preset('vgg300'), 20)
    assert np.array_equal(vec[:, :21], ref[:, :21])


@pytest.mark.parametrize('pname', PRESETS)
def test_detect_golden(pname):
    g = load(f'g45_detect_{pname}.npz')
    preset = su.get_preset_by_name(pname)
    n = 0
    for pi, tag, thr, cap, max_out in detect_cases(g):
        pred = dense_pred(g, pi)
        before = pred.copy()
        dec = su.detect_batch(pred, preset, thr, cap, None, nms=False)[0]
        assert np.array_equal(pred, before), 'pred must not be modified'
        assert np.array_equal(dec['idx'], g[f'idx_{tag}']), f'{tag}: decode order'
        assert np.array_equal(dec['cls'], g[f'cls_{tag}'])
        assert np.array_equal(dec['conf'], g[f'conf_{tag}'])
        assert np.array_equal(dec['box'], g[f'box_{tag}']), f'{tag}: integer boxes'
        det = su.detect_batch(pred, preset, thr, cap, max_out, nms=True)[0]
        keep = g[f'keep_{tag}']
        assert np.array_equal(det['idx'], g[f'idx_{tag}'][keep]), f'{tag}: NMS survivors / order'
        assert np.array_equal(det['conf'], g[f'conf_{tag}'][keep])
        assert np.array_equal(det['box'], g[f'box_{tag}'][keep])
        n += 1
    assert n >= 12


def test_decode_boxes_and_suppress_overlaps_mirror():
    g = load('g45_detect_vgg300.npz')
    preset = su.get_preset_by_name('vgg300')
    anchors = su.get_anchors_for_preset(preset)
    pred = dense_pred(g, 3)
    boxes = su.decode_boxes(pred, anchors, 0.5, {7: 'cat'}, 200)
    tag = '3_0'
    assert len(boxes) == len(g[f'idx_{tag}'])
    oa = ob.anchors(ob.get_preset('vgg300'))
    det = ob.decode(pred, oa, 0.5, 200)
    for (conf, box), c, k, b in zip(boxes, det['conf'], det['cls'], det['box']):
        cx, cy, w, h = ob.abs2prop(*b)
        assert conf == c and box.labelid == k and (box.center.x, box.center.y, box.size.w, box.size.h) == (cx, cy, w, h)
        assert box.label == ('cat' if k == 7 else None)
    sel = su.suppress_overlaps(boxes)
    keep = g[f'keep_{tag}']
    assert [float(s[0]) for s in sel] == [float(c) for c in g[f'conf_{tag}'][keep]]
    assert su.suppress_overlaps([]) == []
    # a list that is not decode_boxes' own goes through the general box-list NMS: same survivors
    plain = list(boxes)
    sel2 = su.suppress_overlaps(plain)
    assert [(float(c), b) for c, b in sel2] == [(float(c), b) for c, b in sel]
    assert su.suppress_overlaps([(0.9, boxes[0][1])]) == [(0.9, boxes[0][1])]
    # ... and a filtered list follows the reference on what is left (oracle: ssdutils.py:232-318 restated)
    part = [bx for i, bx in enumerate(boxes) if i % 3 != 1]
    want = ob.suppress_list([(float(c), b.labelid, ob.prop2abs(b.center.x, b.center.y, b.size.w, b.size.h)) for c, b in part], 0.45)
    got = su.suppress_overlaps(part)
    assert [part.index(g) for g in got] == want


def test_nms_boxes_golden():
    """ssd_nms_boxes behind suppress_overlaps / non_maximum_suppression vs picks captured from the reference (G10)."""
    g = load('g10_nms_lists.npz')
    for case in range(int(g['ncases'][0])):
        box, conf, lab = g[f'box_{case}'], g[f'conf_{case}'], g[f'label_{case}']
        boxes = [(conf[i], Box('c%d' % lab[i], int(lab[i]), Point(float(box[i, 0]), float(box[i, 1])), Size(float(box[i, 2]), float(box[i, 3]))))
                 for i in range(len(conf))]
        got = su.suppress_overlaps(boxes)
        assert [next(i for i, bx in enumerate(boxes) if bx is s_) for s_ in got] == list(g[f'keep_{case}']), case
        one = [boxes[i] for i in g[f'one_{case}']]
        got = su.non_maximum_suppression(one, float(g[f'thr_{case}'][0]))
        assert [next(i for i, bx in enumerate(one) if bx is s_) for s_ in got] == list(g[f'keep1_{case}']), case


def test_nms_boxes_general_threshold_and_order():
    """ssd_nms_boxes against the numpy restatement of non_maximum_suppression / suppress_overlaps on random box lists:
    any threshold, class groups in first-appearance order, negative and tied confidences."""
    rng = np.random.default_rng(77)
    for case in range(12):
        n = int(rng.integers(1, 400))
        cx = rng.uniform(0.1, 0.9, n); cy = rng.uniform(0.1, 0.9, n)
        w = rng.uniform(0.05, 0.5, n); h = rng.uniform(0.05, 0.5, n)
        conf = rng.uniform(-0.2, 1.0, n).astype(np.float32)
        if case % 3 == 0:
            conf = np.round(conf * 8) / 8                      # many exact ties
        lab = rng.integers(0, 6, n) * 7 - 3                     # arbitrary label ids, negative included
        boxes = [(np.float32(conf[i]), Box('c%d' % lab[i], int(lab[i]), Point(float(cx[i]), float(cy[i])), Size(float(w[i]), float(h[i]))))
                 for i in range(n)]
        thr = [0.45, 0.3, 0.6, 0.05][case % 4]
        recs = [(float(c), b.labelid, ob.prop2abs(b.center.x, b.center.y, b.size.w, b.size.h)) for c, b in boxes]
        if thr == 0.45:
            got = su.suppress_overlaps(boxes)
            assert [boxes.index(g) for g in got] == ob.suppress_list(recs, 0.45), case
        one = [bx for bx in boxes if bx[1].labelid == boxes[0][1].labelid]
        got = su.non_maximum_suppression(one, thr)
        want = ob.nms_list([(float(c), ob.prop2abs(b.center.x, b.center.y, b.size.w, b.size.h)) for c, b in one], thr)
        assert [one.index(g) for g in got] == want, case
    assert su.non_maximum_suppression([], 0.5) == []


def test_detect_fast_and_general_paths_agree_with_oracle():
    """<= 1024 candidates per image run entirely in LDS (rank sort), more take the bitonic path: both against the oracle,
    in one batch (image 0 few, image 1 > 1024, image 2 none, image 3 exactly capped)."""
    rng = np.random.default_rng(99)
    A = 8732
    preset = su.get_preset_by_name('vgg300')
    oa = ob.anchors(ob.get_preset('vgg300'))
    pred = np.zeros((4, A, 25), np.float32); pred[:, :, 20] = 1
    for i, ncand in enumerate((300, 1500, 0, 1024)):
        hot = rng.choice(A, ncand, replace=False)
        cls = rng.integers(0, 20, ncand)
        conf = rng.uniform(0.2, 0.99, ncand).astype(np.float32)
        pred[i, hot, 20] = 1 - conf
        pred[i, hot, cls] = conf
        pred[i, :, 21:] = rng.normal(0, 0.3, (A, 4))
    for thr, cap, mo in ((0.2, None, 200), (0.2, 200, None), (0.5, None, None), (0.2, 1100, 400)):
        for nms in (True, False):
            dets = su.detect_batch(pred, preset, thr, cap, mo, nms=nms)
            for i in range(4):
                ref = ob.detect(pred[i], oa, thr, cap, mo) if nms else ob.decode(pred[i], oa, thr, cap)
                n = len(dets[i]['idx'])
                assert np.array_equal(dets[i]['idx'], ref['idx'][:n]) and np.array_equal(dets[i]['box'], ref['box'][:n]), (thr, cap, mo, nms, i)
                assert np.array_equal(dets[i]['conf'], ref['conf'][:n]) and np.array_equal(dets[i]['cls'], ref['cls'][:n])
                assert n == (len(ref['idx']) if mo is None else min(len(ref['idx']), mo))


def test_detect_batch_properties_full_size():
    """BASELINE config 5 shape: b=128 x 8732 anchors; properties that do not need the oracle at
    full size + the oracle on a sample of images."""
    rng = np.random.default_rng(1234)
    A, b = 8732, 128
    logits = rng.normal(0, 1, (b, A, 21)).astype(np.float32)
    logits[:, :, 20] += 4
    for i in range(b):
        hot = rng.choice(A - 6, 50, replace=False)
        cl = rng.integers(0, 20, 50)
        for k in range(6):
            logits[i, hot + k, cl] += 8 + rng.normal(0, 1, 50)
    e = np.exp(logits - logits.max(-1, keepdims=True))
    pred = np.concatenate([e / e.sum(-1, keepdims=True), rng.normal(0, 0.1, (b, A, 4))], -1).astype(np.float32)
    preset = su.get_preset_by_name('vgg300')
    dets = su.detect_batch(pred, preset, 0.5, None, 200, nms=True)
    oa = ob.anchors(ob.get_preset('vgg300'))
    for i, d in enumerate(dets):
        n = len(d['conf'])
        assert n <= 200 and np.all(d['conf'] >= np.float32(0.5))
        # class groups are contiguous; confidence descends inside a group
        seen = []
        for j in range(n):
            if not seen or seen[-1] != d['cls'][j]:
                assert d['cls'][j] not in seen
                seen.append(d['cls'][j])
            elif j:
                assert d['conf'][j] <= d['conf'][j - 1]
        assert np.all(d['box'][:, 0] <= d['box'][:, 1]) and np.all(d['box'][:, 1] <= 999)
        if i % 16 == 0:
            ref = ob.detect(pred[i], oa, 0.5, None, 200)
            assert np.array_equal(d['idx'], ref['idx']) and np.array_equal(d['box'], ref['box'])
            assert np.array_equal(d['conf'], ref['conf'])
    # idempotence of suppression: feeding only the survivors' rows back changes nothing
    i = 0
    keep_rows = dets[i]['idx']
    p2 = np.zeros((A, 25), np.float32); p2[:, 20] = 1
    p2[keep_rows] = pred[i, keep_rows]
    again = su.detect_batch(p2, preset, 0.5, None, None, nms=True)[0]
    assert np.array_equal(np.sort(again['idx']), np.sort(keep_rows))


def test_detect_empty_and_tiny_out_cap():
    preset = su.get_preset_by_name('vgg300')
    pred = np.zeros((2, 8732, 25), np.float32); pred[:, :, 20] = 1
    pred[1, 100, 3] = 0.9; pred[1, 100, 20] = 0.1
    d = su.detect_batch(pred, preset, 0.5, 200, None)
    assert len(d[0]['conf']) == 0 and len(d[1]['conf']) == 1 and d[1]['idx'][0] == 100 and d[1]['cls'][0] == 3


@pytest.mark.parametrize('pname', PRESETS)
def test_overlap_mirrors_golden(pname):
    """jaccard_overlap / compute_overlap / box2array / compute_location under their reference names."""
    g = load(f'g23_labels_{pname}.npz')
    g1 = load(f'g1_anchors_{pname}.npz')
    aabs = g1['anchors_abs'].astype(np.float64)
    for ci in range(int(g['ncases'][0])):
        b = g[f'gt_{ci}'][0]
        box = Box('x', 0, Point(float(b[0]), float(b[1])), Size(float(b[2]), float(b[3])))
        arr = su.box2array(box, Size(1000, 1000))
        ov = su.compute_overlap(arr, aabs, 0.5)
        assert [s.idx for s in ov.good] == list(g[f'good_{ci}'])
        assert np.array_equal(np.array([s.score for s in ov.good]), g[f'goodiou_{ci}'])      # IEEE f64 division: bit-exact
        assert (-1 if ov.best is None else ov.best.idx) == int(g[f'best_{ci}'][0])
    g7 = load('g7_location.npz')
    for i in range(16):
        bx = Box('x', 0, Point(*map(float, g7['box'][i, :2])), Size(*map(float, g7['box'][i, 2:])))
        an = su.Anchor(Point(*map(float, g7['anchor'][i, :2])), Size(*map(float, g7['anchor'][i, 2:])), 0, 0, 0, 0)
        assert np.array_equal(su.compute_location(bx, an), g7['enc'][i])
        c, s = su.decode_location(g7['loc'][i].copy(), an)
        assert [float(c.x), float(c.y), s.w, s.h] == list(g7['dec'][i])
