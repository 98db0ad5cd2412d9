"""CPU: oracle/boxes.py against the golden vectors captured from the imported reference."""
import numpy as np
import pytest
from oracle import boxes as ob
from golden_util import load, dense_pred, detect_cases

PRESETS = ['vgg300', 'vgg512']


@pytest.mark.parametrize('pname', PRESETS)
def test_g1_anchors(pname):
    g = load(f'g1_anchors_{pname}.npz')
    a = ob.anchors(ob.get_preset(pname))
    assert a.shape == (ob.PRESETS[pname]['num_anchors'], 4)
    assert np.array_equal(a, g['anchors'])
    assert np.array_equal(ob.anchors_abs(a), g['anchors_abs'].astype(np.float64))


def test_g1_known_answers():
    # SURVEY.md 8c known answers
    a3 = ob.anchors(ob.get_preset('vgg300')); a5 = ob.anchors(ob.get_preset('vgg512'))
    assert abs(a3.sum() - 11765.6967755814) < 1e-6
    assert abs(a5.sum() - 30856.2833939020) < 1e-6
    b3 = ob.anchors_abs(a3)
    assert list(b3[0]) == [-36, 63, -36, 63] and list(b3[5776]) == [-73, 126, -73, 126]
    assert list(b3[-1]) == [8, 991, 8, 991] and b3.min() == -376 and b3.max() == 1376 and b3.sum() == 17448308
    b5 = ob.anchors_abs(a5)
    assert list(b5[0]) == [-27, 42, -27, 42] and list(b5[-1]) == [13, 986, 13, 986] and b5.sum() == 49081896


def test_unknown_preset_raises():
    with pytest.raises(RuntimeError):
        ob.get_preset('vgg999')


@pytest.mark.parametrize('pname', PRESETS)
def test_g23_labels(pname):
    g = load(f'g23_labels_{pname}.npz')
    preset = ob.get_preset(pname)
    anch = ob.anchors(preset); aabs = ob.anchors_abs(anch)
    saw_empty = False
    for ci in range(int(g['ncases'][0])):
        gt, cls = g[f'gt_{ci}'], g[f'cls_{ci}']
        vec = ob.encode_labels(gt, cls, preset, 20, anch, aabs)
        pos = np.nonzero(vec[:, 20] == 0)[0]
        assert np.array_equal(pos, g[f'pos_{ci}'])
        assert np.array_equal(vec[pos], g[f'rows_{ci}'])
        neg = np.ones(len(vec), bool); neg[pos] = False
        assert np.all(vec[neg, 20] == 1) and np.all(vec[neg, :20] == 0) and np.all(vec[neg, 21:] == 0)
        best, good, iou = ob.overlap(np.array(ob.prop2abs(*gt[0]), np.float64), aabs, 0.5)
        assert np.array_equal(good, g[f'good_{ci}'])
        assert np.array_equal(iou[good], g[f'goodiou_{ci}'])
        assert (-1 if best is None else best) == int(g[f'best_{ci}'][0])
        saw_empty |= best is None
    assert saw_empty, 'fixture holds a GT box with no anchor above 0.5'


def test_labels_empty_gt():
    preset = ob.get_preset('vgg300')
    vec = ob.encode_labels(np.zeros((0, 4)), np.zeros((0,), int), preset, 20)
    assert np.all(vec[:, 20] == 1) and vec[:, :20].sum() == 0 and vec[:, 21:].sum() == 0


@pytest.mark.parametrize('pname', PRESETS)
def test_g45_detect(pname):
    g = load(f'g45_detect_{pname}.npz')
    anch = ob.anchors(ob.get_preset(pname))
    n = 0
    for pi, tag, thr, cap, max_out in detect_cases(g):
        pred = dense_pred(g, pi)
        det = ob.decode(pred, anch, thr, cap)
        assert np.array_equal(det['idx'], g[f'idx_{tag}'])
        assert np.array_equal(det['cls'], g[f'cls_{tag}'])
        assert np.array_equal(det['conf'], g[f'conf_{tag}'])
        assert np.array_equal(det['box'], g[f'box_{tag}'])
        assert np.array_equal(ob.nms_roundtrip(det['box']), g[f'nmsbox_{tag}'])
        assert np.array_equal(ob.suppress(det, max_out), g[f'keep_{tag}'])
        n += 1
    assert n >= 12


def test_g6_roundtrip_table():
    g = load('g6_roundtrip.npz')
    xi, xa = np.triu_indices(1000)
    z = np.zeros_like(xi)
    rt = ob.nms_roundtrip(np.stack([xi, xa, z, z], 1))
    bad = (rt[:, 0] != xi) | (rt[:, 1] != xa)
    assert bad.sum() == 5922
    assert np.array_equal(np.stack([xi[bad], xa[bad], rt[bad, 0], rt[bad, 1]], 1), g['exceptions'])


def test_g7_location():
    g = load('g7_location.npz')
    for i in range(len(g['box'])):
        assert np.array_equal(ob.encode_location(g['box'][i], g['anchor'][i]), g['enc'][i])
        l = g['loc'][i].copy(); l[l > 100] = 100
        x, y, w, h = ob.decode_location_np2(l, g['anchor'][i])
        assert [float(x), float(y), w, h] == list(g['dec'][i])


def test_prop2abs_truncates_toward_zero():
    x0, x1, y0, y1 = ob.prop2abs(0.5 / 38, 0.5 / 38, 0.1, 0.1)
    assert (int(x0), int(x1)) == (-36, 63)      # not -37: int() truncation (utils.py:108)


def test_decode_empty_and_nms_idempotent():
    anch = ob.anchors(ob.get_preset('vgg300'))
    pred = np.zeros((8732, 25), np.float32); pred[:, 20] = 1
    det = ob.decode(pred, anch, 0.5, 200)
    assert len(det['idx']) == 0 and len(ob.suppress(det)) == 0
    g = load('g45_detect_vgg300.npz')
    det = ob.decode(dense_pred(g, 3), anch, 0.5, None)
    keep = ob.suppress(det)
    det2 = {k: v[keep] for k, v in det.items()}
    # survivors re-suppressed (already class-grouped, conf-descending per class): nothing more goes
    assert len(ob.suppress(det2)) == len(keep)


def test_g10_list_nms():
    """suppress_overlaps / non_maximum_suppression on arbitrary box lists (ssdutils.py:232-318) vs the reference's picks."""
    g = load('g10_nms_lists.npz')
    for case in range(int(g['ncases'][0])):
        box, conf, lab = g[f'box_{case}'], g[f'conf_{case}'], g[f'label_{case}']
        recs = [(float(conf[i]), int(lab[i]), tuple(int(v) for v in ob.prop2abs(*box[i]))) for i in range(len(conf))]
        assert ob.suppress_list(recs, 0.45) == list(g[f'keep_{case}'])
        one = list(g[f'one_{case}'])
        assert ob.nms_list([(recs[i][0], recs[i][2]) for i in one], float(g[f'thr_{case}'][0])) == list(g[f'keep1_{case}'])
    assert ob.nms_list([], 0.5) == [] and ob.suppress_list([]) == []
