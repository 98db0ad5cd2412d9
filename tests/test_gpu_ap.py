"""-m gpu: the HIP average-precision kernel (ssd_average_precision) bit-exact against the golden vectors
of the imported reference and against the oracle on a larger seeded case; APCalculator mirror."""
import numpy as np
import pytest
from oracle import average_precision as oap
from golden_util import load
from ssd_tensorflow_amd import average_precision as apm
from ssd_tensorflow_amd._lib import lib, check, np_ptr
from ssd_tensorflow_amd.utils import Box, Point, Size, abs2prop

pytestmark = pytest.mark.gpu


def gpu_aps(db, dc, dk, ds, gb, gk, gs, ncls):
    db = np.ascontiguousarray(db, np.float32); dc = np.ascontiguousarray(dc, np.float32)
    dk = np.ascontiguousarray(dk, np.int32); ds = np.ascontiguousarray(ds, np.int32)
    gb = np.ascontiguousarray(gb, np.float64); gk = np.ascontiguousarray(gk, np.int32); gs = np.ascontiguousarray(gs, np.int32)
    ap = np.zeros(ncls); present = np.zeros(ncls, np.int32)
    check(lib.ssd_average_precision(0, len(dc), np_ptr(db), np_ptr(dc), np_ptr(dk), np_ptr(ds), len(gk), np_ptr(gb), np_ptr(gk),
                                    np_ptr(gs), ncls, 0.5, np_ptr(ap), np_ptr(present)))
    return ap, present


def test_g8_golden():
    g = load('g8_average_precision.npz')
    for c in range(int(g['ncases'][0])):
        ap, present = gpu_aps(g[f'det_box_{c}'], g[f'det_conf_{c}'], g[f'det_cls_{c}'], g[f'det_sample_{c}'],
                              g[f'gt_box_{c}'], g[f'gt_cls_{c}'], g[f'gt_sample_{c}'], 20)
        cls = g[f'ap_cls_{c}']
        assert sorted(np.nonzero(present)[0]) == sorted(cls)
        assert np.array_equal(ap[cls], g[f'ap_{c}'])


def test_large_random_vs_oracle():
    rng = np.random.default_rng(5)
    nimg, ncls = 400, 20
    gb, gk, gs, db, dc, dk, ds = [], [], [], [], [], [], []
    for img in range(nimg):
        for _ in range(int(rng.integers(0, 6))):
            x0, y0 = rng.integers(0, 700, 2); w, h = rng.integers(30, 300, 2)
            k = int(rng.integers(0, ncls))
            gb.append([x0, x0 + w, y0, y0 + h]); gk.append(k); gs.append(img)
            for rep in range(int(rng.integers(0, 4))):
                j = rng.integers(-40, 40, 4)
                db.append([x0 + j[0], x0 + w + j[1], y0 + j[2], y0 + h + j[3]]); dk.append(k if rng.random() < 0.8 else int(rng.integers(0, ncls))); ds.append(img)
                dc.append(rng.uniform(0.01, 1.0))
    dc = ((rng.permutation(len(db)) + 1) / (len(db) + 1.0)).astype(np.float32)     # distinct confidences: no ties
    assert len(np.unique(dc)) == len(dc)
    want = oap.compute_aps(np.array(db, np.float32), dc, dk, ds, np.array(gb, np.float64), gk, gs)
    ap, present = gpu_aps(db, dc, dk, ds, gb, gk, gs, ncls)
    assert len(want) == int(present.sum()) and len(dc) > 1500
    for k, v in want.items():
        assert ap[k] == v, (k, ap[k], v)


def test_apcalculator_mirror():
    g = load('g8_average_precision.npz')
    c = 1
    calc = apm.APCalculator()
    ns = int(max(g[f'det_sample_{c}'].max(), g[f'gt_sample_{c}'].max())) + 1
    for s in range(ns):
        def mk(b, k):
            ce, sz = abs2prop(int(b[0]), int(b[1]), int(b[2]), int(b[3]), Size(1000, 1000))
            return Box('c%d' % k, int(k), ce, sz)
        gts = [mk(b, k) for b, k, sm in zip(g[f'gt_box_{c}'], g[f'gt_cls_{c}'], g[f'gt_sample_{c}']) if sm == s]
        dets = [(cf, mk(b, k)) for b, k, sm, cf in zip(g[f'det_box_{c}'], g[f'det_cls_{c}'], g[f'det_sample_{c}'], g[f'det_conf_{c}']) if sm == s]
        calc.add_detections(gts, dets)
    aps = calc.compute_aps()
    # boxes went ints -> abs2prop -> prop2abs (not always the identity): compare with the oracle on the same round trip
    from oracle import boxes as ob
    rt = lambda arr: ob.nms_roundtrip(np.asarray(arr, np.int64))
    want = oap.compute_aps(rt(g[f'det_box_{c}']).astype(np.float32), g[f'det_conf_{c}'], g[f'det_cls_{c}'], g[f'det_sample_{c}'],
                           rt(g[f'gt_box_{c}']).astype(np.float64), g[f'gt_cls_{c}'], g[f'gt_sample_{c}'])
    assert {int(k[1:]): v for k, v in aps.items()} == want
    assert apm.APs2mAP(aps) == oap.aps2map(want)
    # data parallel (train.py gathered_aps): every rank's state() merged on rank 0 gives the AP of the whole sample --
    # here the samples are dealt to three "ranks" round-robin and merged in rank order
    import pickle
    states = [dict(det_params=[], det_confidence=[], det_labels=[], det_sample_ids=[], gt_boxes=[]) for _ in range(3)]
    for s in range(ns):
        st = states[s % 3]
        local = len(st['gt_boxes'])
        st['gt_boxes'].append(calc.gt_boxes[s])
        for pr, cf, l, sm in zip(calc.det_params, calc.det_confidence, calc.det_labels, calc.det_sample_ids):
            if sm == s:
                st['det_params'].append(pr); st['det_confidence'].append(cf); st['det_labels'].append(l); st['det_sample_ids'].append(local)
    merged = apm.APCalculator()
    for st in states:
        merged.merge(pickle.loads(pickle.dumps(st)))
    assert merged.compute_aps() == aps
    assert len(merged.gt_boxes) == ns and len(merged.det_confidence) == len(calc.det_confidence)
    calc.clear()
    assert calc.compute_aps() == {}
