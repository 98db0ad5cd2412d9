"""CPU: oracle/average_precision.py against the golden vectors captured from the imported reference."""
import numpy as np
from oracle import average_precision as oap
from golden_util import load


def test_g8_average_precision():
    g = load('g8_average_precision.npz')
    for c in range(int(g['ncases'][0])):
        aps = oap.compute_aps(g[f'det_box_{c}'], g[f'det_conf_{c}'], g[f'det_cls_{c}'], g[f'det_sample_{c}'],
                              g[f'gt_box_{c}'], g[f'gt_cls_{c}'], g[f'gt_sample_{c}'])
        assert list(aps) == list(g[f'ap_cls_{c}'])
        assert np.array_equal(np.array(list(aps.values())), g[f'ap_{c}'])
        assert oap.aps2map(aps) == float(g[f'map_{c}'][0])


def test_ap_edge_cases():
    # a class with ground truth but no detection scores 0; detections of classes without ground truth are ignored
    aps = oap.compute_aps(np.zeros((1, 4), np.float32), [0.9], [3], [0], np.array([[0, 10, 0, 10.]]), [1], [0])
    assert aps == {1: 0.0} and oap.aps2map({}) == 0
    # a perfect detection
    aps = oap.compute_aps(np.array([[0, 10, 0, 10]], np.float32), [0.9], [1], [0], np.array([[0, 10, 0, 10.]]), [1], [0])
    assert aps == {1: 1.0}
