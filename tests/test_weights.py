"""CPU: the a-trous fc6/fc7 decimation converter (ssdvgg.py:245-253, 273-280) against its loop restatement."""
import numpy as np
import pytest
from oracle import ssdvgg_ref as ref
from ssd_tensorflow_amd import weights


def test_decimation_matches_reference_loops():
    rng = np.random.default_rng(0)
    fc6_w = rng.normal(size=(7, 7, 512, 4096)).astype(np.float32); fc6_b = rng.normal(size=4096).astype(np.float32)
    fc7_w = rng.normal(size=(1, 1, 4096, 4096)).astype(np.float32); fc7_b = rng.normal(size=4096).astype(np.float32)
    w6, b6, w7, b7 = ref.decimate_fc_loops(fc6_w, fc6_b, fc7_w, fc7_b)
    g6, gb6 = weights.decimate_fc6(fc6_w, fc6_b)
    g7, gb7 = weights.decimate_fc7(fc7_w, fc7_b)
    assert g6.shape == (3, 3, 512, 1024) and g7.shape == (1, 1, 1024, 1024) and g6.dtype == np.float32
    assert np.array_equal(g6, w6.astype(np.float32)) and np.array_equal(gb6, b6.astype(np.float32))
    assert np.array_equal(g7, w7.astype(np.float32)) and np.array_equal(gb7, b7.astype(np.float32))
    with pytest.raises(ValueError):
        weights.decimate_fc6(fc6_w[:3], fc6_b)
    vgg = {n + s: np.zeros((3, 3, 1, 1) if s == '/filter' else (1,), np.float32) for n in weights.VGG_CONVS for s in ('/filter', '/biases')}
    vgg.update({'fc6/weights': fc6_w, 'fc6/biases': fc6_b, 'fc7/weights': fc7_w, 'fc7/biases': fc7_b})
    out = weights.vgg16_to_ssd(vgg)
    assert set(out) == {n + s for n in weights.VGG_CONVS + ['mod_conv6', 'mod_conv7'] for s in ('/filter', '/biases')}
