"""-m gpu: the whole SSD-VGG step (forward, loss, backward, momentum update) through the C ABI
against the CPU oracle (oracle/ssdvgg_ref.py, parity unpinned w.r.t. TensorFlow: see its
header).  Tolerance 1e-3 relative (BASELINE.json north_star)."""
import numpy as np
import pytest

from oracle import boxes as ob
from oracle import ssdvgg_ref as ref
from gpu_util import rel_err, max_rel
from ssd_tensorflow_amd.ssdvgg import SSDVGG, Session, LearningRate

pytestmark = pytest.mark.gpu
TOL = 1e-3


def make_pair(pname, b, seed=42, training=True):
    preset = ob.get_preset(pname)
    w = ref.init_params(preset, 20, seed=seed, bias_scale=0.01)
    m = ref.RefModel(pname, params=w)
    sess = Session(0)
    net = SSDVGG(sess, pname)
    net.build_from_vgg(None, 20, max_batch=b, training=training, weights=w)
    return preset, m, sess, net


def report(tag, e):
    print(f'    {tag:<40s} {e:.3e}')
    return e


def test_forward_loss_backward_vgg300():
    b = 2
    preset, m, sess, net = make_pair('vgg300', b)
    rng = np.random.default_rng(1234)
    x, y, _ = ref.synth_batch(rng, b, preset)
    m.set_optimizer([0.001], [], 0.9, 0.0005)
    net.build_optimizer(learning_rate=0.001, weight_decay=0.0005, momentum=0.9)

    keep = {}
    r_ref, L_ref = m.eval_step(x, y, keep)
    r, L = sess.run([net.result, net.losses], feed_dict={net.image_input: x, net.labels: y})
    worst = 0
    for name in ['conv1_1', 'conv1_2', 'pool1', 'conv2_2', 'conv3_3', 'pool3', 'conv4_3', 'norm_conv4_3', 'conv5_3',
                 'mod_pool5', 'mod_conv6', 'mod_conv7', 'conv8_2', 'conv9_2', 'conv10_2', 'conv11_2']:
        worst = max(worst, report('activation ' + name, max_rel(net.activation(name, b), keep[name].numpy())))
    assert worst < TOL
    assert report('result', max_rel(r, r_ref)) < TOL
    for k in ('total', 'localization', 'confidence', 'l2'):
        assert abs(L[k] - L_ref[k]) < TOL * abs(L_ref[k]), (k, L[k], L_ref[k])

    # gradients of every variable
    _, _, g_ref = m.grads(x, y)
    import torch
    xt = torch.from_numpy(x).cuda(); yt = torch.from_numpy(y).cuda()
    net.forward_backward_dev(xt, yt)
    torch.cuda.synchronize()
    g = net.save_gradients()
    assert set(g) == set(g_ref)
    worst = 0
    for k in g_ref:
        e = rel_err(g[k], g_ref[k])
        if e > 1e-4:
            report('grad ' + k, e)
        worst = max(worst, e)
    print('    worst gradient rel-L2 error', worst)
    assert worst < TOL

    # two optimizer steps (momentum carries over); weights must track the oracle
    for step in range(2):
        m.train_step(x, y)
        _, Ls = sess.run([net.result, net.losses, net.optimizer], feed_dict={net.image_input: x, net.labels: y})[:2]
    assert net.global_step == 2
    w_ref = m.numpy_params(); w = net.save_variables()
    worst = max(rel_err(w[k], w_ref[k]) for k in w_ref)
    assert report('weights after 2 steps', worst) < 1e-5
    # the update must have moved them by lr * accumulated gradient, not by nothing
    k = 'conv4_2/filter'
    w0 = ref.init_params(preset, 20, seed=42, bias_scale=0.01)[k]
    assert np.abs(w[k] - w0).max() > 0
    sess.close()


def test_forward_loss_vgg512_b1():
    b = 1
    preset, m, sess, net = make_pair('vgg512', b)
    rng = np.random.default_rng(77)
    x, y, _ = ref.synth_batch(rng, b, preset)
    net.build_optimizer(learning_rate=0.001)
    m.set_optimizer([0.001], [], 0.9, 0.0005)
    r_ref, L_ref, g_ref = m.grads(x, y)
    r, L = sess.run([net.result, net.losses], feed_dict={net.image_input: x, net.labels: y})
    assert r.shape == (1, 24564, 25)
    assert report('vgg512 result', max_rel(r, r_ref)) < TOL
    for k in L_ref:
        assert abs(L[k] - L_ref[k]) < TOL * abs(L_ref[k]), (k, L[k], L_ref[k])
    import torch
    net.forward_backward_dev(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    torch.cuda.synchronize()
    g = net.save_gradients()
    worst = max(rel_err(g[k], g_ref[k]) for k in g_ref)
    assert report('vgg512 worst gradient', worst) < TOL
    sess.close()


def test_ragged_last_batch_and_no_positive_sample():
    """b smaller than max_batch (training_data.py:189) and a sample without positives (its loss is 0)."""
    preset, m, sess, net = make_pair('vgg300', 4)
    rng = np.random.default_rng(5)
    x, y, _ = ref.synth_batch(rng, 3, preset)
    y[1] = 0; y[1, :, 20] = 1
    net.build_optimizer(learning_rate=0.001)
    r_ref, L_ref, g_ref = m.grads(x, y)
    r, L = sess.run([net.result, net.losses], feed_dict={net.image_input: x, net.labels: y})
    assert r.shape[0] == 3 and max_rel(r, r_ref) < TOL
    for k in L_ref:
        assert abs(L[k] - L_ref[k]) < TOL * abs(L_ref[k]), (k, L[k], L_ref[k])
    sess.close()


def test_infer_and_detect_last_matches_oracle_boxes():
    preset, m, sess, net = make_pair('vgg300', 2, training=False)
    rng = np.random.default_rng(8)
    x = ref.synth_images(rng, 2, preset)
    r = sess.run(net.result, feed_dict={net.image_input: x, net.keep_prob: 1})
    assert max_rel(r, m.infer(x)) < TOL
    # decode + NMS of that very result on the GPU vs the box oracle on the GPU's own floats
    thr = float(np.quantile(r[:, :, :20].max(-1), 0.999))     # untrained net: pick a threshold with candidates
    dets = net.detect_last(2, thr, None, 200)
    oa = ob.anchors(preset)
    for i in range(2):
        want = ob.detect(r[i], oa, thr, None, 200)
        assert np.array_equal(dets[i]['idx'], want['idx']) and np.array_equal(dets[i]['box'], want['box'])
    sess.close()


def test_shape_and_dtype_errors():
    sess = Session(0)
    net = SSDVGG(sess, 'vgg300')
    net.build_from_vgg(None, 20, max_batch=1, training=False)
    with pytest.raises(ValueError):
        net.infer(np.zeros((1, 299, 300, 3), np.float32))
    with pytest.raises(ValueError):
        net.infer(np.zeros((2, 300, 300, 3), np.float32))      # beyond max_batch
    with pytest.raises(RuntimeError):
        net.build_optimizer()                                    # inference-only handle
    sess.close()
    with pytest.raises(RuntimeError, match='No such preset'):
        SSDVGG(None, 'vgg999')
