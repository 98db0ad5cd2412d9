"""-m gpu, SURVEY.md 8f N1: the batch augmentation kernels (csrc/augment.hip through ssd_augment_batch_dev)
against the numpy restatement oracle/augment.py on the reference's own train recipe.

What is pinned to the reference (tests/test_augment.py, golden vectors): every decision, the expand / crop
windows, the surviving boxes, and the brightness / contrast / channel-reorder pixel ops.  What is compared
here is the pixel half end to end.  The OpenCV-backed steps (HSV round trip, the five cv2.resize kernels) are
restatements on both sides (parity with cv2 itself unpinned: no cv2 in the build container).
Tolerances: an image that was expanded is floating point from then on -> 1e-5 relative; an image that never
left uint8 is rounded by cv2.resize -> equal up to 1 LSB on at most 0.1 % of the values (a double sum landing
within 1e-12 of .5 can round the other way)."""
import random

import numpy as np
import pytest
import torch

from oracle import augment as oa
from ssd_tensorflow_amd import transforms as T
from ssd_tensorflow_amd.ssdutils import get_preset_by_name
from ssd_tensorflow_amd.utils import Sample, Box, Point, Size

pytestmark = pytest.mark.gpu


def _sample(name, size, boxes, cls):
    return Sample(name, [Box('c%d' % c, int(c), Point(b[0], b[1]), Size(b[2], b[3])) for b, c in zip(boxes, cls)], Size(*size))


def _compare(got, want, expanded, tag):
    if expanded:
        err = np.abs(got - want).max() / 255.0
        assert err < 1e-5, f'{tag}: float path max err {err:.3e} (of 255)'
    else:
        diff = np.abs(got - want)
        assert diff.max() <= 1.0, f'{tag}: uint8 path max diff {diff.max()}'
        assert (diff > 0).mean() < 1e-3, f'{tag}: uint8 path {100 * (diff > 0).mean():.3f}% of values differ'
        assert np.array_equal(got, np.floor(got)) and got.min() >= 0 and got.max() <= 255


@pytest.mark.parametrize('pname', ['vgg300', 'vgg512'])
def test_train_recipe_batch(pname):
    preset = get_preset_by_name(pname)
    S = (preset.image_size.w, preset.image_size.h)
    nrng = np.random.default_rng(11 if pname == 'vgg300' else 12)
    plans, wants, seen = [], [], set()
    b = 24
    for case in range(b):
        size, boxes, cls = oa.synth_sample(nrng)
        # smooth-ish content so interpolation differences would show: low-frequency noise upsampled + per-pixel noise
        base = nrng.integers(0, 256, (size[1] // 8 + 2, size[0] // 8 + 2, 3)).astype(np.float32)
        img = np.kron(base, np.ones((8, 8, 1), np.float32))[:size[1], :size[0]] * 0.8 + nrng.integers(0, 52, (size[1], size[0], 3))
        img = np.clip(img, 0, 255).astype(np.uint8)
        tfs = [t for t in T.build_train_transforms(preset, 20, 50, 0.5, images={'im': img}) if not isinstance(t, T.LabelCreatorTransform)]
        random.seed(7000 + case)
        args = (None, None, _sample('im', size, boxes, cls))
        for t in tfs:
            args = t(*args)
        plans.append(args[0])
        p = oa.plan(oa.new_rng(7000 + case), size, boxes, cls, 50, 0.5)
        wants.append((oa.apply(p, img, S), p))
        seen.add(p['resize_alg']); seen.update(n for n, v in p['distort'] if v is not None)
    assert seen >= {0, 1, 2, 3, 4, 'contrast', 'saturation', 'hue'}, seen          # every kernel path is exercised
    out = T.augment_batch(plans, S[0], S[1])
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert got.shape == (b, S[1], S[0], 3) and got.dtype == np.float32
    for i, (want, p) in enumerate(wants):
        _compare(got[i], want, p['expand'] is not None, f'{pname} case {i} alg {p["resize_alg"]} distort {p["distort"]}')


@pytest.mark.parametrize('alg', [0, 1, 2, 3, 4])
def test_every_resize_kernel_up_and_down(alg):
    """each cv2.resize restatement alone, enlarging and shrinking, uint8 and (expanded) floating point"""
    nrng = np.random.default_rng(100 + alg)
    plans, wants = [], []
    # 316 -> 300: (d + 0.5) * 316 / 300 - 0.5 falls a hair below an integer for some d (the float-rounded coordinate decides)
    for (w, h) in ((97, 61), (300, 300), (640, 517), (1200, 900), (150, 700), (316, 316)):
        img = nrng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        for expand in (None, (Size(w + 37, h + 11), 5, 20)):
            pl = T.ImagePlan(img)
            pl.expand = expand
            pl.resize = (300, 300, alg)
            plans.append(pl)
            p = dict(brightness=None, distort=[], reorder=None, expand=None if expand is None else (tuple(expand[0]), expand[1], expand[2]),
                     crop=None, flip=False, resize_alg=alg)
            wants.append((oa.apply(p, img, (300, 300)), expand is not None))
    got = T.augment_batch(plans, 300, 300).cpu().numpy()
    for i, (want, expanded) in enumerate(wants):
        _compare(got[i], want, expanded, f'alg {alg} case {i}')


def test_photometric_chain_exact():
    """no resize arithmetic (same size, NEAREST): brightness, every distort order, reorder, flip, crop are exact"""
    nrng = np.random.default_rng(3)
    img = nrng.integers(0, 256, (40, 56, 3)).astype(np.uint8)
    plans, wants = [], []
    for chain in ([('contrast', 1.37), ('saturation', 0.61), ('hue', -17)], [('saturation', 1.49), ('hue', 18), ('contrast', 0.5)],
                  [('hue', 5)], [('saturation', 0.9)], []):
        for br in (None, -32, 19):
            for reorder in (None, [2, 0, 1]):
                for flip in (False, True):
                    crop = (3, 43, 2, 38)          # xmin, xmax, ymin, ymax
                    p = dict(brightness=br, distort=chain, reorder=reorder, expand=None, crop=crop, flip=flip, resize_alg=0)
                    wants.append(oa.apply(p, img, (40, 36)))
                    pl = T.ImagePlan(img)
                    pl.brightness = br
                    pl.distort = [({'contrast': 0, 'saturation': 1, 'hue': 2}[n], float(v)) for n, v in chain]
                    pl.reorder = reorder or [0, 1, 2]
                    pl.crop = (3, 2, 40, 36); pl.flip = flip; pl.resize = (40, 36, 0)
                    plans.append(pl)
    got = T.augment_batch(plans, 40, 36).cpu().numpy()
    for i, want in enumerate(wants):
        assert np.array_equal(got[i], want), f'photometric case {i}: max diff {np.abs(got[i] - want).max()}'


def test_bad_plans_fail_loudly():
    img = np.zeros((10, 10, 3), np.uint8)
    pl = T.ImagePlan(img)
    with pytest.raises(ValueError):
        T.augment_batch([pl], 300, 300)                     # never resized
    pl.resize = (300, 300, 9)
    with pytest.raises(RuntimeError):
        T.augment_batch([pl], 300, 300)                     # unknown algorithm: rejected by the library
    pl.resize = (300, 300, 1); pl.crop = (5, 5, 10, 10)
    with pytest.raises(RuntimeError):
        T.augment_batch([pl], 300, 300)                     # crop window outside the frame


def test_redraw_test_agrees_with_the_label_encoder():
    """has_positive_anchor (host, for the <= 50 redraw loop) == "the encoded label has a non-background row" """
    from ssd_tensorflow_amd.ssdutils import has_positive_anchor, encode_labels_batch
    preset = get_preset_by_name('vgg300')
    nrng = np.random.default_rng(8)
    n_neg = 0
    for case in range(40):
        n = int(nrng.integers(1, 4))
        w = nrng.uniform(0.005, 0.25, n); h = nrng.uniform(0.005, 0.25, n)       # small boxes: some cases have no anchor above 0.5
        boxes = np.stack([nrng.uniform(w / 2, 1 - w / 2), nrng.uniform(h / 2, 1 - h / 2), w, h], 1)
        cls = nrng.integers(0, 20, n)
        vec = encode_labels_batch(preset, 20, [boxes], [cls])[0]
        want = np.count_nonzero(vec[:, 20]) < vec.shape[0]
        got = has_positive_anchor(preset, [Box('x', int(c), Point(b[0], b[1]), Size(b[2], b[3])) for b, c in zip(boxes, cls)])
        assert got == want, case
        n_neg += not want
    assert 3 < n_neg < 37


def test_training_data_augmented_batches_feed_the_step():
    from ssd_tensorflow_amd.training_data import TrainingData
    from ssd_tensorflow_amd.ssdvgg import SSDVGG, Session
    random.seed(5)
    td = TrainingData(None, 'vgg300', num_train=10, num_valid=4, augment=True)
    sess = Session(0)
    net = SSDVGG(sess, td.preset)
    net.build_from_vgg(None, 20, max_batch=4)
    net.build_optimizer(learning_rate=1e-4)
    seen = 0
    for x, y, gt in td.train_generator(4):
        assert torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and tuple(x.shape[1:]) == (300, 300, 3)
        assert torch.is_tensor(y) and y.is_cuda and tuple(y.shape) == (x.shape[0], 8732, 25) and len(gt) == x.shape[0]
        assert bool(torch.isfinite(x).all()) and float(x.min()) >= -200 and float(x.max()) <= 455      # cubic / lanczos overshoot on noise (float images only)
        for k in range(x.shape[0]):                                        # the redraw loop left (almost) no sample without a positive
            assert int((y[k][:, 20] != 0).sum()) < 8732
        res, L, _ = sess.run([net.result, net.losses, net.optimizer], feed_dict={net.image_input: x, net.labels: y})
        assert res.shape == tuple(y.shape) and np.isfinite(L['total'])
        seen += x.shape[0]
    assert seen == 10                                                      # ragged last batch (2 samples) included
    for x, y, gt in td.valid_generator(4):
        res, L = sess.run([net.result, net.losses], feed_dict={net.image_input: x, net.labels: y})
        assert np.isfinite(L['total'])
    sess.close()


def test_free_compositions_on_the_batch_kernel():
    """Round 5: transform lists composed outside the recipe's order (tests/compose_util.py: crop / expand / flip after a flip, two
    expands, a second photometric pass) through the batch kernel against the FREE composition of the oracle's pixel operations
    in the user's order.  Same tolerances as the recipe test."""
    import compose_util as cu
    plans, wants, expanded = [], [], []
    for li, steps in enumerate(cu.FREE_LISTS):
        for rep in range(2):
            img = cu.test_image(300 + 10 * li + rep)
            seed = 9100 + 10 * li + rep
            plan, _ = cu.compose_mirror(steps, img, seed)
            plans.append(plan)
            wants.append(cu.compose_pixels(steps, img, seed))
            expanded.append(plan.expand is not None)
    out = T.augment_batch(plans, 96, 80)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert any(p.extra for p in plans) and any(expanded) and not all(expanded)
    for i, (w, e) in enumerate(zip(wants, expanded)):
        _compare(got[i], w, e, f'free list {i // 2} rep {i % 2}')


def _compare_any(got, want, plan, tag):
    """A plan whose array is float64 to the end is compared as such; one that ends in uint8 (never expanded, or made uint8 again by a
    Brightness / Contrast) within the uint8 path's bounds -- and when steps run BEHIND the resize, the few values the resize rounds
    the other way (< 0.1 %) may come out of those steps further apart than one level (a hue rotation is not 1-Lipschitz)."""
    ends_float = plan.is_float
    if ends_float:
        err = np.abs(got - want).max() / 255.0
        assert err < 1e-5, f'{tag}: float path max err {err:.3e} (of 255)'
        return
    diff = np.abs(got - want)
    assert (diff > 0).mean() < 2e-3, f'{tag}: uint8 path {100 * (diff > 0).mean():.3f}% of values differ'
    if not plan.post:
        assert diff.max() <= 1.0, f'{tag}: uint8 path max diff {diff.max()}'
    assert np.array_equal(got, np.floor(got)) and got.min() >= 0 and got.max() <= 255


def test_round6_compositions_on_the_batch_kernel():
    """Round 6 (transforms.py:117-391 are independent callables): the orders rounds 1-5 refused -- pointwise steps behind an expand
    (image AND canvas), Hue / Saturation behind a crop / on the canvas's rows, an expand behind a crop (visible window), another
    mean value, steps and a flip behind the resize -- through the batch kernel against the FREE composition of the oracle's pixel
    operations in the user's order; then 80 random lists the mirror accepts."""
    import random
    import compose_util as cu
    plans, wants, tags = [], [], []
    for li, steps in enumerate(cu.ROUND6_LISTS):
        for rep in range(2):
            img = cu.test_image(700 + 10 * li + rep)
            seed = 9300 + 10 * li + rep
            plan, _ = cu.compose_mirror(steps, img, seed)
            plans.append(plan); wants.append(cu.compose_pixels(steps, img, seed)); tags.append(f'round-6 list {li} rep {rep}')
    rng = random.Random(78)
    vocab = [('brightness', {}), ('contrast', {}), ('saturation', {}), ('hue', {}), ('reorder', {}), ('expand', dict(max_ratio=1.6)),
             ('expand', dict(max_ratio=1.3, mean_value=[7.25, 99.5, 250.0])), ('crop', cu.WIN), ('crop', dict(window=(0.0, 0.8, 0.1, 1.0))), ('flip', {})]
    case = 0
    while len(plans) < 2 * len(cu.ROUND6_LISTS) + 80:
        case += 1
        steps = [rng.choice(vocab) for _ in range(rng.randint(1, 9))]
        steps.insert(rng.randint(max(0, len(steps) - 3), len(steps)), cu.RS)
        img = cu.test_image(2000 + case, (90, 70))
        try:
            plan, _ = cu.compose_mirror(steps, img, 6000 + case)
        except RuntimeError:
            continue
        plans.append(plan); wants.append(cu.compose_pixels(steps, img, 6000 + case)); tags.append('random list %d %s' % (case, [n for n, _ in steps]))
    got = []
    for i in range(0, len(plans), 32):
        got.append(T.augment_batch(plans[i:i + 32], 96, 80).cpu().numpy())
    got = np.concatenate(got)
    assert any(p.clip is not None for p in plans) and any(p.post for p in plans) and any(p.out_flip for p in plans)
    assert any(p.fill_from is not None and len(p.extra) > p.fill_from for p in plans)
    for g, w, p, tag in zip(got, wants, plans, tags):
        _compare_any(g, w, p, tag)
