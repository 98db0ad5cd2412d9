"""SURVEY.md 8f N1, CPU half: the augmentation DECISIONS and GEOMETRY.
  * oracle/augment.py against the golden vectors captured from the imported reference
    (tests/golden/g9_augment.npz: RandomTransform(Expand) -> SamplePicker, and Brightness -> Contrast -> Reorder pixels);
  * the product's transform mirror (ssd_tensorflow_amd/transforms.py, plan mode: no pixel is touched on the
    host) against the same vectors and, for the whole train recipe, against the oracle's plan() under the same
    `random` stream."""
import random

import numpy as np
import pytest

from golden_util import load
from oracle import augment as oa


def g9():
    return load('g9_augment.npz')


def test_oracle_geometry_matches_reference_goldens():
    g = g9()
    for case in range(int(g['ncases'][0])):
        size = tuple(int(v) for v in g[f'size_{case}']); boxes = [tuple(b) for b in g[f'boxes_{case}']]; cls = list(g[f'cls_{case}'])
        rng = oa.new_rng(int(g[f'seed_{case}'][0]))
        ex = oa.plan_expand(rng, size, boxes, cls, 0.5)
        sz, b, c = (size, boxes, cls) if ex is None else (ex[0], ex[3], ex[4])
        win, sz2, b2, c2 = oa.plan_sample_picker(rng, sz, b, c, 50)
        assert tuple(sz2) == tuple(g[f'out_size_{case}'])
        assert np.array_equal(np.array(b2, np.float64).reshape(-1, 4), g[f'out_boxes_{case}'])          # bit-exact
        assert list(c2) == list(g[f'out_cls_{case}'])
        # the window: the coordinate image's corner pixels say where the result came from
        yy, xx = np.mgrid[0:size[1], 0:size[0]]
        img = np.stack([yy, xx, np.full_like(yy, 7)], -1).astype(np.int32)
        frame = oa.expand(img, *ex[:3]) if ex is not None else img
        x0, y0 = (0, 0) if win is None else (win[0], win[2])
        out = frame[y0:y0 + sz2[1], x0:x0 + sz2[0]]
        assert np.array_equal(np.asarray(out[0, 0], np.float64), g[f'corner00_{case}'])
        assert np.array_equal(np.asarray(out[-1, -1], np.float64), g[f'corner11_{case}'])


def test_oracle_pixels_match_reference_goldens():
    g = g9()
    for case in range(int(g['npix'][0])):
        rng = oa.new_rng(int(g[f'pix_seed_{case}'][0]))
        o = g[f'pix_in_{case}']
        b = oa.plan_brightness(rng)
        o = oa.brightness(o, b) if b is not None else o
        c = oa.plan_contrast(rng)
        o = oa.contrast(o, c) if c is not None else o
        r = oa.plan_reorder(rng)
        o = o[:, :, r] if r is not None else o
        assert o.dtype == np.uint8 and np.array_equal(o, g[f'pix_out_{case}'])


def _sample(size, boxes, cls):
    from ssd_tensorflow_amd.utils import Sample, Box, Point, Size
    return Sample('img', [Box('c%d' % c, int(c), Point(b[0], b[1]), Size(b[2], b[3])) for b, c in zip(boxes, cls)], Size(*size))


def test_product_transforms_match_reference_goldens():
    from ssd_tensorflow_amd import transforms as T
    g = g9()
    rnd_expand = T.RandomTransform(prob=0.5, transform=T.ExpandTransform(max_ratio=4.0, mean_value=[104, 117, 123]))
    picker = T.SamplePickerTransform(samplers=[T.SamplerTransform(sample=False)] + [T.build_sampler(o, 50) for o in (0.1, 0.3, 0.5, 0.7, 0.9, 1.0)])
    for case in range(int(g['ncases'][0])):
        size = tuple(int(v) for v in g[f'size_{case}'])
        gt = _sample(size, g[f'boxes_{case}'], g[f'cls_{case}'])
        plan = T.ImagePlan(np.zeros((size[1], size[0], 3), np.uint8))
        random.seed(int(g[f'seed_{case}'][0]))
        d, _, gt2 = rnd_expand(plan, None, gt)
        d, _, gt2 = picker(d, None, gt2)
        assert (gt2.imgsize.w, gt2.imgsize.h) == tuple(g[f'out_size_{case}']) == tuple(d.size)
        got = np.array([[b.center.x, b.center.y, b.size.w, b.size.h] for b in gt2.boxes], np.float64).reshape(-1, 4)
        assert np.array_equal(got, g[f'out_boxes_{case}']) and [b.labelid for b in gt2.boxes] == list(g[f'out_cls_{case}'])
        # window origin in the loaded image's coordinates: corner pixel (row, col) of the coordinate image, or the mean value
        x0, y0 = (d.crop[0], d.crop[1]) if d.crop is not None else (0, 0)
        if d.expand is not None:
            x0 -= d.expand[2]; y0 -= d.expand[1]
        inside = 0 <= x0 < size[0] and 0 <= y0 < size[1]
        want = g[f'corner00_{case}']
        assert (list(want) == [y0, x0, 7]) if inside else (list(want) == [104, 117, 123])


def test_product_recipe_draws_like_the_oracle_plan():
    """the whole train recipe (process_dataset.py:126-136) minus the GPU label encoder: same stream, same decisions"""
    from ssd_tensorflow_amd import transforms as T
    from ssd_tensorflow_amd.ssdutils import get_preset_by_name
    preset = get_preset_by_name('vgg300')
    nrng = np.random.default_rng(5)
    n_exp = n_crop = n_flip = 0
    for case in range(60):
        size, boxes, cls = oa.synth_sample(nrng)
        img = nrng.integers(0, 256, (size[1], size[0], 3)).astype(np.uint8)
        tfs = [t for t in T.build_train_transforms(preset, 20, 50, 0.5, images={'img': img}) if not isinstance(t, T.LabelCreatorTransform)]
        random.seed(300 + case)
        args = (None, None, _sample(size, boxes, cls))
        for t in tfs:
            args = t(*args)
        d, _, gt = args
        p = oa.plan(oa.new_rng(300 + case), size, boxes, cls, 50, 0.5)
        assert d.brightness == p['brightness']
        kinds = {'contrast': 0, 'saturation': 1, 'hue': 2}
        assert d.distort == [(kinds[n], float(v)) for n, v in p['distort'] if v is not None]
        assert d.reorder == (p['reorder'] if p['reorder'] is not None else [0, 1, 2])
        assert (None if d.expand is None else (tuple(d.expand[0]), d.expand[1], d.expand[2])) == p['expand']
        assert (None if d.crop is None else (d.crop[0], d.crop[0] + d.crop[2], d.crop[1], d.crop[1] + d.crop[3])) == (None if p['crop'] is None else tuple(p['crop']))
        assert d.flip == p['flip'] and d.resize == (300, 300, p['resize_alg'])
        got = np.array([[b.center.x, b.center.y, b.size.w, b.size.h] for b in gt.boxes], np.float64).reshape(-1, 4)
        assert np.array_equal(got, np.array(p['boxes'], np.float64).reshape(-1, 4)) and [b.labelid for b in gt.boxes] == p['cls']
        n_exp += d.expand is not None; n_crop += d.crop is not None; n_flip += d.flip
    assert n_exp > 10 and n_crop > 10 and n_flip > 10          # the cases exercise every branch


def test_plan_order_is_enforced():
    from ssd_tensorflow_amd import transforms as T
    import pytest
    p = T.ImagePlan(np.zeros((8, 8, 3), np.uint8))
    gt = _sample((8, 8), [(0.5, 0.5, 0.5, 0.5)], [1])
    p, _, gt = T.HorizontalFlipTransform()(p, None, gt)
    p, _, gt = T.BrightnessTransform(delta=3)(p, None, gt)         # round 5: a per-pixel step commutes with the flip
    p, _, gt = T.ExpandTransform(max_ratio=2.0, mean_value=[104, 117, 123])(p, None, gt)
    q, _, _ = T.BrightnessTransform(delta=3)(p, None, gt)           # round 6: behind an expand it transforms the canvas too (an extra step)
    assert q.extra and q.fill_from == 0 and not q.is_float and p.is_float and not p.extra      # (and the plan handed in is not touched)
    with pytest.raises(RuntimeError):                              # cv2.cvtColor on the float64 canvas: the reference raises too
        T.HueTransform(delta=3)(p, None, gt)
    r, _, _ = T.ResizeTransform(width=4, height=4, algorithms=[1])(p, None, gt)
    with pytest.raises(NotImplementedError):                       # no intermediate image behind the resize
        T.ExpandTransform(max_ratio=2.0, mean_value=[104, 117, 123])(r, None, gt)
    with pytest.raises(ValueError):
        T.ImagePlan(np.zeros((8, 8, 3), np.float32))


def test_native_sampler_trials_are_the_python_loops():
    """ssd_sampler_trials (csrc/planner.hip: all samplers of a SamplePickerTransform in one native call on the state of
    Python's Mersenne Twister) against the Python trial loops: the same plans, the same transformed boxes and the same
    generator state afterwards on the whole train recipe -- so the golden-vector cases above, which run the native path,
    pin it to the reference as well."""
    from ssd_tensorflow_amd import transforms as T
    from ssd_tensorflow_amd.ssdutils import get_preset_by_name
    from ssd_tensorflow_amd.utils import Box, Point, Sample, Size
    preset = get_preset_by_name('vgg300')
    host = [t for t in T.build_train_transforms(preset, 20, 50, 0.5) if not isinstance(t, T.LabelCreatorTransform)]
    rng = np.random.default_rng(0)
    samples = []
    for i in range(120):
        W, H = int(rng.integers(120, 700)), int(rng.integers(120, 700))
        n = int(rng.integers(1, 6)); w = rng.uniform(0.02, 0.7, n); h = rng.uniform(0.02, 0.7, n)
        cx = rng.uniform(w / 2, 1 - w / 2); cy = rng.uniform(h / 2, 1 - h / 2)
        boxes = [Box('x', 0, Point(float(a), float(b)), Size(float(c), float(d))) for a, b, c, d in zip(cx, cy, w, h)]
        samples.append(({'f%d' % i: np.zeros((H, W, 3), np.uint8)}, Sample('f%d' % i, boxes, Size(W, H))))

    def run(native):
        saved, T.NATIVE_SAMPLER = T.NATIVE_SAMPLER, native
        try:
            out = []
            for k, (imgs, s) in enumerate(samples):
                random.seed(1000 + k)
                host[0].images = imgs
                a = (None, None, s)
                for t in host:
                    a = t(*a)
                out.append((a[0].crop, a[0].flip, a[0].expand, a[0].resize, a[2].boxes, random.getstate()))
            return out
        finally:
            T.NATIVE_SAMPLER = saved
    assert run(False) == run(True)
    # no ground-truth box left (all dropped by an expand): the loops' numpy reduction raises, so does the native path
    picker = [t for t in host if isinstance(t, T.SamplePickerTransform)][0]
    with pytest.raises(ValueError):
        picker(T.ImagePlan(np.zeros((50, 50, 3), np.uint8)), None, Sample('e', [], Size(50, 50)))


def test_native_sampler_exhausted_trials_and_degenerate_pickers():
    """Samplers that never find a window (a tiny box against min_jaccard_overlap = 1.0: all max_trials spent), pickers with a
    single pass-through sampler, and few trials: the native loop consumes exactly the draws the Python loops consume."""
    from ssd_tensorflow_amd import transforms as T
    from ssd_tensorflow_amd.utils import Box, Point, Sample, Size
    tiny = Sample('t', [Box('x', 0, Point(0.5, 0.5), Size(0.01, 0.01))], Size(333, 222))
    big = Sample('b', [Box('x', 0, Point(0.5, 0.5), Size(0.9, 0.9)), Box('y', 1, Point(0.2, 0.3), Size(0.1, 0.2))], Size(640, 480))
    pickers = [
        T.SamplePickerTransform(samplers=[T.SamplerTransform(sample=False)]),
        T.SamplePickerTransform(samplers=[T.SamplerTransform(sample=False), T.build_sampler(1.0, 7), T.build_sampler(0.9, 3)]),
        T.SamplePickerTransform(samplers=[T.SamplerTransform(sample=False)] + [T.build_sampler(o, 50) for o in (0.1, 0.3, 0.5, 0.7, 0.9, 1.0)]),
    ]
    for gt in (tiny, big):
        for pk in pickers:
            for seed in range(25):
                outs = []
                for native in (False, True):
                    saved, T.NATIVE_SAMPLER = T.NATIVE_SAMPLER, native
                    try:
                        random.seed(seed)
                        plan = T.ImagePlan(np.zeros((gt.imgsize.h, gt.imgsize.w, 3), np.uint8))
                        d, _, g = pk(plan, None, gt)
                        outs.append((d.crop, g.boxes, g.imgsize, random.getstate()))
                    finally:
                        T.NATIVE_SAMPLER = saved
                assert outs[0] == outs[1], (gt.filename, seed)
    # the tiny box never satisfies overlap 1.0: only the pass-through candidate (and sometimes 0.9's) remains
    random.seed(3)
    d, _, g = pickers[1](T.ImagePlan(np.zeros((222, 333, 3), np.uint8)), None, tiny)
    assert d.crop is None or d.crop[2] > 0


def test_free_compositions_are_rewritten_into_the_plan():
    """Round 5 (transforms.py:117-391 composes freely): transform lists outside the recipe's order -- crop /
    expand / flip after a flip, two expands, a second photometric pass, photometric steps after a flip or a reorder -- become ONE
    plan of the batch kernel's canonical form.  The plan, executed with the oracle's pixel operations in ITS order, must give the
    pixels of the free composition executed in the USER'S order, exactly; and the ground-truth boxes are the mirror's own."""
    import compose_util as cu
    for li, steps in enumerate(cu.FREE_LISTS + cu.ROUND6_LISTS):
        for rep in range(3):
            img = cu.test_image(100 * li + rep)
            seed = 9000 + 10 * li + rep
            plan, gt = cu.compose_mirror(steps, img, seed)
            want = cu.compose_pixels(steps, img, seed)
            got = cu.run_plan(plan)
            assert got.shape == want.shape == (80, 96, 3), (li, got.shape, want.shape)
            assert np.array_equal(got, want), f'list {li} rep {rep}: the plan is not the composition (max diff {np.abs(got - want).max()})'
            assert len(plan.extra) <= 16 and len(plan.extra_r0) == len(plan.extra)
    # the recipe itself still lands in the canonical slots, nothing spills into the extra list
    from ssd_tensorflow_amd import transforms as T
    from ssd_tensorflow_amd.ssdutils import get_preset_by_name
    preset = get_preset_by_name('vgg300')
    img = cu.test_image(5, (200, 160))
    for seed in range(20):
        random.seed(seed)
        args = (None, None, _sample((200, 160), [(0.5, 0.5, 0.5, 0.5)], [3]))
        for t in [t for t in T.build_train_transforms(preset, 20, 50, 0.5, images={'img': img}) if not isinstance(t, T.LabelCreatorTransform)]:
            args = t(*args)
        assert args[0].extra == []


def test_what_is_not_a_composition_is_refused():
    """What is left outside the plan's form (round 6): a geometric step or a second resize behind ResizeTransform, canvases with
    different histories, too many steps -- NotImplementedError; and Hue / Saturation on the float64 array an expand leaves behind,
    which the REFERENCE refuses too (cv2.cvtColor has no CV_64F path): RuntimeError here, TypeError in the oracle's twin."""
    import compose_util as cu
    for steps in cu.REFUSED_LISTS:
        with pytest.raises((NotImplementedError, RuntimeError)) as ei:
            cu.compose_mirror(steps, cu.test_image(1), 1)
        names = [n for n, _ in steps]
        on_float = any(n in ('hue', 'saturation') for n in names) and 'expand' in names
        assert isinstance(ei.value, NotImplementedError) != on_float, (names, ei.value)
        if on_float:
            with pytest.raises(TypeError):
                cu.compose_pixels(steps, cu.test_image(1), 1)


def test_random_transform_lists_are_composed_or_refused():
    """400 random lists over the whole transform vocabulary, the resize anywhere: a list the mirror ACCEPTS must give, executed as
    a plan, exactly the pixels of its free composition; a list it REFUSES must be one the oracle's free composition refuses as
    well (Hue / Saturation on a float64 array: the reference's cv2.error) or contain one of the documented non-plans (a geometric
    step or a second resize behind the resize, a second expand behind steps that followed the first, more than 16 extra / 4 post
    steps).  Nothing is silently approximated and nothing composable is refused for another reason."""
    import compose_util as cu
    rng = random.Random(77)
    vocab = [('brightness', {}), ('contrast', {}), ('saturation', {}), ('hue', {}), ('reorder', {}), ('expand', dict(max_ratio=1.6)),
             ('crop', cu.WIN), ('crop', dict(window=(0.0, 0.8, 0.1, 1.0))), ('flip', {})]
    pointwise = {'brightness', 'contrast', 'saturation', 'hue', 'reorder'}
    accepted = refused = unsupported_by_reference = 0
    for case in range(400):
        steps = [rng.choice(vocab) for _ in range(rng.randint(1, 9))]
        steps.insert(rng.randint(max(0, len(steps) - 3), len(steps)), cu.RS)        # the resize: last, or up to three steps earlier
        names = [n for n, _ in steps]
        img = cu.test_image(1000 + case, (90, 70))
        seed = 5000 + case
        try:
            plan, _ = cu.compose_mirror(steps, img, seed)
        except RuntimeError as e:
            if isinstance(e, NotImplementedError):
                refused += 1
                r = names.index('resize')
                behind_resize = any(n in ('crop', 'expand') for n in names[r + 1:])
                second_expand = any(n == 'expand' and 'expand' in names[:i] and any(m in pointwise for m in names[names.index('expand'):i])
                                    for i, n in enumerate(names))
                many = sum(n in pointwise for n in names[:r]) > 16 or sum(n in pointwise for n in names[r + 1:]) > 4
                assert behind_resize or second_expand or many, f'case {case}: refused without a reason: {names}'
            else:       # the reference refuses this list itself
                unsupported_by_reference += 1
                with pytest.raises(TypeError):
                    cu.compose_pixels(steps, img, seed)
            continue
        accepted += 1
        want = cu.compose_pixels(steps, img, seed)
        got = cu.run_plan(plan)
        assert got.shape == want.shape and np.array_equal(got, want), f'case {case}: {names} (max diff {np.abs(got - want).max()})'
    assert accepted >= 200 and refused >= 20 and unsupported_by_reference >= 20, (accepted, refused, unsupported_by_reference)
