"""-m gpu: does the step LEARN?  The parity tests compare single steps with the oracle; this file trains.

The reference validates itself by mAP (README.md:26-29; train.py:311-331 computes the AP of the training and the
validation sample every epoch).  No dataset can be downloaded here, so the data is the learnable synthetic set
`TrainingData('shapes')`: 1..3 textured rectangles per image on low-contrast noise, class = texture (horizontal /
vertical / diagonal stripes, checkerboard) -- texture, not colour, because the reference's training recipe permutes
channels and shifts hue / saturation.  Everything runs through the product's own driver, `ssd_tensorflow_amd.train.main`
(feeder with worker processes, StepLoop, decode + NMS of every batch from epoch 2 on, GPU APCalculator), from Xavier
weights (there is no vgg.zip), SGD + momentum at the reference's magnitude of learning rate.

Schedule: 96 steps at 3e-4, the reference's 7.5e-4 (train.py:66) until step 768, then 1e-4 -- 1280 steps at batch 32 over
1024 training images.  (Round 6 moved the fp32 trunk to the Winograd form.  On Lavin & Gray's interpolation points its rounding, 1e-5, was
a hundred times the direct kernels' -- at this rate, the edge of stability (next paragraph), kick enough for a spike in two of two such trees
(step ~740 of this schedule: loss 3.4 -> 8.5, mAP 0.91 -> 0.0; step ~1,090 of a 3,000-step run at 7.5e-4) where the direct kernels' runs
had none: profiles/r06_ar_learn_probe.txt, r06_br_learn_probe_long_reference_lr.txt.  On the points the round ended with (0, +-3/4, +-3/2:
2.3-3.8x less rounding) the same runs show no spike -- 3,520 steps at 7.5e-4 end at loss 2.219 / mAP 1.000, this schedule at 2.758 / 1.000 /
1.000 against bf16's 2.721 / 0.977 / 1.000: profiles/r06_bu_learn_probe_long_scaled_points.txt, r06_bv_learn_probe_test_schedule.txt.  In between the
test ran its middle phase at 6e-4; at the trained state every filter gradient of the Winograd step agrees with the direct step's to 5e-5:
profiles/r06_aq_wino_probe.txt.)  Measured on an MI355X (profiles/r04_l_learning_probe.txt: the same run without the final decay):
fp32 passes mAP 0.5 on the training sample at step ~480, 0.9 at ~670 and sits at 1.000 / 1.000 (training / held-out)
from step ~900 on, total loss 16.5 -> 2.6 (of which 2.18 is the l2 term); bf16 follows the same curve to 0.95 at step
~930.  Left at 7.5e-4 or 1e-3 for thousands of steps, a run in EITHER dtype occasionally collapses (a loss spike, mAP back to
~0, then it re-learns: profiles/r04_k_learning_probe.txt) -- VGG-16 from Xavier weights on raw 0..255 inputs without
normalisation layers is at the edge of stability at these rates, which is what the final decay is for.  With the full
augmentation recipe on (--augment true) the same net learns more slowly (training mAP 0.28 after 1600 steps and rising), so
the thresholds are asserted on the un-augmented set and the augmented run is asserted to make progress."""
import contextlib
import io
import re

import numpy as np
import pytest
import torch

from oracle import boxes as ob
from oracle import ssdvgg_ref as ref
from ssd_tensorflow_amd import train
from ssd_tensorflow_amd.ssdvgg import SSDVGG, Session, LearningRate
from ssd_tensorflow_amd.training_data import TrainingData

pytestmark = pytest.mark.gpu

EPOCHS, NTRAIN, NVALID, BATCH = 40, 1024, 128, 32
LR_VALUES, LR_BOUNDARIES = '0.0003;0.00075;0.0001', '96;768'


def run_driver(tmp_path, tag, dtype, epochs=EPOCHS, augment='false', workers=4):
    out = io.StringIO()
    argv = ['--name', str(tmp_path / ('run_' + tag)), '--tensorboard-dir', str(tmp_path / ('tb_' + tag)), '--data-dir', 'shapes',
            '--synthetic-train', str(NTRAIN), '--synthetic-valid', str(NVALID), '--num-workers', str(workers), '--batch-size', str(BATCH),
            '--checkpoint-interval', '1000', '--lr-values', LR_VALUES, '--lr-boundaries', LR_BOUNDARIES, '--epochs', str(epochs), '--dtype', dtype,
            '--augment', augment]
    with contextlib.redirect_stdout(out):
        rc = train.main(argv)
    text = out.getvalue()
    assert rc == 0, text[-2000:]
    tr = [tuple(map(float, m.groups())) for m in re.finditer(r'\[i\] Train +\d+/\d+ +total ([\d.naninf]+) +localization ([\d.naninf]+) +confidence ([\d.naninf]+)', text)]
    va = [tuple(map(float, m.groups())) for m in re.finditer(r'\[i\] Valid +\d+/\d+ +total ([\d.naninf]+) +localization ([\d.naninf]+) +confidence ([\d.naninf]+)', text)]
    maps = [tuple(map(float, m.groups())) for m in re.finditer(r'\[i\] mAP +\d+/\d+ +training ([\d.]+) +validation ([\d.]+)', text)]
    assert len(tr) == epochs and len(va) == epochs and len(maps) == epochs - 1, text[-2000:]
    return dict(train=tr, valid=va, maps=maps, text=text)


def test_shapes_training_converges_in_both_dtypes(tmp_path):
    """1280 steps of the HIP training step from Xavier weights: the loss falls, the detector detects -- in fp32 and, from the
    same seed and the same batches, in bf16."""
    res = {}
    for dtype in ('bf16', 'f32'):
        r = res[dtype] = run_driver(tmp_path, dtype, dtype)
        first, last = r['train'][0], r['train'][-1]
        tmap, vmap = r['maps'][-1]
        print(f'    {dtype}: train loss {first[0]:.3f} -> {last[0]:.3f} (localization {first[1]:.3f} -> {last[1]:.3f}, confidence '
              f'{first[2]:.3f} -> {last[2]:.3f}); valid loss {r["valid"][-1][0]:.3f}; mAP training {tmap:.4f} validation {vmap:.4f}; '
              f'mAP by epoch {[round(m[0], 2) for m in r["maps"][::5]]}')
        assert np.isfinite(last[0])
        assert last[0] * 3.0 <= first[0], 'the total loss must fall by at least 3x'
        assert last[1] * 3.0 <= first[1] and last[2] * 3.0 <= first[2], 'both data terms must fall by at least 3x'
        assert tmap >= 0.5, 'training-sample mAP (train.py:311-331) must reach 0.5'
        assert vmap >= 0.5, 'held-out mAP must reach 0.5'
    # bf16 against fp32, same seed, same batches, at the end of training.  Measured (profiles/r04_m_gpu_learning_tests.log): mAP
    # 0.976 / 0.977 (bf16, training / held-out) against 0.999 / 0.976 (fp32); total loss 3.019 against 2.903 (+4.0 %), held-out
    # 3.395 against 3.293 (+3.1 %).  The two are different trajectories of the same chaotic training run, so the bounds leave room.
    # Round 6 (fp32 on the Winograd form, scaled interpolation points): 0.977 / 1.000 against 1.000 / 1.000, total loss 2.721 against 2.758
    # (-1.3 %), held-out 3.162 against 3.163.
    a, b = res['bf16'], res['f32']
    assert abs(a['maps'][-1][0] - b['maps'][-1][0]) <= 0.05 and abs(a['maps'][-1][1] - b['maps'][-1][1]) <= 0.05
    assert abs(a['train'][-1][0] - b['train'][-1][0]) <= 0.08 * b['train'][-1][0]
    assert abs(a['valid'][-1][0] - b['valid'][-1][0]) <= 0.08 * b['valid'][-1][0]


def test_augmented_shapes_training_makes_progress(tmp_path):
    """The whole training recipe (expand, sample-picker crops, photometric distortion, flip: process_dataset.py:66-140) in front
    of the same step: slower (see the module docstring), so only progress is asserted -- and that decode + NMS of the training
    batches yields true positives."""
    r = run_driver(tmp_path, 'aug', 'bf16', epochs=20, augment='true', workers=8)
    first, last = r['train'][0], r['train'][-1]
    print(f'    augmented bf16: train loss {first[0]:.3f} -> {last[0]:.3f}; valid loss {r["valid"][0][0]:.3f} -> {r["valid"][-1][0]:.3f}; '
          f'training mAP by epoch {[round(m[0], 3) for m in r["maps"][::3]]}')
    assert np.isfinite(last[0])
    assert r['valid'][-1][0] < 0.7 * r['valid'][0][0]


def test_first_training_steps_track_the_oracle_on_shapes():
    """fp32, batch 4, momentum 0.9, lr 1e-4: the HIP losses against oracle.RefModel.train_step on the same batches, optimizer step
    after optimizer step.  The two trajectories are separate computations of a chaotic system (tests/test_gpu_model.py header: a
    relu mask, a pool argmax or a hard-negative pick flips wherever two fp32 values agree to ~1e-6), so the agreement decays with
    the step count and HOW fast depends on the batches: on round 4's data set 1e-7, 9e-6, 8e-5, 5e-4, 8e-4, 8e-4 for steps 0-5, on
    another 7.7e-3 by step 5 (one mined negative swapped; profiles/r05_zz_gpu_tests.log).  A bound tuned on one set proves that set,
    so the claim is asserted as a distribution: THREE data sets and weight seeds, four optimizer steps each (twelve oracle steps in
    all, 3-7 s each on the pool's hosts) -- every set within 1e-3 for steps 0-2 (measured <= 1e-4 there), every set within 1e-2 at
    step 3, and the median of the three within 2e-3 at step 3.  SSD_TEST_TRACK_STEPS lengthens the runs (no bound past step 3 but
    3e-2)."""
    import os
    b, steps = 4, int(os.environ.get('SSD_TEST_TRACK_STEPS', 4))
    preset = ob.get_preset('vgg300')
    lr = LearningRate([float(os.environ.get('SSD_TEST_TRACK_LR', 1e-4))], [])
    table = []
    for data_seed, weight_seed in ((5, 11), (6, 12), (7, 13)):
        td = TrainingData('shapes', 'vgg300', num_train=b * steps, num_valid=b, seed=data_seed, device_tensors=False)
        w = ref.init_params(preset, 20, seed=weight_seed, alive=True)
        m = ref.RefModel('vgg300', params=w)
        m.set_optimizer(lr.values, lr.boundaries, 0.9, 0.0005)
        errs = []
        with Session(0) as sess:
            net = SSDVGG(sess, 'vgg300')
            net.build_from_vgg(None, 20, max_batch=b, weights=w)
            net.build_optimizer(learning_rate=lr, weight_decay=0.0005, momentum=0.9)
            for k, (x, y, gt) in enumerate(td.train_generator(b)):
                if k >= steps:
                    break
                x, y = np.asarray(x, np.float32), np.asarray(y, np.float32)
                _, L_ref = m.train_step(x, y)
                L, _ = sess.run([net.losses, net.optimizer], feed_dict={net.image_input: x, net.labels: y})
                errs.append(max(abs(L[n] - L_ref[n]) / abs(L_ref[n]) for n in ('total', 'localization', 'confidence', 'l2')))
        assert len(errs) == steps
        print(f'    data set {data_seed}, weights {weight_seed}: worst relative loss error per step', ' '.join(f'{e:.1e}' for e in errs))
        table.append(errs)
    t = np.array(table)
    assert t[:, :3].max() < 1e-3, 'steps 0-2 must agree to 1e-3 on every data set'
    assert t[:, 3].max() < 1e-2 and np.median(t[:, 3]) < 2e-3, 'step 3: every set within 1e-2, the median within 2e-3'
    assert t.max() < 3e-2
