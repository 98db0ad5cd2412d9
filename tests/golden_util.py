"""Loaders for tests/golden/*.npz (made by tools/make_golden.py from the imported reference)."""
import os
import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def dense_pred(d, pi):
    """Rebuild the dense [A,25] f32 prediction: rows not stored are pure background."""
    A = int(d['A'][0])
    row = np.zeros(25, np.float32); row[20] = 1
    pred = np.tile(row, (A, 1))
    pred[d[f'predrows_{pi}']] = d[f'predvals_{pi}']
    return pred


def detect_cases(d):
    """Yield (pi, tag, thr, cap, max_out) for every stored setting."""
    for pi in range(int(d['npred'][0])):
        for si in range(int(d['nset'][0])):
            tag = f'{pi}_{si}'
            if f'set_{tag}' not in d.files:
                continue
            thr, cap, max_out = d[f'set_{tag}']
            yield pi, tag, float(thr), (None if cap < 0 else int(cap)), (None if max_out < 0 else int(max_out))
