"""GPU aid: the learning test's driver run (tests/test_gpu_learning.py run_driver) under a given schedule; prints loss / mAP by epoch.
   tools/learn_probe.py <dtype> <lr-values> <lr-boundaries> [epochs] [feeder workers]"""
import os, sys, tempfile, pathlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import test_gpu_learning as T
dtype, lrv, lrb = sys.argv[1], sys.argv[2], sys.argv[3]
epochs = int(sys.argv[4]) if len(sys.argv) > 4 else T.EPOCHS
workers = int(sys.argv[5]) if len(sys.argv) > 5 else 4
T.LR_VALUES, T.LR_BOUNDARIES = lrv, lrb
r = T.run_driver(pathlib.Path(tempfile.mkdtemp()), 'probe', dtype, epochs=epochs, workers=workers)
print(dtype, lrv, lrb, 'workers', workers, 'SSD_WINOGRAD=' + os.environ.get('SSD_WINOGRAD', 'default'))
print('  train total by epoch', [round(t[0], 2) for t in r['train']])
print('  mAP training by epoch', [round(m[0], 2) for m in r['maps']])
print('  final: train %.3f valid %.3f mAP %.4f / %.4f' % (r['train'][-1][0], r['valid'][-1][0], r['maps'][-1][0], r['maps'][-1][1]))
