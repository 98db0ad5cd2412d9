#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05ac; mkdir -p "$O"; cd "$R"
timeout 500 python -m pytest tests/test_gpu_model.py tests/test_gpu_boxes.py tests/test_gpu_drivers.py -q -p no:cacheprovider -x 2>&1 | grep -i -E "passed|failed|error" | tail -5
python - > "$O/infer_detect.txt" 2>&1 <<'PY'
# inference batch 128 + decode/NMS of it, with and without the hand-over (same box, interleaved)
import time, numpy as np, torch
from ssd_tensorflow_amd.ssdvgg import SSDVGG, Session
for dtype in ('bf16', 'f32'):
    sess = Session(0); net = SSDVGG(sess, 'vgg300'); net.build_from_vgg(None, 20, max_batch=128, seed=1, dtype=dtype, training=False)
    x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (128, 300, 300, 3)).astype(np.float32)).cuda()
    net.infer_dev(x); r = net._dev_result(128, True)
    thr = float(np.quantile(r[:, :, :20].max(-1), 0.97))
    def run(arm, n):
        net.set_detect_threshold(thr if arm else None)
        for _ in range(3):
            net.infer_dev(x); net.detect_last_launch(128, thr, None, 200)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            net.infer_dev(x); t = net.detect_last_launch(128, thr, None, 200)
        t.get(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    n = 40 if dtype == 'bf16' else 8
    res = {False: [], True: []}
    for rep in range(3):
        for arm in (False, True):
            res[arm].append(run(arm, n))
    print(dtype, 'infer + detect, batch 128, ms per batch:  scan', ' '.join('%.4f' % v for v in res[False]), '  handed over', ' '.join('%.4f' % v for v in res[True]), flush=True)
    sess.close()
PY
cat "$O/infer_detect.txt" | grep -v amdgpu
