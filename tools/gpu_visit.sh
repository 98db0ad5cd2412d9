#!/bin/bash
# One GPU-box visit (overwritten per visit; results land in gpurun_out/<tag>/).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=r06a
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
# (1) baseline step times of the tree as it stands
for dt in bf16 f32; do
  st=40; [ $dt = f32 ] && st=10
  python tools/step_time.py --dtype $dt --steps $st --reps 3 --tag base 2>/dev/null >> "$O/step_time.txt"
done
cat "$O/step_time.txt"
# (2) decode hand-over where decode is NOT noise: low thresholds (infer.py:233-235 runs at 0.01)
python - > "$O/infer_detect_lowthr.txt" 2>&1 <<'PY'
import time, numpy as np, torch
from ssd_tensorflow_amd.ssdvgg import SSDVGG, Session
for dtype in ('bf16',):
    sess = Session(0); net = SSDVGG(sess, 'vgg300'); net.build_from_vgg(None, 20, max_batch=128, seed=1, dtype=dtype, training=False)
    x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (128, 300, 300, 3)).astype(np.float32)).cuda()
    net.infer_dev(x); r = net._dev_result(128, True)
    conf = r[:, :, :20].max(-1)
    for qq in (0.97, 0.9, 0.5, None):
        thr = 0.01 if qq is None else float(np.quantile(conf, qq))
        def run(arm, n):
            net.set_detect_threshold(thr if arm else None)
            for _ in range(3):
                net.infer_dev(x); net.detect_last_launch(128, thr, None, 200)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n):
                net.infer_dev(x); t = net.detect_last_launch(128, thr, None, 200)
            t.get(); torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        res = {False: [], True: []}
        for rep in range(3):
            for arm in (False, True):
                res[arm].append(run(arm, 40))
        print(dtype, 'thr', thr, 'cands/img', float((conf >= thr).sum(1).mean()), 'infer+detect b128 ms: scan', ' '.join('%.4f' % v for v in res[False]), ' handed', ' '.join('%.4f' % v for v in res[True]), flush=True)
    sess.close()
PY
grep -v amdgpu "$O/infer_detect_lowthr.txt"
# (3) does RCCL accept two ranks on ONE device?
cat > /tmp/nccl2.py <<'PY'
import os, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda:0'))
t = torch.ones(1024, device='cuda') * (dist.get_rank() + 1)
dist.all_reduce(t); torch.cuda.synchronize()
print('rank', dist.get_rank(), 'sum', float(t[0]))
dist.destroy_process_group()
PY
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 /tmp/nccl2.py > "$O/nccl_two_ranks_one_gpu.txt" 2>&1; echo "nccl2 rc=$?" >> "$O/nccl_two_ranks_one_gpu.txt"
tail -15 "$O/nccl_two_ranks_one_gpu.txt"
