"""where does the data-parallel step's single-GPU overhead come from?  variants of one training step, same kernels."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import boxes as ob, ssdvgg_ref as ref
from ssd_tensorflow_amd.ssdvgg import SSDVGG, Session

dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
if os.environ.get('DP_SPLIT_PG') == '1':      # a single-rank RCCL group exists (as in bench.py --force-collectives), whether or not it is used
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544'); os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    if os.environ.get('DP_SPLIT_PG_USE') == '1':
        t = torch.zeros(1024, device='cuda'); dist.all_reduce(t); torch.cuda.synchronize()
steps = 40 if dtype == 'bf16' else 12
preset = ob.get_preset('vgg300')
rng = np.random.default_rng(0)
x, y, _ = ref.synth_batch(rng, 32, preset)
with Session(0) as sess:
    net = SSDVGG(sess, 'vgg300')
    net.build_from_vgg(None, 20, max_batch=32, dtype=dtype)
    net.build_optimizer(learning_rate=1e-4, weight_decay=0.0005, momentum=0.9)
    net.set_stream(torch.cuda.current_stream().cuda_stream)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    cs = torch.cuda.Stream()

    def plain():
        net.train_step_dev(xd, yd)

    def split():
        net.forward_backward_dev(xd, yd)
        net.apply_gradients_dev(1.0)

    def split_hop():
        net.forward_backward_dev(xd, yd)
        cs.wait_stream(torch.cuda.current_stream())
        ev = torch.cuda.Event(); ev.record(cs)
        torch.cuda.current_stream().wait_event(ev)
        net.apply_gradients_dev(1.0)

    def staged(bucket):
        def f():
            side = net.use_torch_wgrad_stream()
            net.forward_dev(xd, yd)
            ranges = net.backward_staged(yd, 32, bucket, sync_main=False)
            evs = []
            for off, cnt in ranges:
                with torch.cuda.stream(side):
                    ev = torch.cuda.Event(); ev.record(side); evs.append(ev)
            for ev in evs:
                torch.cuda.current_stream().wait_event(ev)
            net.apply_gradients_dev(1.0)
        return f

    def split_rccl():
        net.forward_backward_dev(xd, yd)
        dist.all_reduce(net.grads_flat)
        net.apply_gradients_dev(1.0)

    variants = [('plain train_step_dev', plain), ('forward_backward_dev + apply', split), ('... + event hop through a side stream', split_hop),
                ('staged backward, 44 MB ranges, events only', staged(11_000_000)), ('plain again', plain)]
    if os.environ.get('DP_SPLIT_PG') == '1':
        variants = [('PG: plain train_step_dev', plain), ('PG: fb + dist.all_reduce(arena) + apply', split_rccl), ('PG: plain again', plain)]
    for name, f in variants:
        for _ in range(6):
            f()
        torch.cuda.synchronize()
        out = []
        for r in range(2):
            t0 = time.perf_counter()
            for _ in range(steps):
                f()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / steps * 1e3)
        print('%-52s %s %s ms/step' % (name, dtype, ' '.join('%.3f' % v for v in out)), flush=True)
