import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from ssd_tensorflow_amd.ssdvgg import SSDVGG, Session
from ssd_tensorflow_amd._lib import lib, check
import bench
dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
def make(b, stream):
    sess = Session(0); net = SSDVGG(sess, 'vgg300'); net.build_from_vgg(None, 20, max_batch=b, seed=42, dtype=dtype)
    net.build_optimizer(learning_rate=0.00075)
    net.set_stream(stream.cuda_stream)
    rng = np.random.default_rng(1234)
    x = torch.from_numpy(rng.integers(0, 256, (b, 300, 300, 3)).astype(np.float32)).cuda()
    y = torch.empty((b, 8732, 25), dtype=torch.float32, device='cuda')
    gt, cls, offs = bench.synth_gt(rng, b)
    check(lib.ssd_encode_labels_dev(b'vgg300', 20, 0, gt.ctypes.data, cls.ctypes.data, offs.ctypes.data, b, y.data_ptr(), None))
    return sess, net, x, y
torch.cuda.synchronize()
def run(nets, steps):
    for _ in range(3):
        for s, n, x, y in nets: n.train_step_dev(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        for s, n, x, y in nets: n.train_step_dev(x, y)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return sum(x.shape[0] for _, _, x, _ in nets) * steps / dt
one = [make(32, torch.cuda.Stream())]
print('1 x b32:', round(run(one, 20), 1), 'img/s')
one[0][0].close()
two = [make(16, torch.cuda.Stream()), make(16, torch.cuda.Stream())]
print('2 x b16 on two streams:', round(run(two, 20), 1), 'img/s')
single16 = [two[0]]
print('1 x b16:', round(run(single16, 20), 1), 'img/s')
