"""What does the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) sustain on this chip for the GEMMs the big layers are?
A reference point for the roofline discussion: the convolutions here are hand-written and never call it.
    python tools/probes/gemm_ceiling.py"""
import time
import torch

def run(m, n, k, dtype, secs=1.5):
    a = torch.randn(m, k, device='cuda', dtype=dtype)
    b = torch.randn(k, n, device='cuda', dtype=dtype)
    for _ in range(5):
        a @ b
    torch.cuda.synchronize()
    it = 0
    t0 = time.perf_counter()
    while True:
        for _ in range(20):
            a @ b
        it += 20
        torch.cuda.synchronize()
        if time.perf_counter() - t0 > secs:
            break
    dt = time.perf_counter() - t0
    return 2.0 * m * n * k * it / dt / 1e12, dt / it * 1e6

shapes = [('conv4_2 fwd  (M=46208 N=512 K=4608)', 46208, 512, 4608), ('conv3_2 fwd  (M=180000 N=256 K=2304)', 180000, 256, 2304),
          ('conv2_2 fwd  (M=720000 N=128 K=1152)', 720000, 128, 1152), ('conv5_2 fwd  (M=11552 N=512 K=4608)', 11552, 512, 4608),
          ('conv4_2 wgrad (M=4608 N=512 K=46208)', 4608, 512, 46208), ('square 8192', 8192, 8192, 8192)]
for dtype, name in ((torch.bfloat16, 'bf16'), (torch.float32, 'f32')):
    for label, m, n, k in shapes:
        tf, us = run(m, n, k, dtype)
        print('%-5s %-40s %8.1f TFLOP/s  %9.1f us' % (name, label, tf, us), flush=True)
