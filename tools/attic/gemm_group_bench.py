import os, sys
sys.path.insert(0, 'aes-lac-2018_amd'); sys.path.insert(0, '.')
import torch, numpy as np
from ds2hip import ops
rows = 4240
dgh = [torch.randn(rows, 2400, device='cuda') for _ in range(2)]      # per direction: [rows][rz 1600 | n 800]
h = [torch.randn(rows, 800, device='cuda') for _ in range(2)]
c = [torch.zeros(m, 800, device='cuda') for m in (1600, 800, 1600, 800)]
probs = []
for d in range(2):
    probs.append((dgh[d].data_ptr(), 2400, 1600, h[d].data_ptr(), 800, c[2 * d].data_ptr(), 800))
    probs.append((dgh[d].data_ptr() + 1600 * 4, 2400, 800, h[d].data_ptr(), 800, c[2 * d + 1].data_ptr(), 800))
def tm(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))
fl = 2.0 * rows * 800 * (1600 + 800) * 2
for mode in (0, 6):
    ops.gemm_split_mode(mode)
    t = tm(lambda: ops.gemm_tn_group(probs, 800, rows))
    print('mode %d grouped dW_hh (4 problems): %.1f us  %.1f TF' % (mode, t, fl / t / 1e6))
    x = torch.randn(rows, 800, device='cuda'); dgi = torch.randn(rows, 4800, device='cuda'); g = torch.zeros(4800, 800, device='cuda')
    t = tm(lambda: ops.gemm(dgi, x, trans_a=True, out=g, split_k=0))
    print('mode %d dW_ih 4800x800x%d: %.1f us  %.1f TF' % (mode, rows, t, 2.0 * rows * 800 * 4800 / t / 1e6))
