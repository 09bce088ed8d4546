"""ds2_gemm_f32 against torch.matmul (rocBLAS / hipBLASLt fp32) on the step's GEMM shapes."""
import os, sys
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(_ROOT, 'aes-lac-2018_amd')); sys.path.insert(0, _ROOT)
import torch, numpy as np
from ds2hip import ops
def tm(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e-3)
    return float(np.median(ts))
rows = int(os.environ.get('ROWS', '4050'))
for name, ta, tb, m, n, k in [('gi NT', 0, 1, rows, 4800, 800), ('dX NN', 0, 0, rows, 800, 4800), ('dWih TN', 1, 0, 4800, 800, rows),
                              ('dWhh TN', 1, 0, 1600, 800, rows), ('sq NT', 0, 1, 4096, 4096, 4096)]:
    a = torch.randn((k, m) if ta else (m, k), device='cuda'); b = torch.randn((n, k) if tb else (k, n), device='cuda')
    c = torch.empty(m, n, device='cuda')
    t1 = tm(lambda: ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=c, split_k=0))
    aa = a.t() if ta else a; bb = b.t() if tb else b
    t2 = tm(lambda: torch.matmul(aa, bb, out=c))
    fl = 2.0 * m * n * k
    print('%-8s %5d %5d %5d  ds2 %7.1f us %6.1f TF | torch %7.1f us %6.1f TF' % (name, m, n, k, t1 * 1e6, fl / t1 / 1e12, t2 * 1e6, fl / t2 / 1e12))
