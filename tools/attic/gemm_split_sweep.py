"""Split-K sweep of the weight-gradient GEMM shapes (TN, K = T*B rows)."""
import os, sys
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(_ROOT, 'aes-lac-2018_amd')); sys.path.insert(0, _ROOT)
import torch, numpy as np
from ds2hip import ops
rows = int(os.environ.get('ROWS', '4050'))
only = os.environ.get('ONLY')
for name, ta, tb, m, n, k in [('dWih TN', 1, 0, 4800, 800, rows), ('dWih0 TN', 1, 0, 4800, 672, rows), ('dWhh TN', 1, 0, 1600, 800, rows), ('dWhn TN', 1, 0, 800, 800, rows), ('dX NN', 0, 0, rows, 800, 4800), ('sq TN', 1, 0, 4096, 4096, 4096), ('sq NT', 0, 1, 4096, 4096, 4096)]:
    if only and only not in name: continue
    a = torch.randn((k, m) if ta else (m, k), device='cuda'); b = torch.randn((n, k) if tb else (k, n), device='cuda')
    c = torch.empty(m, n, device='cuda')
    out = []
    for sk in [int(v) for v in os.environ.get('SPLITS', '0,0,4,8,12,16,24,32').split(',')]:
        for _ in range(2): ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=c, split_k=sk)
        torch.cuda.synchronize(); ts = []
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=c, split_k=sk); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3)
        t = float(np.median(ts)); out.append('%d:%.0fus/%.0fTF' % (sk, t * 1e6, 2.0 * m * n * k / t / 1e12))
    print(name, m, n, k, ' '.join(out))
