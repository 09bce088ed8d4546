"""Stand-alone timing of ds2_gemm_f32 on the shapes the DeepSpeech2 step uses (B=10, T~424 -> rows 4240)."""
import os, sys
sys.path.insert(0, 'aes-lac-2018_amd'); sys.path.insert(0, '.')
import torch, numpy as np
from ds2hip import lib
if os.environ.get('DS2_LIB_VARIANT'): lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), 'libds2hip_%s.so' % os.environ['DS2_LIB_VARIANT'])
from ds2hip import ops
rows = int(os.environ.get('ROWS', '4240'))
shapes = [('warm NT', 0, 1, 4096, 4096, 4096, 1), ('gi   NT', 0, 1, rows, 4800, 800, 0), ('gi0  NT', 0, 1, rows, 4800, 672, 0),
          ('dX   NN', 0, 0, rows, 800, 4800, 0), ('dWih TN', 1, 0, 4800, 800, rows, 0),
          ('dWhh TN', 1, 0, 1600, 800, rows, 0), ('dWhn TN', 1, 0, 800, 800, rows, 0),
          ('sq   NT', 0, 1, 4096, 4096, 4096, 1), ('k800 NT', 0, 1, 4096, 4096, 800, 1), ('k800 NT', 0, 1, 4096, 4736, 800, 1),
          ('k800 NT', 0, 1, 8192, 4096, 800, 1), ('k800 NT', 0, 1, 2048, 4096, 800, 1)]
for name, ta, tb, m, n, k, sk in shapes:
    a = torch.randn((k, m) if ta else (m, k), device='cuda'); b = torch.randn((n, k) if tb else (k, n), device='cuda')
    c = torch.empty(m, n, device='cuda')
    for _ in range(3): ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=c, split_k=sk)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=c, split_k=sk); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    t = float(np.median(ts))
    print('%s M=%5d N=%5d K=%5d  %8.1f us  %6.1f TFLOP/s' % (name, m, n, k, t * 1e6, 2.0 * m * n * k / t / 1e12))
