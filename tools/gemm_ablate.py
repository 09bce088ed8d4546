"""Timing-only ablations of the two-buffer GEMM main loops (libds2hip_gemmabl<N>.so, built by build.build_gemm_variant with
-DDS2_GEMM_ABL=N; results are WRONG): 1 = no LDS stores, 2 = no global loads, 4 = no LDS operand reads; split-operand kernels
also 8 = two of the six products, 16 = no split arithmetic (raw bits stored).  DS2_GEMM_SPLIT=0 times the f32-input kernels."""
import os, sys
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(_ROOT, 'aes-lac-2018_amd')); sys.path.insert(0, _ROOT)
import numpy as np, torch
from ds2hip import lib
v = os.environ.get('ABL', '')
if v:
    lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), 'libds2hip_gemmabl%s.so' % v)
from ds2hip import ops
os.environ['DS2_GEMM_V2'] = '0'
for name, ta, tb, m, n, k in [('sq NT', 0, 1, 4096, 4096, 4096), ('sq NN', 0, 0, 4096, 4096, 4096), ('sq TN', 1, 0, 4096, 4096, 4096),
                              ('k800 NT', 0, 1, 4096, 4096, 800), ('gi NT', 0, 1, 4050, 4800, 800), ('dX NN', 0, 0, 4050, 800, 4800),
                              ('dWih TN', 1, 0, 4800, 800, 4050)]:
    a = torch.randn((k, m) if ta else (m, k), device='cuda'); b = torch.randn((n, k) if tb else (k, n), device='cuda')
    c = torch.empty(m, n, device='cuda'); sk = 1 if name.startswith(('sq', 'k800')) else 0
    for _ in range(3): ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=c, split_k=sk)
    torch.cuda.synchronize(); ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=c, split_k=sk); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    t = float(np.median(ts))
    print('ABL=%-2s %s  %8.1f us  %6.1f TFLOP/s' % (v or '0', name, t * 1e6, 2.0 * m * n * k / t / 1e12), flush=True)
