"""Accuracy of ds2_gemm_f32 against an fp64 product, per operand layout, for the kernel family DS2_GEMM_SPLIT selects
(0 = f32-input MFMA, 6 / 9 = bf16 split-operand kernels): max and rms error relative to sum_k |a||b|."""
import os, sys
sys.path.insert(0, 'aes-lac-2018_amd'); sys.path.insert(0, '.')
import torch, numpy as np
from ds2hip import ops
torch.manual_seed(1)
print('DS2_GEMM_SPLIT =', os.environ.get('DS2_GEMM_SPLIT', '0'))
for name, ta, tb, m, n, k, sk in [('NT', 0, 1, 1000, 800, 672, 1), ('NN', 0, 0, 1111, 800, 4800, 1), ('TN', 1, 0, 4800, 800, 3001, 1),
                                  ('TN split', 1, 0, 1600, 800, 4240, 0), ('TT', 1, 1, 300, 200, 515, 1), ('NT ragged', 0, 1, 130, 29, 1600, 1),
                                  ('NT wide-range', 0, 1, 512, 512, 2048, 1)]:
    a = torch.randn((k, m) if ta else (m, k), device='cuda'); b = torch.randn((n, k) if tb else (k, n), device='cuda')
    if 'wide' in name:
        a = a * torch.exp(8 * torch.randn_like(a)); b = b * torch.exp(8 * torch.randn_like(b))
    c = torch.empty(m, n, device='cuda')
    ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=c, split_k=sk)
    a64 = (a.t() if ta else a).double(); b64 = (b.t() if tb else b).double()
    ref = a64 @ b64
    scale = a64.abs() @ b64.abs()
    err = ((c.double() - ref).abs() / scale)
    c2 = c.clone()
    ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=c2, beta=1.0, split_k=1)
    err2 = ((c2.double() - c.double() - ref).abs() / scale)
    print('%-14s M=%5d N=%5d K=%5d  max %.2e rms %.2e   (beta=1: max %.2e)' % (name, m, n, k, float(err.max()), float(err.pow(2).mean().sqrt()), float(err2.max())))
