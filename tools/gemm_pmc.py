"""A few launches of ds2_gemm_f32 on three shapes, for rocprofv3 --pmc passes (tools/gemm_pmc.sh)."""
import os, sys
sys.path.insert(0, 'aes-lac-2018_amd'); sys.path.insert(0, '.')
import torch
from ds2hip import ops
shapes = [('gi NT', 0, 1, 4050, 4800, 800, 0), ('dX NN', 0, 0, 4050, 800, 4800, 0), ('dWih TN', 1, 0, 4800, 800, 4050, 0),
          ('sq NT', 0, 1, 4096, 4096, 4096, 1)]
for name, ta, tb, m, n, k, sk in shapes:
    a = torch.randn((k, m) if ta else (m, k), device='cuda'); b = torch.randn((n, k) if tb else (k, n), device='cuda')
    c = torch.empty(m, n, device='cuda')
    for _ in range(4):
        ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out=c, split_k=sk)
    torch.cuda.synchronize()
